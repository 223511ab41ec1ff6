"""ctypes binding of libdrn.so (the C ABI of include/drn.h).

PyTorch-ROCm tensors in, raw device pointers + shapes + the current HIP stream out.
There is NO fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_LIB = None
_LIB_NAME = "libdrn.so"

EPI_NONE, EPI_GELU, EPI_GATE_RES = 0, 1, 2
ACT_NONE, ACT_SILU = 0, 1

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

SUB_FA, SUB_CA, SUB_MLP = 0, 1, 2


class DitSub(Structure):                      # drn_dit_sub (include/drn.h)
    _fields_ = [("kind", c_int32), ("site", c_int32), ("ca_index", c_int32), ("reserved", c_int32),
                ("w_a", c_void_p), ("w_b", c_void_p), ("qn", c_void_p), ("kn", c_void_p)]


class DitForwardArgs(Structure):              # drn_dit_forward_args (include/drn.h), field for field
    _fields_ = [("struct_bytes", c_int64), ("S", c_int64), ("B", c_int64), ("D", c_int64), ("hidden", c_int64),
                ("heads", c_int32), ("n_sub", c_int32), ("subs", POINTER(DitSub)),
                ("shift", c_void_p), ("scale", c_void_p), ("gate", c_void_p),
                ("shift_site_stride", c_int64), ("scale_site_stride", c_int64), ("gate_site_stride", c_int64),
                ("addvec", c_void_p), ("addvec_stride", c_int64), ("cos", c_void_p), ("sin", c_void_p),
                ("P", c_void_p), ("kpad", c_int64), ("w_patch", c_void_p),
                ("final_shift", c_void_p), ("final_scale", c_void_p), ("w_final", c_void_p), ("n_final", c_int64),
                ("X", c_void_p), ("H", c_void_p), ("QKV", c_void_p), ("O", c_void_p), ("U", c_void_p), ("Y", c_void_p),
                ("gemm_ws", c_void_p), ("gemm_ws_bytes", c_int64), ("attn_ws", c_void_p), ("attn_ws_bytes", c_int64),
                ("timer", c_void_p), ("eps", c_float), ("reserved", c_int32)]


# name -> argtypes (restype is int unless listed in _RESTYPES); must match include/drn.h one-to-one
SIGNATURES = {
    "drn_attention_plan": [_I, _L, _L, POINTER(c_int64)],
    "drn_dit_forward": [POINTER(DitForwardArgs), _P],
    "drn_dit_forward_args_bytes": [],
    "drn_dit_sub_bytes": [],
    "drn_dit_forward_gemm_workspace_bytes": [_L, _L, _L, _L, _L, _L],
    "drn_dit_forward_attn_workspace_bytes": [_L, _I, _L],
    "drn_timer_create": [_I, _I],
    "drn_timer_destroy": [_P],
    "drn_timer_count": [_P],
    "drn_timer_seen": [_P, _I],
    "drn_timer_read": [_P, _I, POINTER(c_int), POINTER(c_float), POINTER(c_double), POINTER(c_double)],
    "drn_abi_version": [],
    "drn_error_string": [_I],
    "drn_gemm_bf16": [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _P, _P, _L, _L, _P],
    "drn_gemm_bf16_blocked": [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _P, _P, _L, _L, _L, _L, _L, _L, _P],
    "drn_gemm_bf16_splitk": [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _P, _P, _L, _L, _I, _P, _P],
    "drn_gemm_splitk_workspace_bytes": [_L, _L, _I],
    "drn_gemm_bf16_f32out": [_P, _P, _P, _L, _L, _L, _L, _L, _P],
    "drn_gemm_bf16_splitk_partials": [_P, _P, _L, _L, _L, _L, _L, _L, _I, _P, _P],
    "drn_splitk_gate_res_ln_modulate": [_P, _I, _P, _P, _P, _P, _P, _P, _L, _L, _L, _F, _P],
    "drn_gemm_splitk_choice": [_L, _L, _L],
    "drn_gemm_tile_choice": [_L, _L],
    "drn_gemm_force_tile": [_I],
    "drn_gemm_force_res_prefetch": [_I],
    "drn_gemm_tall_force_shape": [_I],
    "drn_gemv_bf16": [_P, _P, _P, _L, _L, _I, _I, _L, _L, _L, _L, _L, _P, _L, _L, _P, _L, _L, _I, _P],
    "drn_ln_modulate": [_P, _P, _P, _P, _P, _L, _L, _L, _F, _P],
    "drn_ln_force_kernel": [_I],
    "drn_bcast_add": [_P, _P, _L, _L, _L, _P],
    "drn_permute_021": [_P, _P, _L, _L, _L, _P],
    "drn_rmsnorm": [_P, _P, _P, _L, _L, _F, _P],
    "drn_qk_norm_rope": [_P, _P, _P, _P, _P, _P, _L, _I, _L, _L, _L, _L, _F, _P],
    "drn_attention_bf16": [_P, _P, _P, _P, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P],
    "drn_attention_splitkv_bf16": [_P, _P, _P, _P, _I, _I, _L, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _I, _P, _P],
    "drn_attention_splitkv_workspace_bytes": [_I, _I, _L, _I],
    "drn_attention_force_shape16": [_I],
    "drn_patchify_concat": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _P],
    "drn_unpatchify": [_P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "drn_edm_scale_input": [_P, _P, _L, _F, _P],
    "drn_edm_step": [_P, _P, _P, _L, _F, _F, _F, _F, _P],
    "drn_cfg_combine": [_P, _P, _P, _L, _F, _P],
    "drn_postprocess_u8": [_P, _P, _I, _I, _I, _I, _I, _P],
}
_RESTYPES = {"drn_error_string": c_char_p, "drn_attention_splitkv_workspace_bytes": c_int64,
             "drn_gemm_splitk_workspace_bytes": c_int64, "drn_dit_forward_gemm_workspace_bytes": c_int64,
             "drn_dit_forward_attn_workspace_bytes": c_int64, "drn_dit_forward_args_bytes": c_int64, "drn_dit_sub_bytes": c_int64, "drn_timer_create": c_void_p, "drn_timer_destroy": None,
             "drn_ln_force_kernel": None, "drn_attention_force_shape16": None}


def library_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load_library():
    """Load libdrn.so and bind every symbol declared in include/drn.h.  Raises if anything is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
                           "there is no CPU / eager fallback for the hot path")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.drn_abi_version() != 1:
        raise RuntimeError("libdrn.so ABI version mismatch")
    if lib.drn_dit_forward_args_bytes() != ctypes.sizeof(DitForwardArgs) or lib.drn_dit_sub_bytes() != ctypes.sizeof(DitSub):
        raise RuntimeError("libdrn.so: drn_dit_forward_args / drn_dit_sub layout differs from the ctypes mirror in native.py")
    _LIB = lib
    return lib


def _check(code: int, what: str):
    if code != 0:
        msg = load_library().drn_error_string(code)
        raise RuntimeError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "device tensor required"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype == torch.bfloat16, f"bf16 tensor required, got {t.dtype}"


# ----------------------------------------------------------------------------------------------- kernel timing hook
class KernelTimer:
    """HIP-event timing of individual launches on the launch stream (bench.py's roofline leg).

    torch.cuda.Event records on torch's current stream, which is the stream every wrapper below launches on.
    """

    def __init__(self, names=("gemm", "attention"), sample_every=1):
        """sample_every = n times every n-th launch of each name only (an event pair costs ~35 us of queue time on the
        GPU, ~10 ms per DiT forward if every launch is bracketed); use an n coprime with the launch pattern's period."""
        self.names = set(names)
        self.sample_every = max(1, int(sample_every))
        self.counts = {}
        self.records = []          # (name, start_event, end_event, flops, bytes)
        self._native = None        # drn_timer handle: launches enqueued by drn_dit_forward are bracketed in C

    def native_handle(self, capacity=8192):
        """The event pool drn_dit_forward records into (same sampling rule, kinds 0 = gemm, 1 = attention); None when this
        timer does not watch those kernels."""
        if not ({"gemm", "attention"} <= self.names):
            return None
        if self._native is None:
            h = load_library().drn_timer_create(capacity, self.sample_every)
            if not h:
                raise RuntimeError("drn_timer_create failed")
            self._native = c_void_p(h)
        return self._native

    def __del__(self):
        if getattr(self, "_native", None) is not None and _LIB is not None:
            _LIB.drn_timer_destroy(self._native)
            self._native = None

    def begin(self, name):
        if name not in self.names:
            return None
        c = self.counts.get(name, 0)
        self.counts[name] = c + 1
        if c % self.sample_every:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, name, start, flops, nbytes):
        if start is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((name, start, e, flops, nbytes))

    def summary(self):
        """name -> dict(launches, ms_total, ms_avg, flops, bytes); call after a synchronize."""
        out = {}
        for name, s, e, fl, by in self.records:
            d = out.setdefault(name, {"launches": 0, "ms_total": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms_total"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        seen = dict(self.counts)
        if self._native is not None:
            lib = load_library()
            kind, ms, fl, by = c_int(), c_float(), c_double(), c_double()
            for i in range(lib.drn_timer_count(self._native)):
                _check(lib.drn_timer_read(self._native, i, kind, ms, fl, by), "drn_timer_read")
                name = ("gemm", "attention")[kind.value]
                d = out.setdefault(name, {"launches": 0, "ms_total": 0.0, "flops": 0.0, "bytes": 0.0})
                d["launches"] += 1
                d["ms_total"] += ms.value
                d["flops"] += fl.value
                d["bytes"] += by.value
            for k, name in enumerate(("gemm", "attention")):
                seen[name] = seen.get(name, 0) + lib.drn_timer_seen(self._native, k)
        for name, d in out.items():
            d["ms_avg"] = d["ms_total"] / max(d["launches"], 1)
            d["launches_seen"] = seen.get(name, d["launches"])      # all launches, timed or not
        return out


_TIMER = None


def set_timer(t):
    global _TIMER
    _TIMER = t


# ----------------------------------------------------------------------------------------------- wrappers

def gemm(a, w, out=None, epilogue=EPI_NONE, gate=None, residual=None, rows_per_batch=None, splitk=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T).  a/w/out may be row-strided 2-D views (last dim contiguous)."""
    _bf16(a, w, out, gate, residual)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    ldr = residual.stride(0) if residual is not None else 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
    t0 = _TIMER.begin("gemm") if _TIMER is not None else None
    lib = load_library()
    rpb = rows_per_batch if rows_per_batch else max(M, 1)
    # launch decisions that change the summation order come from ONE clip's rows (batch-invariant results)
    Mb = rpb if (0 < rpb < M and M % rpb == 0) else M
    splits = lib.drn_gemm_splitk_choice(Mb, N, K) if (Mb <= 1024 and splitk is None) else (splitk or 1)
    if splits > 1:
        # few tokens: the product streams the weights; K is split over several workgroups per tile to keep the CUs busy
        nbytes = lib.drn_gemm_splitk_workspace_bytes(M, N, splits)
        key = (a.device, "gemm")
        ws = _SPLIT_WS.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
            _SPLIT_WS[key] = ws
        _check(lib.drn_gemm_bf16_splitk(_ptr(a), _ptr(w), _ptr(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), epilogue,
                                        _ptr(gate), _ptr(residual), ldr, rpb, splits, ws.data_ptr(), _stream()),
               "drn_gemm_bf16_splitk")
    else:
        _check(lib.drn_gemm_bf16(_ptr(a), _ptr(w), _ptr(out), M, N, K, a.stride(0), w.stride(0), out.stride(0),
                                 epilogue, _ptr(gate), _ptr(residual), ldr, rpb, _stream()), "drn_gemm_bf16")
    if t0 is not None:
        _TIMER.end("gemm", t0, 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N * (2 if residual is not None else 1)))
    return out


def gemm_blocked_ok(M, N) -> bool:
    """True when gemm_blocked can run an [M, N] output (every tile kernel but the 128x128 one has the blocked layouts)."""
    return load_library().drn_gemm_tile_choice(M, N) != 0


def gemm_blocked(a, w, out, M, epilogue=EPI_NONE, gate=None, residual=None, a_planes=False, c_planes=False):
    """epi(A @ w^T) with A and / or C stored as planes of columns (drn_gemm_bf16_blocked).
    a_planes: `a` is [P, M, Kb] contiguous = logical A[M, P*Kb] (plane p holds columns p*Kb ..);
    c_planes: `out` is [P, M, Nb] contiguous = logical C[M, P*Nb].  Otherwise plain [M, K] / [M, N] row-strided views."""
    _bf16(a, w, out, gate, residual)
    N, K = w.shape
    assert w.stride(1) == 1
    if a_planes:
        P, Ma, Kb = a.shape
        assert a.is_contiguous() and Ma == M and P * Kb == K
        lda, abc, abs_ = Kb, Kb, M * Kb
    else:
        assert a.shape == (M, K) and a.stride(1) == 1
        lda, abc, abs_ = a.stride(0), 0, 0
    if c_planes:
        P, Mc, Nb = out.shape
        assert out.is_contiguous() and Mc == M and P * Nb == N
        ldc, cbc, cbs = Nb, Nb, M * Nb
    else:
        assert out.shape == (M, N) and out.stride(1) == 1
        ldc, cbc, cbs = out.stride(0), 0, 0
    ldr = residual.stride(0) if residual is not None else 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.stride(1) == 1
    t0 = _TIMER.begin("gemm") if _TIMER is not None else None
    _check(load_library().drn_gemm_bf16_blocked(_ptr(a), _ptr(w), _ptr(out), M, N, K, lda, w.stride(0), ldc, epilogue,
                                                _ptr(gate), _ptr(residual), ldr, max(M, 1), abc, abs_, cbc, cbs, _stream()),
           "drn_gemm_bf16_blocked")
    if t0 is not None:
        _TIMER.end("gemm", t0, 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N * (2 if residual is not None else 1)))
    return out


def gemv(x, w, out=None, add=None, mul=None, act=ACT_NONE):
    """Grouped batch-1 GEMV.  x: [G|1, B, K]; w: [G, N, K]; out/add/mul: [G, B, N] (add/mul may have G == 1)."""
    _bf16(x, w, out, add, mul)
    G, N, K = w.shape
    assert x.dim() == 3 and x.shape[2] == K and x.is_contiguous() and w.is_contiguous()
    B = x.shape[1]
    if out is None:
        out = torch.empty((G, B, N), dtype=torch.bfloat16, device=x.device)
    assert out.shape == (G, B, N) and out.is_contiguous()

    def gs(t, inner):
        if t is None:
            return 0, 0
        assert t.is_contiguous() and t.dim() == 3 and t.shape[1] == B and t.shape[2] == inner
        return (0 if t.shape[0] == 1 else B * inner), inner

    xg, xb = gs(x, K)
    ag, ab = gs(add, N)
    mg, mb = gs(mul, N)
    _check(load_library().drn_gemv_bf16(_ptr(x), _ptr(w), _ptr(out), N, K, G, B, xg, xb, N * K, B * N, N,
                                        _ptr(add), ag, ab, _ptr(mul), mg, mb, act, _stream()), "drn_gemv_bf16")
    return out


def ln_modulate(x, shift, scale, out=None, add_vec=None, rows_per_batch=None, eps=1e-6):
    """out = bf16(bf16(LN(x) * bf16(1+scale)) + shift); if add_vec: x <- bf16(x + add_vec) in place first."""
    _bf16(x, shift, scale, out, add_vec)
    rows, D = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    _check(load_library().drn_ln_modulate(_ptr(x), _ptr(add_vec), _ptr(shift), _ptr(scale), _ptr(out), rows, D,
                                          rows_per_batch if rows_per_batch else max(rows, 1), eps, _stream()),
           "drn_ln_modulate")
    return out


def bcast_add(x, vec, rows_per_batch=None):
    _bf16(x, vec)
    rows, D = x.shape
    assert x.is_contiguous()
    _check(load_library().drn_bcast_add(_ptr(x), _ptr(vec), rows, D, rows_per_batch if rows_per_batch else max(rows, 1),
                                        _stream()), "drn_bcast_add")
    return x


def permute_021(x, out=None):
    """[A, B, C] -> [B, A, C] (contiguous both sides)."""
    _bf16(x, out)
    A, B, C = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty((B, A, C), dtype=torch.bfloat16, device=x.device)
    assert out.is_contiguous() and out.numel() == x.numel()
    _check(load_library().drn_permute_021(_ptr(x), _ptr(out), A, B, C, _stream()), "drn_permute_021")
    return out


def rmsnorm(x, w, eps=1e-6):
    _bf16(x, w)
    rows, D = x.shape
    out = torch.empty_like(x)
    _check(load_library().drn_rmsnorm(_ptr(x), _ptr(w), _ptr(out), rows, D, eps, _stream()), "drn_rmsnorm")
    return out


def qk_norm_rope(q, k, wq, wk, cos, sin, heads, tokens_per_batch=None, pos_offset=0, eps=1e-6):
    """In place on q and/or k: [tokens, heads*128] views with their own row strides; either may be None."""
    _bf16(q, k, wq, wk, cos, sin)
    ref = q if q is not None else k
    tokens = ref.shape[0]
    for t in (q, k):
        if t is not None:
            assert t.shape == (tokens, heads * 128) and t.stride(1) == 1
    _check(load_library().drn_qk_norm_rope(_ptr(q), _ptr(k), _ptr(wq), _ptr(wk), _ptr(cos), _ptr(sin), tokens, heads,
                                           q.stride(0) if q is not None else 0, k.stride(0) if k is not None else 0,
                                           tokens_per_batch if tokens_per_batch else max(tokens, 1),
                                           pos_offset, eps, _stream()), "drn_qk_norm_rope")


_NUM_CUS = 256          # MI355X
_SPLIT_WS = {}


def attention_plan(batch, heads, Sq, Sk):
    """How to cover the (q-block, head) grid of ONE clip with whole rounds of the 256 CUs: a list of (q_begin, q_end, kv_splits)
    from drn_attention_plan (csrc/dit_forward.hip: the same plan drn_dit_forward uses).  The q-blocks that fill whole rounds run
    unsplit; a fractional last round runs as a second launch with its keys cut into chunks (split-KV + combine).  `batch` is
    ignored on purpose: the plan of one clip applies to every clip of a batch (batch-invariant summation order)."""
    buf = (c_int64 * 6)()
    n = load_library().drn_attention_plan(heads, Sq, Sk, buf)
    if n <= 0:
        raise ValueError(f"no attention plan for heads={heads} Sq={Sq} Sk={Sk}")
    return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(n)]


def attention(q, k, v, out=None, heads=None, scale=None, kv_splits=None):
    """q: [B, Sq, H*128], k/v: [B, Sk, H*128] (token-strided views allowed) -> out [B, Sq, H*128].
    kv_splits: None = automatic (attention_plan), 1 = single pass, n > 1 = split-KV + combine."""
    _bf16(q, k, v, out)
    B, Sq, HD = q.shape
    Sk = k.shape[1]
    H = heads if heads else HD // 128
    assert HD == H * 128 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    if out is None:
        out = torch.empty((B, Sq, HD), dtype=torch.bfloat16, device=q.device)
    if scale is None:
        scale = 1.0 / (128 ** 0.5)
    # the plan of ONE clip, applied to every clip of the batch: a split of the keys changes the summation order, so it must not
    # depend on how many clips are stepped together
    plan = attention_plan(1, H, Sq, Sk) if kv_splits is None else [(0, Sq, int(kv_splits))]
    t0 = _TIMER.begin("attention") if _TIMER is not None else None
    lib = load_library()
    for q0, q1, ns in plan:
        qs, os_, n = q[:, q0:q1], out[:, q0:q1], q1 - q0
        if ns > 1:
            nbytes = lib.drn_attention_splitkv_workspace_bytes(B, H, n, ns)
            key = (q.device, nbytes)
            ws = _SPLIT_WS.get(key)
            if ws is None:
                _SPLIT_WS.clear()
                ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
                _SPLIT_WS[key] = ws
            _check(lib.drn_attention_splitkv_bf16(_ptr(qs), _ptr(k), _ptr(v), _ptr(os_), B, H, n, Sk,
                                                  q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                                  q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, ns,
                                                  ws.data_ptr(), _stream()), "drn_attention_splitkv_bf16")
        else:
            _check(lib.drn_attention_bf16(_ptr(qs), _ptr(k), _ptr(v), _ptr(os_), B, H, n, Sk,
                                          q.stride(1), k.stride(1), v.stride(1), out.stride(1),
                                          q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, _stream()),
                   "drn_attention_bf16")
    if t0 is not None:
        _TIMER.end("attention", t0, 4.0 * B * H * Sq * Sk * 128, 2.0 * B * H * 128 * (2 * Sq + 2 * Sk))
    return out


def patchify_concat(x, cond, with_mask, pt, ps, ldo):
    _bf16(x, cond)
    B, Cx, T, H, W = x.shape
    Cc = cond.shape[1] if cond is not None else 0
    assert x.is_contiguous() and (cond is None or cond.is_contiguous())
    rows = B * (T // pt) * (H // ps) * (W // ps)
    out = torch.empty((rows, ldo), dtype=torch.bfloat16, device=x.device)
    _check(load_library().drn_patchify_concat(_ptr(x), _ptr(cond), _ptr(out), B, Cx, Cc, 1 if with_mask else 0, T, H, W,
                                              pt, ps, ldo, _stream()), "drn_patchify_concat")
    return out


def unpatchify(y, B, C, Tp, Hp, Wp, pt, ps):
    _bf16(y)
    assert y.stride(1) == 1
    out = torch.empty((B, C, Tp * pt, Hp * ps, Wp * ps), dtype=torch.bfloat16, device=y.device)
    _check(load_library().drn_unpatchify(_ptr(y), y.stride(0), _ptr(out), B, C, Tp, Hp, Wp, pt, ps, _stream()),
           "drn_unpatchify")
    return out


def edm_scale_input(x, c_in: float):
    _bf16(x)
    assert x.is_contiguous()
    out = torch.empty_like(x)
    _check(load_library().drn_edm_scale_input(_ptr(x), _ptr(out), x.numel(), c_in, _stream()), "drn_edm_scale_input")
    return out


def edm_step(model_out, sample, c_skip: float, c_out: float, sigma: float, dt: float):
    _bf16(model_out, sample)
    assert model_out.is_contiguous() and sample.is_contiguous() and model_out.numel() == sample.numel()
    out = torch.empty_like(sample)
    _check(load_library().drn_edm_step(_ptr(model_out), _ptr(sample), _ptr(out), sample.numel(), c_skip, c_out, sigma, dt,
                                       _stream()), "drn_edm_step")
    return out


def cfg_combine(cond, uncond, guidance: float):
    _bf16(cond, uncond)
    assert cond.is_contiguous() and uncond.is_contiguous()
    out = torch.empty_like(cond)
    _check(load_library().drn_cfg_combine(_ptr(cond), _ptr(uncond), _ptr(out), cond.numel(), guidance, _stream()),
           "drn_cfg_combine")
    return out


def postprocess_u8(video, normalize_normal: bool):
    """video [B,3,T,H,W] bf16 -> uint8 [B,T,H,W,3]."""
    _bf16(video)
    B, C, T, H, W = video.shape
    assert C == 3 and video.is_contiguous()
    out = torch.empty((B, T, H, W, 3), dtype=torch.uint8, device=video.device)
    _check(load_library().drn_postprocess_u8(_ptr(video), _ptr(out), B, T, H, W, 1 if normalize_normal else 0, _stream()),
           "drn_postprocess_u8")
    return out


def dit_forward(args: "DitForwardArgs"):
    """Enqueue one whole DiT forward (drn_dit_forward).  `args` holds raw pointers: the caller keeps every tensor alive."""
    args.struct_bytes = ctypes.sizeof(DitForwardArgs)
    if _TIMER is not None:
        h = _TIMER.native_handle()
        args.timer = h.value if h is not None else None
    else:
        args.timer = None
    _check(load_library().drn_dit_forward(ctypes.byref(args), _stream()), "drn_dit_forward")
