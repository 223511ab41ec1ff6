"""ctypes wrappers for the tokenizer entry points of include/drn.h (extends native.SIGNATURES)."""
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

from . import native as N

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

N.SIGNATURES.update({
    "drn_conv3d_igemm": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                         _L, _L, _I, _F, _P],
    "drn_conv_force_tile": [_I],
    "drn_conv_last_tile": [],
    "drn_groupnorm_silu": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "drn_groupnorm_workspace_bytes": [_I],
    "drn_groupnorm_stats": [_P, _P, _I, _I, _I, _I, _I, _P],
    "drn_groupnorm_apply": [_P, _P, _I, _F, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "drn_haar_patch": [_P, _P, _I, _I, _I, _I, _I, _P],
    "drn_haar_unpatch": [_P, _P, _I, _I, _I, _I, _I, _P],
    "drn_resample": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "drn_softmax_rows": [_P, _P, _L, _I, _L, _L, _P],
    "drn_softmax_rows_scaled": [_P, _P, _L, _I, _L, _L, _F, _P],
    "drn_transpose_bf16": [_P, _P, _I, _I, _L, _L, _P],
    "drn_temporal_attention": [_P, _P, _P, _P, _I, _L, _I, _F, _P],
    "drn_planar_to_cl": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "drn_cl_to_planar": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
})
N._RESTYPES["drn_groupnorm_workspace_bytes"] = c_int64

BF = torch.bfloat16


# Activation buffers whose halo is known to be zero, by (shape, device): every kernel writes interior pixels only, so a buffer that
# was zero-filled once keeps a zero halo for life and the next activation of that shape takes it over as it is.  Before round 3
# every activation was a fresh torch.zeros: 435 fills of up to 99 MB per encode + decode at the headline clip (6 ms of 46).
_POOL = {}
_POOL_ON = [False]          # on inside HipCosmosTokenizer.encode / decode only: nothing there keeps `.t` of a dead activation


def pool_clear():
    _POOL.clear()


class pooled_buffers:
    """with pooled_buffers(): activations allocated inside recycle zero-halo buffers of their shape."""

    def __enter__(self):
        self.prev = _POOL_ON[0]
        _POOL_ON[0] = True

    def __exit__(self, *exc):
        _POOL_ON[0] = self.prev
        return False


class CL:
    """Channels-last activation [T][H+2*halo][W+2*halo][C] bf16 with a zero halo (interior-only writes keep it zero).
    Buffers allocated here go back to a per-shape pool when the activation dies (same stream: the next writer is ordered behind
    the last reader); `dirty_halo()` takes a buffer out of that cycle (the band-sharded tokenizer writes neighbour rows into it)."""

    __slots__ = ("t", "T", "H", "W", "C", "halo", "_key")

    def __init__(self, T, H, W, C, halo=1, device=None, tensor=None, live=None):
        """live: how many of the C stored channels the producing kernel writes (16-channel latents live in 64-channel buffers
        whose tail must stay zero): buffers are only recycled between activations with the same zero regions."""
        self.T, self.H, self.W, self.C, self.halo = T, H, W, C, halo
        self._key = None
        if tensor is None:
            shape = (T, H + 2 * halo, W + 2 * halo, C)
            if halo and _POOL_ON[0]:
                key = (shape, C if live is None else int(live), torch.device(device) if device is not None else None)
                free = _POOL.get(key)
                tensor = free.pop() if free else torch.zeros(shape, dtype=BF, device=device)
                self._key = key
            elif halo or (live is not None and live < C):
                tensor = torch.zeros(shape, dtype=BF, device=device)
            else:
                tensor = torch.empty(shape, dtype=BF, device=device)      # no halo: every element is written
        self.t = tensor

    def dirty_halo(self):
        self._key = None

    def __del__(self):
        key = getattr(self, "_key", None)
        if key is not None and _POOL is not None:
            _POOL.setdefault(key, []).append(self.t)

    def interior(self):
        h = self.halo
        return self.t[:, h:h + self.H, h:h + self.W, :] if h else self.t


def conv3d(x: CL, w, bias, N_out, k, stride=(1, 1, 1), pad=0, t_off=None, out: CL = None, residual: CL = None,
           out_halo=1, out_dims=None, out_channels_stored=None):
    """Causal conv as implicit GEMM.  w: repacked [N_out, kT*kH*kW*C]; returns CL of [To,Ho,Wo,N_out (stored ldc)]."""
    kT, kH, kW = k
    sT, sH, sW = stride
    if t_off is None:
        t_off = (kT - 1) + (1 - sT)
    if out_dims is None:
        To = (x.T + t_off - kT) // sT + 1 if (kT > 1 or sT > 1) else x.T
        Ho = (x.H + 2 * pad - kH) // sH + 1
        Wo = (x.W + 2 * pad - kW) // sW + 1
    else:
        To, Ho, Wo = out_dims
    if out is None:
        out = CL(To, Ho, Wo, out_channels_stored or N_out, out_halo, x.t.device, live=N_out)
    assert (out.T, out.H, out.W) == (To, Ho, Wo) and out.C >= N_out
    assert w.shape == (N_out, kT * kH * kW * x.C), (w.shape, N_out, k, x.C)
    if residual is not None:
        assert (residual.T, residual.H, residual.W, residual.halo) == (To, Ho, Wo, out.halo)
    t0 = N._TIMER.begin("conv") if N._TIMER is not None else None
    if t0 is not None:
        M = To * Ho * Wo
        taps = kT * kH * kW
        # algorithmic bytes: each input / output / residual element once + the weights (im2col reuse is on-chip)
        nbytes = 2.0 * (x.T * x.H * x.W * x.C + M * N_out * (2 if residual is not None else 1) + N_out * taps * x.C)
        N._TIMER.end_later = (t0, 2.0 * M * N_out * taps * x.C, nbytes)
    N._check(N.load_library().drn_conv3d_igemm(
        x.t.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.t.data_ptr(),
        residual.t.data_ptr() if residual is not None else None, x.T, x.H, x.W, x.C, x.halo, N_out, kT, kH, kW, sT, sH, sW,
        pad, t_off, To, Ho, Wo, out.halo, out.C, residual.C if residual is not None else 0, 0, 1.0, N._stream()),
        "drn_conv3d_igemm")
    if t0 is not None:
        _, fl, by = N._TIMER.end_later
        N._TIMER.end("conv", t0, fl, by)
    return out


def dense_gemm(a, w, out=None, out_f32=False, alpha=1.0, bias=None):
    """out[M,N] = alpha * (a[M,K] @ w[N,K]^T (+bias)) through the implicit-GEMM kernel (compact 1x1x1 case)."""
    M, K = a.shape
    Nn = w.shape[0]
    assert a.is_contiguous() and w.is_contiguous() and w.shape[1] == K and K % 64 == 0 and Nn % 4 == 0
    if out is None:
        out = torch.empty((M, Nn), dtype=torch.float32 if out_f32 else BF, device=a.device)
    assert out.is_contiguous()
    N._check(N.load_library().drn_conv3d_igemm(
        a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), None,
        1, 1, M, K, 0, Nn, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, M, 0, out.shape[1], 0, 1 if out_f32 else 0, alpha, N._stream()),
        "drn_conv3d_igemm(dense)")
    return out


_GN_WS = {}


def groupnorm_silu(x: CL, gamma, beta, silu=True, out: CL = None, eps=1e-6):
    if out is None:
        out = CL(x.T, x.H, x.W, x.C, x.halo, x.t.device)
    lib = N.load_library()
    key = (x.t.device, x.T)
    ws = _GN_WS.get(key)
    if ws is None:
        ws = torch.empty(lib.drn_groupnorm_workspace_bytes(x.T), dtype=torch.uint8, device=x.t.device)
        _GN_WS[key] = ws
    N._check(lib.drn_groupnorm_silu(x.t.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.t.data_ptr(), ws.data_ptr(),
                                    x.T, x.H, x.W, x.C, x.halo, eps, 1 if silu else 0, N._stream()), "drn_groupnorm_silu")
    return out


def groupnorm_stats(x: CL) -> torch.Tensor:
    """Partial sums of this rank's rows: [T, 64, 2] fp64 (sum, sum of squares); the halo of x must be zero."""
    part = torch.empty((x.T, 64, 2), dtype=torch.float64, device=x.t.device)
    N._check(N.load_library().drn_groupnorm_stats(x.t.data_ptr(), part.data_ptr(), x.T, x.H, x.W, x.C, x.halo, N._stream()),
             "drn_groupnorm_stats")
    return part


def groupnorm_apply(x: CL, part: torch.Tensor, count: float, gamma, beta, silu=True, eps=1e-6) -> CL:
    """Normalise this rank's rows with the statistics of the whole frame: part [T, nparts, 2] fp64, count elements per frame."""
    assert part.dtype == torch.float64 and part.is_contiguous() and part.shape[0] == x.T and part.shape[2] == 2
    out = CL(x.T, x.H, x.W, x.C, x.halo, x.t.device)
    N._check(N.load_library().drn_groupnorm_apply(x.t.data_ptr(), part.data_ptr(), part.shape[1], float(count), gamma.data_ptr(),
                                                  beta.data_ptr(), out.t.data_ptr(), x.T, x.H, x.W, x.C, x.halo, eps,
                                                  1 if silu else 0, N._stream()), "drn_groupnorm_apply")
    return out


def haar_patch(video, halo=1) -> CL:
    """video [C,T,H,W] bf16 planar -> CL [(T+3)/4, H/4, W/4, 64*C]."""
    C, T, H, W = video.shape
    assert video.is_contiguous() and video.dtype == BF
    out = CL((T + 3) // 4, H // 4, W // 4, 64 * C, halo, video.device)
    N._check(N.load_library().drn_haar_patch(video.data_ptr(), out.t.data_ptr(), C, T, H, W, halo, N._stream()), "drn_haar_patch")
    return out


def haar_unpatch(x: CL, Cimg=3):
    assert x.C == 64 * Cimg
    video = torch.empty((Cimg, 4 * x.T - 3, 4 * x.H, 4 * x.W), dtype=BF, device=x.t.device)
    N._check(N.load_library().drn_haar_unpatch(x.t.data_ptr(), video.data_ptr(), Cimg, x.T, x.H, x.W, x.halo, N._stream()),
             "drn_haar_unpatch")
    return video


def resample(x: CL, mode: int) -> CL:
    T, H, W = x.T, x.H, x.W
    To, Ho, Wo = {0: (T, H // 2, W // 2), 1: ((T + 1) // 2, H, W), 2: (2 * T - 1 if T > 1 else 1, H, W),
                  3: (T, 2 * H, 2 * W)}[mode]
    out = CL(To, Ho, Wo, x.C, x.halo, x.t.device)
    N._check(N.load_library().drn_resample(x.t.data_ptr(), out.t.data_ptr(), mode, T, H, W, x.C, To, Ho, Wo, x.halo,
                                           N._stream()), "drn_resample")
    return out


def softmax_rows(scores, n, ldp, scale=1.0):
    """bf16 softmax(scale * scores[:, :n]) per row, zero-filled up to ldp columns."""
    rows = scores.shape[0]
    probs = torch.empty((rows, ldp), dtype=BF, device=scores.device)
    N._check(N.load_library().drn_softmax_rows_scaled(scores.data_ptr(), probs.data_ptr(), rows, n, scores.stride(0), ldp,
                                                      float(scale), N._stream()), "drn_softmax_rows_scaled")
    return probs


_SCORES_TILE = os.environ.get("DRN_VAE_SCORES_TILE", "1") != "0"      # 0: always the implicit-GEMM kernel (A/B runs)


def scores_f32(q, k):
    """fp32 q[M,K] @ k[N,K]^T as it leaves the accumulators: the 256 x 256 tile kernel when the shape allows (M, N % 256 == 0,
    K % 64 == 0), else the implicit-GEMM kernel - same K order per element, same bits."""
    M, K = q.shape
    Nn = k.shape[0]
    if _SCORES_TILE and M % 256 == 0 and Nn % 256 == 0 and K % 64 == 0 and q.is_contiguous() and k.is_contiguous():
        out = torch.empty((M, Nn), dtype=torch.float32, device=q.device)
        N._check(N.load_library().drn_gemm_bf16_f32out(q.data_ptr(), k.data_ptr(), out.data_ptr(), M, Nn, K, K, K, N._stream()),
                 "drn_gemm_bf16_f32out")
        return out
    return dense_gemm(q, k, out_f32=True)


def transpose(x, ldo):
    rows, cols = x.shape
    y = torch.empty((cols, ldo), dtype=BF, device=x.device)
    N._check(N.load_library().drn_transpose_bf16(x.data_ptr(), y.data_ptr(), rows, cols, x.stride(0), ldo, N._stream()),
             "drn_transpose_bf16")
    return y


def temporal_attention(q, k, v, scale):
    """q,k,v: [T, P, C] compact -> [T, P, C]."""
    T, P, C = q.shape
    o = torch.empty_like(q)
    N._check(N.load_library().drn_temporal_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), T, P, C, scale,
                                                     N._stream()), "drn_temporal_attention")
    return o


def planar_to_cl(x, Cs, halo=1) -> CL:
    C, T, H, W = x.shape
    out = CL(T, H, W, Cs, halo, x.device, live=C)
    N._check(N.load_library().drn_planar_to_cl(x.contiguous().data_ptr(), out.t.data_ptr(), C, T, H, W, Cs, halo, N._stream()),
             "drn_planar_to_cl")
    return out


def cl_to_planar(x: CL, C):
    out = torch.empty((C, x.T, x.H, x.W), dtype=BF, device=x.t.device)
    N._check(N.load_library().drn_cl_to_planar(x.t.data_ptr(), out.data_ptr(), C, x.T, x.H, x.W, x.C, x.halo, N._stream()),
             "drn_cl_to_planar")
    return out
