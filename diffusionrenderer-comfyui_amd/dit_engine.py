"""HipDiT: the CleanDiffusionRendererGeneralDIT forward on hand-written gfx950 kernels.

Host-side mirror of the reference operator `net(x, timesteps, latent_condition, context_index)`
(CleanGeneralDIT.py:731-751 -> :656-718): same call signature, same state-dict names in, same
[B,16,F,h,w] tensor out.  All arithmetic on activations runs in libdrn.so (native.py); torch is used
for device memory, tiny host tables and (multi-GPU) the RCCL all-gather.

Data layout in HBM: tokens are rows.  X [S, D] bf16 (S = F*(h/2)*(w/2) tokens in (T H W) order, the
reference's 'B T H W D -> (T H W) B D' with B = 1), QKV [S, 3D] (q | k | v, head-major 128-wide slices),
MLP hidden [S, 4D].  Weights are repacked once at load: q/k/v fused to [3D, D]; patch-embed K padded to a
multiple of 64; final projection N padded to 128; all 3*L AdaLN-LoRA down/up projections stacked for
two grouped GEMV launches per timestep.

Exact shortcuts: the cross-attention has ONE key, so softmax == 1 and its output is
to_out(to_v(context)) for every token, independent of x; its LayerNorm/modulate/q-projection are dead.
The block reduces to x += bf16(gate * c_i) with c_i cached per context index, and that broadcast add is
fused into the next sub-block's LayerNorm pass.

Sequence parallelism (one process per GPU): rank r owns a contiguous band of latent frames; every op is
token-local except self-attention.  There the ranks trade token bands for heads with one all-to-all (each rank
attends over ALL tokens of heads/world heads) and trade back afterwards; when the world size does not divide the
head count, K/V rows (after RMSNorm + RoPE) are all-gathered instead (parallel.py).
"""
from typing import Dict, Optional

import torch

from . import native as N
from .host_tables import rope_cos_sin, timestep_sinusoid
from .parallel import ShardPlan, allgather_rows_, alltoall_bands_, alltoall_rows_, group_info, wait_exchange


def _pad_cols(w: torch.Tensor, mult: int) -> torch.Tensor:
    n, k = w.shape
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w.contiguous()
    out = torch.zeros((n, kp), dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out


def _pad_rows(w: torch.Tensor, mult: int) -> torch.Tensor:
    n, k = w.shape
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w.contiguous()
    out = torch.zeros((npad, k), dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


class HipDiT:
    def __init__(self, net: dict, state_dict: Dict[str, torch.Tensor], device=None, prefix: str = "net.",
                 process_group=None):
        self.net = dict(net)
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.D = net["model_channels"]
        self.heads = net["num_heads"]
        self.L = net["num_blocks"]
        self.kinds = [k.strip().lower() for k in net["block_config"].split("-")]
        self.pt, self.ps = net["patch_temporal"], net["patch_spatial"]
        self.out_ch = net["out_channels"]
        self.use_ctx = net.get("use_context_embedding", True)
        self.ctx_dim = net["crossattn_emb_channels"]
        self.with_mask = net.get("concat_padding_mask", True)
        if self.D // self.heads != 128:
            raise ValueError("HipDiT kernels are specialised for head_dim 128 (the renderer's only configuration)")
        N.load_library()
        self.pg = process_group
        import os
        world = group_info(process_group)[1] if process_group is not None else 1
        mode = os.environ.get("DRN_SP_EXCHANGE", "auto")
        if mode not in ("auto", "a2a", "gather"):
            raise ValueError("DRN_SP_EXCHANGE must be auto, a2a or gather")
        if mode == "a2a" and self.heads % world:
            raise ValueError(f"head all-to-all needs the world size ({world}) to divide the head count ({self.heads})")
        if process_group is None:
            self.exchange = "none"
        elif mode == "auto":
            self.exchange = "none" if world == 1 else ("gather" if self.heads % world else "a2a")
        else:
            self.exchange = mode               # explicit: also honoured by a 1-rank group (exercises the RCCL calls on one GPU)
        self.world = world
        self._load(state_dict, prefix)
        self._rope_cache = {}
        self._time_cache = {}
        self._ctx_cache = {}
        self._ws = {}
        self._graphs = {}
        self.trace = None          # tests: dict filled with per-sub-block activations "block{i}.{j}" -> [S, D]
        # DRN_PER_LAUNCH=1: one ctypes call per kernel (the path the sharded engine and the traces use) instead of the
        # drn_dit_forward sequencer - same kernels, same bits (tests compare the two)
        self._per_launch = os.environ.get("DRN_PER_LAUNCH", "0") == "1"
        # DRN_SP_SPLIT_RETURN=0: the return all-to-all as ONE collective after the whole attention (A/B runs)
        self._split_return = os.environ.get("DRN_SP_SPLIT_RETURN", "1") != "0"

    # ------------------------------------------------------------------ weights
    def _load(self, sd, p):
        dev, bf = self.device, torch.bfloat16

        def g(name):
            return sd[p + name].to(device=dev, dtype=bf)

        self.w_patch = _pad_cols(g("x_embedder.proj.1.weight"), 64)
        self.kpad = self.w_patch.shape[1]
        self.w_t1 = g("t_embedder.1.linear_1.weight").contiguous().unsqueeze(0)
        self.w_t2 = g("t_embedder.1.linear_2.weight").contiguous().unsqueeze(0)
        self.w_affnorm = g("affline_norm.weight").contiguous()
        self.seq = sd[p + "pos_embedder.seq"]
        self.ctx_table = g("context_embedding.weight") if self.use_ctx else None
        self.w_final = _pad_rows(g("final_layer.linear.weight"), 128)
        self.final_cols = self.out_ch * self.ps * self.ps * self.pt
        self.w_fa1 = g("final_layer.adaLN_modulation.1.weight").contiguous().unsqueeze(0)
        self.w_fa2 = g("final_layer.adaLN_modulation.2.weight").contiguous().unsqueeze(0)

        a1, a2 = [], []
        self.blocks = []
        ca_v, ca_o = [], []
        for i in range(self.L):
            subs = []
            for j, kind in enumerate(self.kinds):
                q = f"blocks.block{i}.blocks.{j}."
                a1.append(g(q + "adaLN_modulation.1.weight"))
                a2.append(g(q + "adaLN_modulation.2.weight"))
                if kind == "fa":
                    a = q + "block.attn."
                    wq, wk, wv = g(a + "to_q.0.weight"), g(a + "to_k.0.weight"), g(a + "to_v.0.weight")
                    if self.exchange == "a2a":
                        # K|V output columns grouped by the rank that will own the heads: [rank][k | v][heads/world * 128];
                        # q rows are rank-major as they are (a rank's heads are contiguous).  Kept as [q ; k|v] row blocks.
                        W = self.D // self.world
                        wkv = torch.stack([wk.view(self.world, W, -1), wv.view(self.world, W, -1)], 1).reshape(2 * self.D, -1)
                        wqkv = torch.cat([wq, wkv], 0).contiguous()
                    else:
                        wqkv = torch.cat([wq, wk, wv], 0).contiguous()
                    subs.append({"kind": "fa", "wqkv": wqkv,
                                 "qn": g(a + "to_q.1.weight").contiguous(), "kn": g(a + "to_k.1.weight").contiguous(),
                                 "wo": g(a + "to_out.0.weight").contiguous()})
                elif kind == "ca":
                    a = q + "block.attn."
                    subs.append({"kind": "ca", "idx": len(ca_v)})
                    ca_v.append(g(a + "to_v.0.weight"))
                    ca_o.append(g(a + "to_out.0.weight"))
                else:
                    subs.append({"kind": "mlp", "w1": g(q + "block.layer1.weight").contiguous(),
                                 "w2": g(q + "block.layer2.weight").contiguous()})
            self.blocks.append(subs)
        self.n_sites = len(a1)
        self.r = a1[0].shape[0]
        self.w_a1 = torch.cat(a1, 0).contiguous().unsqueeze(0)          # [1, sites*r, D]
        self.w_a2 = torch.stack(a2, 0).contiguous()                     # [sites, 3D, r]
        self.n_ca = len(ca_v)
        self.w_cav = torch.stack(ca_v, 0).contiguous() if ca_v else None   # [n_ca, D, ctx]
        self.w_cao = torch.stack(ca_o, 0).contiguous() if ca_o else None   # [n_ca, D, D]
        # site index of every cross-attention sub-block (for its gate)
        self.ca_sites = [i * len(self.kinds) + j for i in range(self.L) for j, k in enumerate(self.kinds) if k == "ca"]
        # the same sub-block list as the host table drn_dit_forward walks (weights never move after load)
        flat = [sb for subs in self.blocks for sb in subs]
        self._subs_c = (N.DitSub * len(flat))()
        for site, sb in enumerate(flat):
            e = self._subs_c[site]
            e.site, e.ca_index = site, -1
            if sb["kind"] == "fa":
                e.kind, e.w_a, e.w_b = N.SUB_FA, sb["wqkv"].data_ptr(), sb["wo"].data_ptr()
                e.qn, e.kn = sb["qn"].data_ptr(), sb["kn"].data_ptr()
            elif sb["kind"] == "ca":
                e.kind, e.ca_index = N.SUB_CA, sb["idx"]
            else:
                e.kind, e.w_a, e.w_b = N.SUB_MLP, sb["w1"].data_ptr(), sb["w2"].data_ptr()

    # ------------------------------------------------------------------ per-timestep vectors (K10, K11)
    def prepare_timesteps(self, sigmas) -> None:
        """AdaLN vectors of a whole sigma schedule in ONE batched pass (host table -> one H2D copy -> batched GEMVs).

        The sampler knows every sigma before its loop starts; computing their vectors up front keeps host->device copies
        (which block the host until the stream drains) out of the denoising loop, so kernel launches run ahead of the GPU.
        Same arithmetic per sigma as the reference evaluates inside every forward (CleanGeneralDIT.py:664-666, :500-505)."""
        todo = []
        for s_ in sigmas:
            k = float(s_)
            if k not in self._time_cache and k not in todo:
                todo.append(k)
        if not todo:
            return
        D, B = self.D, len(todo)
        t_emb = torch.cat([timestep_sinusoid(k, D) for k in todo], 0).to(self.device)        # [B, D]
        x = t_emb.view(1, B, D)
        h1 = N.gemv(x, self.w_t1)                                                # linear_1
        lora = N.gemv(h1, self.w_t2, act=N.ACT_SILU)                             # linear_2(silu(.))   [1,B,3D]
        emb = N.rmsnorm(t_emb, self.w_affnorm).view(1, B, D)                     # affline_norm
        a = N.gemv(emb, self.w_a1, act=N.ACT_SILU)                               # [1,B,sites*r]
        a = a.view(B, self.n_sites, self.r).permute(1, 0, 2).contiguous()        # [sites,B,r]
        mod = N.gemv(a, self.w_a2, add=lora)                                     # [sites,B,3D]
        af = N.gemv(emb, self.w_fa1, act=N.ACT_SILU)
        modf = N.gemv(af, self.w_fa2, add=lora[:, :, : 2 * D].contiguous())      # [1,B,2D]
        if len(self._time_cache) + B > 512:
            self._time_cache.clear()
        for i, k in enumerate(todo):
            self._time_cache[k] = (mod[:, i, :], modf[0, i])                     # rows stay contiguous: [sites][3D], [2D]

    def time_vectors(self, sigma: float):
        """AdaLN vectors for one sigma: mod [sites, 3D] (shift|scale|gate), final [2D].  Cached per sigma."""
        key = float(sigma)
        hit = self._time_cache.get(key)
        if hit is None:
            self.prepare_timesteps([key])
            hit = self._time_cache[key]
        return hit

    def context_vectors(self, context_index) -> Optional[torch.Tensor]:
        """c_i = to_out_i(to_v_i(ctx)) for every cross-attention block: [n_ca, D].  Cached per index (F8)."""
        if self.n_ca == 0:
            return None
        key = int(context_index) if self.use_ctx else -1
        hit = self._ctx_cache.get(key)
        if hit is not None:
            return hit
        if self.use_ctx:
            ctx = self.ctx_table[key].view(1, 1, self.ctx_dim).contiguous()
        else:
            ctx = torch.zeros((1, 1, self.ctx_dim), dtype=torch.bfloat16, device=self.device)
        v = N.gemv(ctx, self.w_cav)                         # [n_ca,1,D]  to_v (value norm is Identity)
        c = N.gemv(v, self.w_cao).view(self.n_ca, self.D)   # to_out
        self._ctx_cache[key] = c
        return c

    def rope(self, Tp, Hp, Wp):
        key = (Tp, Hp, Wp)
        hit = self._rope_cache.get(key)
        if hit is None:
            cos, sin = rope_cos_sin(Tp, Hp, Wp, 128, self.seq, torch.bfloat16)
            hit = (cos.to(self.device), sin.to(self.device))
            self._rope_cache[key] = hit
        return hit

    def _workspace(self, S, rows, B=1):
        """Activation buffers for `rows` local tokens of S total, B clips stacked along the rows (one shape kept resident)."""
        key = (S, rows, B)
        ws = self._ws.get(key)
        if ws is None:
            D, dev, bf = self.D, self.device, torch.bfloat16
            n = B * rows
            ws = {"x": torch.empty((n, D), dtype=bf, device=dev), "h": torch.empty((n, D), dtype=bf, device=dev),
                  "o": torch.empty((n, D), dtype=bf, device=dev),
                  "u": torch.empty((n, int(D * self.net["mlp_ratio"])), dtype=bf, device=dev),
                  "y": torch.empty((B * S, self.w_final.shape[0]), dtype=bf, device=dev)}
            if self.exchange == "none":
                ws["qkv"] = torch.empty((B * S, 3 * D), dtype=bf, device=dev)      # q | k | v, fused projection
                lib = N.load_library()
                nb = lib.drn_dit_forward_gemm_workspace_bytes(B, S, D, ws["u"].shape[1], self.w_final.shape[0], self.kpad)
                ws["gemm_ws"] = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None      # split-K partials (few tokens)
                nb = lib.drn_dit_forward_attn_workspace_bytes(B, self.heads, S)
                ws["attn_ws"] = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None      # split-KV partials
            elif self.exchange == "a2a":
                W = D // self.world                                                # columns of this rank's heads
                ws["qb"] = torch.empty((rows, D), dtype=bf, device=dev)            # band projections: [rows][rank][W]
                ws["kvb"] = torch.empty((rows, 2 * D), dtype=bf, device=dev)       #                   [rows][rank][k | v][W]
                ws["sq"] = torch.empty((self.world, rows, W), dtype=bf, device=dev)        # send slabs, rank-major
                ws["skv"] = torch.empty((self.world, rows, 2 * W), dtype=bf, device=dev)
                ws["rq"] = torch.empty((S, W), dtype=bf, device=dev)               # all tokens, own heads
                ws["rkv"] = torch.empty((S, 2 * W), dtype=bf, device=dev)
                ws["oh"] = torch.empty((S, W), dtype=bf, device=dev)               # attention output, own heads
                ws["oback"] = torch.empty((self.world, rows, W), dtype=bf, device=dev)
            else:
                ws["q"] = torch.empty((rows, D), dtype=bf, device=dev)             # local queries
                ws["kv"] = torch.empty((S, 2 * D), dtype=bf, device=dev)           # k | v of ALL tokens (all-gathered)
            self._ws = {key: ws}
        return ws

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def __call__(self, x, timesteps, latent_condition, context_index=None, **_):
        return self.forward(x, timesteps, latent_condition, context_index)

    @torch.no_grad()
    def forward(self, x, timesteps, latent_condition, context_index=None):
        """net(x, timesteps, latent_condition, context_index) -> [B, out_ch, F, h, w] (full latent on every rank).

        B > 1 stacks B clips of one shape and one sigma along the token rows (row = b*S + s): the G-buffer passes of one
        clip and the cond / uncond halves of classifier-free guidance are such batches.  Every op is
        row-local except self-attention, which runs per clip (batch stride), so each clip's result is that of its own
        B = 1 forward.  `context_index`: int, list of B ints or a [B, 1] tensor; `latent_condition` [B or 1, ...]."""
        dev, bf = self.device, torch.bfloat16
        x = x.to(device=dev, dtype=bf).contiguous()
        cond = latent_condition.to(device=dev, dtype=bf)
        B, C, F_, h, w = x.shape
        if cond.shape[0] == 1 and B > 1:
            cond = cond.expand(B, *cond.shape[1:])
        cond = cond.contiguous()
        if cond.shape[0] != B or tuple(cond.shape[2:]) != (F_, h, w):
            # the reference fails here too: torch.cat of x and latent_condition (CleanGeneralDIT.py:675)
            raise ValueError(f"latent_condition {tuple(cond.shape)} does not match x {tuple(x.shape)} outside the channel dim")
        if C + cond.shape[1] + (1 if self.with_mask else 0) != self.net["in_channels"] + self.net.get("additional_concat_ch", 16) + (1 if self.with_mask else 0):
            raise ValueError("channel count of x | latent_condition does not match the patch-embed weights")
        if torch.is_tensor(timesteps):
            ts = timesteps.flatten().tolist()
            if any(float(t) != float(ts[0]) for t in ts):
                raise ValueError("all clips of a batch share one sigma (the sampler steps them together)")
            sigma = float(ts[0])
        else:
            sigma = float(timesteps)
        cis = [0] * B
        if self.use_ctx:
            if torch.is_tensor(context_index):
                cis = [int(v) for v in context_index.flatten().tolist()]
            elif isinstance(context_index, (list, tuple)):
                cis = [int(v) for v in context_index]
            else:
                cis = [int(context_index)]
            if len(cis) == 1 and B > 1:
                cis = cis * B
            if len(cis) != B:
                raise ValueError(f"context_index has {len(cis)} entries for a batch of {B}")
        D = self.D
        Tp, Hp, Wp = F_ // self.pt, h // self.ps, w // self.ps
        S = Tp * Hp * Wp
        rank, world = group_info(self.pg) if self.pg is not None else (0, 1)
        plan = ShardPlan(S, rank, world)          # (a sharded batch: every clip is cut into the same token bands, rows = b * band + r)

        mod, modf = self.time_vectors(sigma)
        addvec = None
        if self.n_ca:
            gates = mod[self.ca_sites, 2 * D:]                 # [n_ca, D]
            if B == 1:
                addvec = gates * self.context_vectors(cis[0])  # bf16(gate * c): the whole cross-attention block
            else:
                cv = torch.stack([self.context_vectors(c) for c in cis], 1)       # [n_ca, B, D]
                addvec = (gates.unsqueeze(1) * cv).contiguous()
        if self._graphable(S, world) and B == 1:
            return self._graph_forward(x, cond, mod, modf, addvec, Tp, Hp, Wp)
        return self._run(x, cond, mod, modf, addvec, Tp, Hp, Wp, plan)

    # ------------------------------------------------------------------ hipGraph replay for launch-bound (small) shapes
    def _graphable(self, S, world) -> bool:
        """Replay the ~300 launches of a small-shape forward as one hipGraph (DRN_GRAPHS=1).  Measured: no gain - even at
        S = 256 the eager launches run ahead of the GPU once nothing synchronises inside the denoising loop."""
        import os
        return (self.exchange == "none" and S <= 4096 and self.trace is None and N._TIMER is None
                and os.environ.get("DRN_GRAPHS", "0") == "1")     # opt-in: measured null on MI355X (13.64 vs 13.66 ms at S=256)

    def _graph_forward(self, x, cond, mod, modf, addvec, Tp, Hp, Wp):
        key = (tuple(x.shape), tuple(cond.shape), addvec is not None)
        ent = self._graphs.get(key)
        if ent is None:
            st = {"x": x.clone(), "cond": cond.clone(), "mod": mod.clone(), "modf": modf.clone(),
                  "addvec": addvec.clone() if addvec is not None else None}
            plan = ShardPlan(Tp * Hp * Wp, 0, 1)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                        # warm-up outside capture (workspaces, code objects)
                self._run(st["x"], st["cond"], st["mod"], st["modf"], st["addvec"], Tp, Hp, Wp, plan)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._run(st["x"], st["cond"], st["mod"], st["modf"], st["addvec"], Tp, Hp, Wp, plan)
            ent = (g, st, out)
            if len(self._graphs) >= 4:
                self._graphs.clear()
            self._graphs[key] = ent
        g, st, out = ent
        st["x"].copy_(x)
        st["cond"].copy_(cond)
        st["mod"].copy_(mod)
        st["modf"].copy_(modf)
        if addvec is not None:
            st["addvec"].copy_(addvec)
        g.replay()
        return out.clone()                                       # the graph's output buffer is reused by the next replay

    @staticmethod
    def _traced(X, pending, B):
        if pending is None:
            return X.clone()
        return (X.view(B, -1, X.shape[1]) + pending.view(B, 1, -1)).view_as(X)

    def _fa_sharded(self, sb, Hb, X, O, ws, plan, cos, sin, gate, fused):
        """Self-attention sub-block of ONE clip's token band with an exchange (Hb: modulated input rows, X: residual stream rows,
        updated in place; O: scratch rows)."""
        D = self.D
        S, rows, world = plan.S, plan.rows, plan.world
        if self.exchange == "a2a":
            # tokens -> heads: project the band, regroup rank-major, all-to-all; norm + RoPE + attention over all S
            # tokens of this rank's heads; heads -> tokens: all-to-all back, regroup, output projection.
            # K|V go first so their exchange (RCCL's stream, 2/3 of the bytes) overlaps the Q projection.
            W, hpr = D // world, self.heads // world
            Oh, oback, rq, rkv = ws["oh"], ws["oback"], ws["rq"], ws["rkv"]
            # (the projections write the rank-major send slabs themselves where the tile kernel can; else a regroup pass)
            if fused:
                N.gemm_blocked(Hb, sb["wqkv"][D:], ws["skv"], rows, c_planes=True)
            else:
                N.gemm(Hb, sb["wqkv"][D:], out=ws["kvb"])
                N.permute_021(ws["kvb"].view(rows, world, 2 * W), out=ws["skv"])
            work_kv = alltoall_rows_(ws["skv"], rkv.view(world, rows, 2 * W), self.pg, async_op=True)
            if fused:
                N.gemm_blocked(Hb, sb["wqkv"][:D], ws["sq"], rows, c_planes=True)
            else:
                N.gemm(Hb, sb["wqkv"][:D], out=ws["qb"])
                N.permute_021(ws["qb"].view(rows, world, W), out=ws["sq"])
            work_q = alltoall_rows_(ws["sq"], rq.view(world, rows, W), self.pg, async_op=True)
            wait_exchange(work_kv, "a2a k|v")
            k, v = rkv[:, :W], rkv[:, W:]
            N.qk_norm_rope(None, k, None, sb["kn"], cos, sin, hpr, tokens_per_batch=S)
            wait_exchange(work_q, "a2a q")
            N.qk_norm_rope(rq, None, sb["qn"], None, cos, sin, hpr, tokens_per_batch=S)
            # heads -> tokens.  The attention of a rank's heads is usually two launches (native.attention_plan: the
            # q-blocks that fill whole rounds of the CUs, then the rest with its keys split): the token bands whose
            # queries the first launch has finished go home while the second one runs, only the last bands' slabs
            # (1 of 8 at world 8) travel exposed.  Same launches as the single call: same bits.
            aplan = N.attention_plan(1, hpr, S, S)
            nb = aplan[0][1] // rows if len(aplan) == 2 else 0             # complete bands of the first launch
            if nb >= 1 and self._split_return:
                (_, q_cut, ns0), (_, _, ns1) = aplan
                N.attention(rq[:q_cut].unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0), out=Oh[:q_cut].unsqueeze(0),
                            heads=hpr, kv_splits=ns0)
                work0 = alltoall_bands_(Oh.view(world, rows, W), oback, 0, nb, self.pg, async_op=True)
                N.attention(rq[q_cut:].unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0), out=Oh[q_cut:].unsqueeze(0),
                            heads=hpr, kv_splits=ns1)
                work1 = alltoall_bands_(Oh.view(world, rows, W), oback, nb, world, self.pg, async_op=True)
                wait_exchange(work0, "a2a o (return, under the attention tail)")
                wait_exchange(work1, "a2a o (return)")
            else:
                N.attention(rq.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0), out=Oh.unsqueeze(0), heads=hpr)
                work = alltoall_rows_(Oh.view(world, rows, W), oback, self.pg, async_op=True)
                wait_exchange(work, "a2a o (return)")
            if fused:
                N.gemm_blocked(oback, sb["wo"], X, rows, epilogue=N.EPI_GATE_RES, gate=gate, residual=X, a_planes=True)
                return
            N.permute_021(oback, out=O.view(rows, world, W))
        else:
            # local projections; K|V land directly in this rank's band of the gather buffer.  K|V first, so the
            # exchange (RCCL's own stream) overlaps the Q projection + q-norm; wait() orders attention after it.
            q, KV = ws["q"], ws["kv"]
            kv_loc = plan.band(KV)
            N.gemm(Hb, sb["wqkv"][D:], out=kv_loc)
            N.qk_norm_rope(None, kv_loc[:, :D], None, sb["kn"], cos, sin, self.heads,
                           tokens_per_batch=rows, pos_offset=plan.start)
            work = allgather_rows_(KV, plan, self.pg, async_op=True)     # the one exchange of the block (xGMI)
            N.gemm(Hb, sb["wqkv"][:D], out=q)
            N.qk_norm_rope(q, None, sb["qn"], None, cos, sin, self.heads,
                           tokens_per_batch=rows, pos_offset=plan.start)
            wait_exchange(work, "gather k|v")
            k, v = KV[:, :D], KV[:, D:]
            N.attention(q.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0), out=O.unsqueeze(0), heads=self.heads)
        N.gemm(O, sb["wo"], out=X, epilogue=N.EPI_GATE_RES, gate=gate, residual=X)

    def _run(self, x, cond, mod, modf, addvec, Tp, Hp, Wp, plan):
        """The kernel sequence of one forward (all shapes / pointers fixed for a given input shape -> capturable)."""
        D = self.D
        S, rows, world = plan.S, plan.rows, plan.world
        B = x.shape[0]                                           # B > 1 only without an exchange (rows == S)
        cos, sin = self.rope(Tp, Hp, Wp)
        ws = self._workspace(S, rows, B)
        X, Hb, O, U, Y = ws["x"], ws["h"], ws["o"], ws["u"], ws["y"]
        if B > 1:
            # shift | scale rows per clip for the batched LayerNorm pass: [sites, 2, B, D] (one sigma -> B equal rows)
            modB = mod[:, :2 * D].reshape(-1, 2, 1, D).expand(-1, 2, B, D).contiguous()
            modfB = modf.view(2, 1, D).expand(2, B, D).contiguous()
            gateB = mod[:, 2 * D:].reshape(-1, 1, D).expand(-1, B, D).contiguous()     # one gate row per clip: [sites, B, D]

        # the latent is tiny: every rank patchifies it all and keeps its own token band
        P = N.patchify_concat(x, cond, self.with_mask, self.pt, self.ps, self.kpad)
        if self.exchange == "none" and self.trace is None and not self._per_launch:
            # one GPU: the whole launch sequence below is enqueued by ONE C call (csrc/dit_forward.hip: same kernels, same
            # arguments, same order -> same bits; ~570 ctypes round trips less per forward)
            a = N.DitForwardArgs()
            a.S, a.B, a.D, a.hidden, a.heads = S, B, D, U.shape[1], self.heads
            a.n_sub, a.subs = len(self._subs_c), self._subs_c
            if B == 1:
                a.shift, a.scale, a.gate = mod.data_ptr(), mod.data_ptr() + 2 * D, mod.data_ptr() + 4 * D
                assert mod.stride(1) == 1
                a.shift_site_stride = a.scale_site_stride = a.gate_site_stride = mod.stride(0)    # (rows of a batched sigma table)
                a.final_shift, a.final_scale = modf.data_ptr(), modf.data_ptr() + 2 * D
            else:
                a.shift, a.scale, a.gate = modB.data_ptr(), modB.data_ptr() + 2 * B * D, gateB.data_ptr()
                a.shift_site_stride = a.scale_site_stride = 2 * B * D
                a.gate_site_stride = B * D
                a.final_shift, a.final_scale = modfB.data_ptr(), modfB.data_ptr() + 2 * B * D
            if addvec is not None:
                a.addvec, a.addvec_stride = addvec.data_ptr(), addvec[0].numel()
            a.cos, a.sin = cos.data_ptr(), sin.data_ptr()
            a.P, a.kpad, a.w_patch = P.data_ptr(), self.kpad, self.w_patch.data_ptr()
            a.w_final, a.n_final = self.w_final.data_ptr(), self.w_final.shape[0]
            a.X, a.H, a.QKV, a.O, a.U, a.Y = (t.data_ptr() for t in (X, Hb, ws["qkv"], O, U, Y))
            gws, aws = ws["gemm_ws"], ws["attn_ws"]
            a.gemm_ws, a.gemm_ws_bytes = (gws.data_ptr(), gws.numel()) if gws is not None else (None, 0)
            a.attn_ws, a.attn_ws_bytes = (aws.data_ptr(), aws.numel()) if aws is not None else (None, 0)
            a.eps = 1e-6
            N.dit_forward(a)
            return N.unpatchify(Y, B, self.out_ch, Tp, Hp, Wp, self.pt, self.ps)
        sharded = self.exchange != "none"
        if sharded and B > 1:
            for b in range(B):                                   # this rank's band of every clip (P holds whole clips)
                N.gemm(plan.band(P[b * S:(b + 1) * S]), self.w_patch, out=X[b * rows:(b + 1) * rows])
        else:
            N.gemm(plan.band(P) if B == 1 else P, self.w_patch, out=X, rows_per_batch=rows)

        pending = None
        site = 0
        nk = len(self.kinds)
        fused = None                                             # a2a exchange: blocked-layout GEMMs instead of regroup passes
        for subs in self.blocks:
            for sb in subs:
                m = mod[site]
                shift, scale, gate = m[:D], m[D:2 * D], m[2 * D:]
                if B > 1:
                    shift, scale = modB[site, 0], modB[site, 1]
                if self.trace is not None and site > 0:
                    self.trace[f"block{(site - 1) // nk}.{(site - 1) % nk}"] = self._traced(X, pending, B)
                site += 1
                if sb["kind"] == "ca":
                    if pending is not None:
                        N.bcast_add(X, pending, rows_per_batch=rows)
                    pending = addvec[sb["idx"]]
                    continue
                N.ln_modulate(X, shift, scale, out=Hb, add_vec=pending, rows_per_batch=rows)
                pending = None
                if sb["kind"] == "fa":
                    if self.exchange == "none":
                        QKV = ws["qkv"]
                        N.gemm(Hb, sb["wqkv"], out=QKV, rows_per_batch=rows)
                        q, k, v = QKV[:, :D], QKV[:, D:2 * D], QKV[:, 2 * D:]
                        N.qk_norm_rope(q, k, sb["qn"], sb["kn"], cos, sin, self.heads, tokens_per_batch=S)
                        if B == 1:
                            N.attention(q.unsqueeze(0), k.unsqueeze(0), v.unsqueeze(0), out=O.unsqueeze(0), heads=self.heads)
                        else:
                            Q3 = QKV.view(B, S, 3 * D)
                            N.attention(Q3[:, :, :D], Q3[:, :, D:2 * D], Q3[:, :, 2 * D:], out=O.view(B, S, D), heads=self.heads)
                        N.gemm(O, sb["wo"], out=X, epilogue=N.EPI_GATE_RES, gate=gateB[site - 1] if B > 1 else gate, residual=X,
                               rows_per_batch=rows)
                    else:
                        # sharded: the exchanges (and the projections that write / read their slabs) run clip by clip on this
                        # rank's band of each clip; every clip of a batch has the same sigma, hence the same gate row
                        if fused is None:
                            W_ = D // world
                            fused = (self.exchange == "a2a" and world > 1 and W_ >= 512 and N.gemm_blocked_ok(rows, 2 * D)
                                     and N.gemm_blocked_ok(rows, D))
                        for b in range(B):
                            band = slice(b * rows, (b + 1) * rows)
                            self._fa_sharded(sb, Hb[band], X[band], O[band], ws, plan, cos, sin, gate, fused)
                else:
                    N.gemm(Hb, sb["w1"], out=U, epilogue=N.EPI_GELU, rows_per_batch=rows)
                    N.gemm(U, sb["w2"], out=X, epilogue=N.EPI_GATE_RES, gate=gateB[site - 1] if B > 1 else gate, residual=X,
                           rows_per_batch=rows)

        if self.trace is not None:
            self.trace[f"block{(site - 1) // nk}.{(site - 1) % nk}"] = self._traced(X, pending, B)
        if B == 1:
            N.ln_modulate(X, modf[:D], modf[D:], out=Hb, add_vec=pending)
            N.gemm(Hb, self.w_final, out=plan.band(Y))
            allgather_rows_(Y, plan, self.pg)                       # 2.4 MB at cfg 3: every rank gets the full latent
        elif sharded:
            N.ln_modulate(X, modfB[0], modfB[1], out=Hb, add_vec=pending, rows_per_batch=rows)
            for b in range(B):
                Yb = Y[b * S:(b + 1) * S]
                N.gemm(Hb[b * rows:(b + 1) * rows], self.w_final, out=plan.band(Yb))
                allgather_rows_(Yb, plan, self.pg)
        else:
            N.ln_modulate(X, modfB[0], modfB[1], out=Hb, add_vec=pending, rows_per_batch=rows)
            N.gemm(Hb, self.w_final, out=Y, rows_per_batch=rows)
        return N.unpatchify(Y, B, self.out_ch, Tp, Hp, Wp, self.pt, self.ps)
