"""N > 1 path on CPU: two gloo ranks run the DiT with the token-band decomposition of parallel.py (the same
ShardPlan / all-gather helpers the HIP engine uses) and must reproduce the unsharded oracle.  This checks what the
multi-GPU design rests on: every op but self-attention is token-local, the K/V bands gathered rank-major ARE the global
token order, RoPE rows are offset by the band start, and the gathered output rows unpatchify to the full latent.
"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from conftest import ROOT, tiny_net


_RDZV_N = [0]


def _rdzv():
    """A file:// rendezvous for the spawned ranks: no TCP port is picked ahead of time, so nothing else can take it in between
    (a probed free port was found busy once on a GPU box: EADDRINUSE)."""
    import tempfile
    _RDZV_N[0] += 1
    path = os.path.join(tempfile.gettempdir(), f"drn_rdzv_{os.getpid()}_{_RDZV_N[0]}")
    if os.path.exists(path):
        os.remove(path)
    return "file://" + path


def _sharded_forward(orc, O, par, x, t, cond, ci, group, exchange="gather"):
    """DitOracle.forward restated over one token band (mirrors HipDiT.forward's structure)."""
    rank, world = par.group_info(group)
    dt = orc.dtype
    D = orc.D
    ctx = F.embedding(ci.long(), orc.w("context_embedding.weight"))
    ts = t.to(orc.tables_dtype).flatten()
    t_emb = O.timestep_sinusoid(ts, D).to(dt)
    lora = F.linear(F.silu(F.linear(t_emb, orc.w("t_embedder.1.linear_1.weight"))), orc.w("t_embedder.1.linear_2.weight"))
    emb = O.rms_norm(t_emb, orc.w("affline_norm.weight"))
    xc = torch.cat([x.to(dt), cond.to(dt), torch.ones(1, 1, *x.shape[2:], dtype=dt)], 1)
    patches = O.patchify(xc, 1, 2)
    _, Tp, Hp, Wp, K = patches.shape
    S = Tp * Hp * Wp
    plan = par.ShardPlan(S, rank, world)
    xs = F.linear(plan.band(patches.reshape(S, K)), orc.w("x_embedder.proj.1.weight")).unsqueeze(1)    # [rows,1,D]
    ang = O.rope_angles(Tp, Hp, Wp, orc.dh, orc.w("pos_embedder.seq"), orc.tables_dtype)
    cos, sin = O.rope_cos_sin(ang, orc.tables_dtype)
    cos_all, sin_all = cos.to(dt), sin.to(dt)
    cos, sin = plan.band(cos_all), plan.band(sin_all)                                                  # pos_offset = band start
    cs = ctx.permute(1, 0, 2)
    for i in range(orc.L):
        for j, kind in enumerate(orc.kinds):
            pre = f"blocks.block{i}.blocks.{j}."
            m = F.linear(F.linear(F.silu(emb), orc.w(pre + "adaLN_modulation.1.weight")),
                         orc.w(pre + "adaLN_modulation.2.weight")) + lora
            shift, scale, gate = m.chunk(3, dim=1)
            h = O.modulate(F.layer_norm(xs, (D,), eps=1e-6), shift, scale)
            if kind == "mlp":
                out = orc.mlp(pre + "block.", h)
            elif kind == "ca":
                out = orc.attention(pre + "block.attn.", h, cs, None, None)
            elif exchange == "a2a":
                # HipDiT's head <-> token exchange: rank-major fused projection, all-to-all, norm + RoPE + attention over all
                # S tokens of this rank's heads, all-to-all back, regroup to token-major head order
                a = pre + "block.attn."
                W, hpr, rows = D // world, orc.Hn // world, plan.rows
                wq, wk, wv = (orc.w(a + f"to_{n}.0.weight") for n in "qkv")
                wkv = torch.stack([wk.view(world, W, -1), wv.view(world, W, -1)], 1).reshape(2 * D, -1)
                hb = h.reshape(rows, D)
                skv = F.linear(hb, wkv).view(rows, world, 2 * W).permute(1, 0, 2).contiguous()     # [rows][rank][k|v][W] -> rank-major
                rkv = torch.empty_like(skv)
                par.alltoall_rows_(skv, rkv, group)                                               # exchange 1 (K|V first)
                sq = F.linear(hb, wq).view(rows, world, W).permute(1, 0, 2).contiguous()
                rq = torch.empty_like(sq)
                par.alltoall_rows_(sq, rq, group)                                                 # exchange 2 (Q)
                rkv, rq = rkv.view(S, 2 * W), rq.view(S, W)
                q = rq.reshape(S, 1, hpr, orc.dh)
                k = rkv[:, :W].reshape(S, 1, hpr, orc.dh)
                v = rkv[:, W:].reshape(S, 1, hpr, orc.dh)
                q = O.apply_rope(O.rms_norm(q, orc.w(a + "to_q.1.weight")), cos_all, sin_all)
                k = O.apply_rope(O.rms_norm(k, orc.w(a + "to_k.1.weight")), cos_all, sin_all)
                o = F.scaled_dot_product_attention(q.permute(1, 2, 0, 3), k.permute(1, 2, 0, 3), v.permute(1, 2, 0, 3))
                oh = o.permute(2, 0, 1, 3).reshape(world, rows, W).contiguous()             # [S, W] seen as per-band slabs
                back = torch.empty_like(oh)
                par.alltoall_rows_(oh, back, group)
                o = back.permute(1, 0, 2).reshape(rows, 1, D)                               # drn_permute_021
                out = F.linear(o, orc.w(a + "to_out.0.weight"))
            else:
                a = pre + "block.attn."
                q = F.linear(h, orc.w(a + "to_q.0.weight")).reshape(plan.rows, 1, orc.Hn, orc.dh)
                k = F.linear(h, orc.w(a + "to_k.0.weight")).reshape(plan.rows, 1, orc.Hn, orc.dh)
                v = F.linear(h, orc.w(a + "to_v.0.weight")).reshape(plan.rows, orc.Hn * orc.dh)
                q = O.apply_rope(O.rms_norm(q, orc.w(a + "to_q.1.weight")), cos, sin)
                k = O.apply_rope(O.rms_norm(k, orc.w(a + "to_k.1.weight")), cos, sin)
                kv = torch.zeros(S, 2 * D, dtype=dt)
                plan.band(kv)[:, :D] = k.reshape(plan.rows, D)
                plan.band(kv)[:, D:] = v
                par.allgather_rows_(kv, plan, group)                        # the one exchange of the block
                kf = kv[:, :D].reshape(S, 1, orc.Hn, orc.dh)
                vf = kv[:, D:].reshape(S, 1, orc.Hn, orc.dh)
                o = F.scaled_dot_product_attention(q.permute(1, 2, 0, 3), kf.permute(1, 2, 0, 3), vf.permute(1, 2, 0, 3))
                o = o.permute(2, 0, 1, 3).reshape(plan.rows, 1, D)
                out = F.linear(o, orc.w(a + "to_out.0.weight"))
            xs = xs + gate.unsqueeze(0) * out
    m = F.linear(F.linear(F.silu(emb), orc.w("final_layer.adaLN_modulation.1.weight")),
                 orc.w("final_layer.adaLN_modulation.2.weight")) + lora[:, : 2 * D]
    shift, scale = m.chunk(2, dim=1)
    xm = F.layer_norm(xs, (D,), eps=1e-6) * (1 + scale.unsqueeze(0)) + shift.unsqueeze(0)
    y_loc = F.linear(xm, orc.w("final_layer.linear.weight")).squeeze(1)
    y = par.allgather_rows(y_loc.contiguous(), plan, group)
    return O.unpatchify(y.reshape(Tp, Hp * Wp, -1), 1, Tp, Hp, Wp, 1, 2, 16)


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        from oracle import dit_oracle as O
        pkg = load_package()
        par = pkg.parallel
        net = tiny_net(pkg, 256, 2, 2)
        sw = pkg.synthetic_weights
        sd = sw.synth_state_dict(net, torch.bfloat16)
        x = sw.synth_tensor("par.x", (1, 16, 2, 8, 8), torch.float32, scale=2.0)
        cond = sw.synth_tensor("par.c", (1, 16, 2, 8, 8), torch.float32, scale=1.0)
        t, ci = torch.tensor(1.3), torch.full((1, 1), 2, dtype=torch.long)
        orc = O.DitOracle(sd, net, dtype=torch.float32)
        with torch.no_grad():
            full = orc.forward(x, t, cond, ci)
            shard = _sharded_forward(orc, O, par, x, t, cond, ci, dist.group.WORLD)
            shard2 = _sharded_forward(orc, O, par, x, t, cond, ci, dist.group.WORLD, exchange="a2a")
        err = max(((shard - full).norm() / full.norm()).item(), ((shard2 - full).norm() / full.norm()).item())
        # helpers on their own
        plan = par.ShardPlan(12, rank, world)
        buf = torch.zeros(12, 3)
        plan.band(buf)[:] = rank + 1
        par.allgather_rows_(buf, plan, dist.group.WORLD)
        ok_gather = bool((buf[:6] == 1).all() and (buf[6:] == 2).all())
        snd = torch.full((world, 3, 2), float(rank)) + torch.arange(world).view(world, 1, 1) * 10       # slab r -> rank r
        rcv = torch.empty_like(snd)
        par.alltoall_rows_(snd, rcv, dist.group.WORLD)
        ok_gather = ok_gather and all(bool((rcv[r] == r + rank * 10).all()) for r in range(world))
        # the exchange in two parts by destination rank (the return all-to-all sent under the attention tail): parts add up to the whole
        for cut in (0, 1, world):
            rcv2 = torch.full_like(snd, -1.0)
            w0 = par.alltoall_bands_(snd, rcv2, 0, cut, dist.group.WORLD, async_op=True)
            w1 = par.alltoall_bands_(snd, rcv2, cut, world, dist.group.WORLD, async_op=True)
            for w in (w0, w1):
                if w is not None:
                    w.wait()
            ok_gather = ok_gather and torch.equal(rcv2, rcv)
        q.put((rank, err, ok_gather))
    finally:
        dist.destroy_process_group()


def test_token_band_sharding_equals_unsharded_oracle():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    for rank, err, ok in res:
        assert ok, f"rank {rank}: all-gather layout wrong"
        assert err < 1e-5, f"rank {rank}: sharded vs unsharded rel-L2 {err}"


def test_shard_plan_rejects_uneven_split(pkg):
    with pytest.raises(ValueError):
        pkg.parallel.ShardPlan(10, 0, 3)
    p = pkg.parallel.ShardPlan(18432, 5, 8)
    assert (p.rows, p.start, p.stop) == (2304, 11520, 13824)


def _bands_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        par = load_package().parallel
        n, C = 5, 3
        snd = (torch.arange(world * n * C, dtype=torch.float32).view(world, n, C) + 1000.0 * rank).to(torch.bfloat16)
        whole = torch.empty_like(snd)
        par.alltoall_rows_(snd, whole, dist.group.WORLD)
        ok = all(bool((whole[r] == (torch.arange(n * C, dtype=torch.float32).view(n, C) + rank * n * C + 1000.0 * r).to(torch.bfloat16)).all())
                 for r in range(world))
        for cut in range(world + 1):                     # every way of sending the first `cut` destination bands early
            got = torch.full_like(snd, -7.0)
            works = [par.alltoall_bands_(snd, got, 0, cut, dist.group.WORLD, async_op=True),
                     par.alltoall_bands_(snd, got, cut, world, dist.group.WORLD, async_op=True)]
            for w in works:
                if w is not None:
                    w.wait()
            ok = ok and torch.equal(got, whole)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_return_exchange_in_two_parts_equals_one_all_to_all(world):
    """parallel.alltoall_bands_ at the world sizes the driver runs (4, 8): for every cut, the bands [0, cut) and [cut, world) sent as
    two uneven all-to-alls deliver exactly what the one equal-split all-to-all delivers."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_bands_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    for rank, ok in sorted(q.get(timeout=10) for _ in range(world)):
        assert ok, f"rank {rank}"
