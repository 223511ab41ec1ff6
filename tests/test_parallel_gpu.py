"""The HIP engine's token-band path on real hardware: two ranks (both on the one visible GPU, gloo for the exchange -
RCCL refuses two ranks on one device) run HipDiT sharded and must reproduce the single-rank HipDiT output.  This
exercises both exchanges with the real kernels - the head <-> token all-to-all (rank-major fused projection, regroup kernel,
norm + RoPE after the exchange) and the K|V all-gather (split Q / K|V projections into the gather buffer, RoPE position
offset) - and the gathered final projection; the RCCL transport itself is exercised by bench.py --gpus N."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, tiny_net

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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


def _worker(rank, world, port, q, exchange, wide=False):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["DRN_SP_EXCHANGE"] = exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        pkg = load_package()
        dev = torch.device("cuda:0")
        net = tiny_net(pkg, 1024, 1, 8) if wide else tiny_net(pkg, 256, 2, 2)
        lat = (2, 64, 64) if wide else (2, 16, 16)
        if wide == "clip":
            # the headline clip's token count: 4 heads x 72 q-blocks = 288 workgroups per rank -> the attention runs as a whole
            # round + a split-KV tail, and the return all-to-all goes in two parts (band 0 under the tail)
            lat = (8, 72, 128)
        sw = pkg.synthetic_weights
        sd = sw.synth_state_dict(net, torch.bfloat16, device=dev)
        x = sw.synth_tensor("pg.x", (1, 16) + lat, torch.float32, scale=2.0).to(torch.bfloat16).to(dev)
        cond = sw.synth_tensor("pg.c", (1, 16) + lat, torch.float32, scale=1.0).to(torch.bfloat16).to(dev)
        if wide:
            # heads/world * 128 = 512 columns per rank: the projections write / read the rank-major slabs through the
            # blocked-layout GEMM (forced onto the 256^2 tile: the cost model keeps shapes this small on the 128^2 kernel)
            pkg.native.load_library().drn_gemm_force_tile(1)
        single = pkg.dit_engine.HipDiT(net, sd, device=dev)
        sharded = pkg.dit_engine.HipDiT(net, sd, device=dev, process_group=dist.group.WORLD)
        assert sharded.exchange == exchange
        if wide == "clip":
            S = lat[0] * lat[1] * lat[2] // 4
            ap = pkg.native.attention_plan(1, net["num_heads"] // world, S, S)
            assert len(ap) == 2 and ap[0][1] // (S // world) >= 1 and sharded._split_return
        y1 = single(x, torch.tensor(1.7), cond, 2)
        y2 = sharded(x, torch.tensor(1.7), cond, 2)
        torch.cuda.synchronize()
        if wide == "clip":
            # one rank covers its 8 heads x 72 q-blocks with a different split-KV tail (64 workgroups in 4 key chunks) than a rank
            # of the sharded run (32 workgroups in 8): the two agree to rounding, not bit for bit.  What must hold bit for bit
            # is the exchange: the return all-to-all in two parts against the same engine sending it as one collective
            whole = pkg.dit_engine.HipDiT(net, sd, device=dev, process_group=dist.group.WORLD)
            whole._split_return = False
            y3 = whole(x, torch.tensor(1.7), cond, 2)
            torch.cuda.synchronize()
            rel = float((y1.float() - y2.float()).norm() / y1.float().norm())
            assert rel < 2e-3, rel
            y1 = y3
        same = bool(torch.equal(y1, y2))
        if wide != "clip":
            # two clips stepped as ONE sharded batch (token-local ops batched over the clips, the exchanges clip by clip) against
            # the same two clips run one after the other: bit for bit (SURVEY.md 8f N1 under sequence parallelism)
            xb = torch.cat([x, sw.synth_tensor("pg.x2", (1, 16) + lat, torch.float32, scale=2.0).to(torch.bfloat16).to(dev)], 0)
            cb = torch.cat([cond, sw.synth_tensor("pg.c2", (1, 16) + lat, torch.float32, scale=1.0).to(torch.bfloat16).to(dev)], 0)
            yb = sharded(xb, torch.tensor(1.7), cb, [2, 4])
            y_one = torch.cat([sharded(xb[i:i + 1], torch.tensor(1.7), cb[i:i + 1], ci) for i, ci in enumerate([2, 4])], 0)
            torch.cuda.synchronize()
            same = same and bool(torch.equal(yb, y_one)) and bool(torch.equal(yb[:1], y2)) and not bool(torch.equal(yb[:1], yb[1:]))
        q.put((rank, same, float((y1.float() - y2.float()).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,wide,world", [("a2a", False, 2), ("gather", False, 2), ("a2a", True, 2), ("a2a", "clip", 2),
                                                 ("a2a", True, 4), ("gather", False, 4)])
def test_sharded_hipdit_equals_single_rank(gpu, exchange, wide, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange, wide)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank, same, maxdiff in sorted(q.get(timeout=10) for _ in range(world)):
        # token-local kernels are row-independent and attention sums keys in the same tile order -> identical bits
        # (DESIGN.md section 5 claims bit-identity: nothing weaker is accepted)
        assert same, f"rank {rank}: sharded != single (max |diff| {maxdiff})"
        print(f"rank {rank}: identical={same} max|diff|={maxdiff}")


def _rccl_worker(port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=port, rank=0, world_size=1, device_id=dev)
    try:
        from __graft_entry__ import load_package
        pkg = load_package()
        pkg.parallel.SINGLE_RANK_COLLECTIVES = True
        net = tiny_net(pkg, 256, 2, 2)
        sw = pkg.synthetic_weights
        sd = sw.synth_state_dict(net, torch.bfloat16, device=dev)
        x = sw.synth_tensor("pg.x", (1, 16, 2, 16, 16), torch.float32, scale=2.0).to(torch.bfloat16).to(dev)
        cond = sw.synth_tensor("pg.c", (1, 16, 2, 16, 16), torch.float32, scale=1.0).to(torch.bfloat16).to(dev)
        y1 = pkg.dit_engine.HipDiT(net, sd, device=dev)(x, torch.tensor(1.7), cond, 2)
        res = {}
        for mode in ("a2a", "gather"):
            os.environ["DRN_SP_EXCHANGE"] = mode
            eng = pkg.dit_engine.HipDiT(net, sd, device=dev, process_group=dist.group.WORLD)
            assert eng.exchange == mode
            for _ in range(3):                                   # repeated: buffer reuse across async exchanges
                y2 = eng(x, torch.tensor(1.7), cond, 2)
            torch.cuda.synchronize()
            res[mode] = bool(torch.equal(y1, y2))
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_exchange_paths_over_rccl_single_rank(gpu):
    """The real transport: a 1-rank RCCL group drives both exchange paths (async all_to_all_single / all_gather_into_tensor on
    device buffers, work.wait() stream ordering, workspace reuse over repeated forwards) and must not change a bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_rdzv(), q))
    p.start()
    p.join(timeout=300)
    assert p.exitcode == 0
    res = q.get(timeout=10)
    assert res == {"a2a": True, "gather": True}, res


def _cond_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        pkg = load_package()
        dev = torch.device("cuda:0")
        sw, cfgm = pkg.synthetic_weights, pkg.diffusion_renderer_config
        net = tiny_net(pkg, 256, 1, 2, forward=True)
        cfg = dict(cfgm.get_forward_renderer_config(), net=net, model_type="forward")
        sd = sw.synth_state_dict(net, torch.bfloat16, device=dev)
        vsd = sw.synth_vae_state_dict(device=dev)
        outs = []
        for pg in (None, dist.group.WORLD):
            model = pkg.model_diffusion_renderer.CleanDiffusionRendererModel(dict(cfg), device=dev, process_group=pg)
            model.load_state_dict(sd, strict=True)
            model.vae = pkg.CleanVAE.CleanVAE(state_dict=vsd, device=dev, process_group=pg)   # row-band tokenizer when sharded
            batch = {k: sw.synth_tensor("cw." + k, (1, 3, 9, 32, 32), torch.float32).to(torch.bfloat16).to(dev)
                     for k in cfgm.FORWARD_CONDITION_KEYS if k != "metallic"}          # one key missing -> zero latent + zero mask
            batch["video"] = batch["depth"]
            lat = model.prepare_diffusion_renderer_latent_conditions(batch)
            x0 = model.generate_samples_from_batch(dict(batch), guidance=0.0, seed=7, state_shape=[16, 2, 4, 4], num_steps=2,
                                                   init_noise=sw.synth_tensor("cw.n", (1, 16, 2, 4, 4), torch.float32, scale=80.0))
            outs.append((lat, x0, model.decode(x0)))
        torch.cuda.synchronize()
        q.put((rank, bool(torch.equal(outs[0][0], outs[1][0])),
               bool(torch.equal(outs[0][1], outs[1][1])) and bool(torch.equal(outs[0][2], outs[1][2])), tuple(outs[0][0].shape)))
    finally:
        dist.destroy_process_group()


def test_condition_encodes_spread_over_ranks(gpu):
    """Forward renderer on 2 ranks: each rank encodes every second condition map (alone, although its tokenizer could cut a map
    into row bands) and the latents are all-gathered; the decode runs band-sharded.  Condition latent, sampled latent and
    decoded video must equal the single-rank ones bit for bit."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_cond_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank, same_lat, same_x0, shape in sorted(q.get(timeout=10) for _ in range(world)):
        assert shape == (1, 8 * 17, 2, 4, 4)
        assert same_lat and same_x0, f"rank {rank}: {same_lat} {same_x0}"


def _tok_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=port, rank=rank, world_size=world)
    try:
        from __graft_entry__ import load_package
        pkg = load_package()
        dev = torch.device("cuda:0")
        sw = pkg.synthetic_weights
        sd = sw.synth_vae_state_dict(device=dev)
        one = pkg.CleanVAE.CleanVAE(state_dict=sd, device=dev)
        banded = pkg.CleanVAE.CleanVAE(state_dict=sd, device=dev, process_group=dist.group.WORLD)
        T, H, W = 9, 64, 96
        clip = sw.synth_tensor("tok.clip", (1, 3, T, H, W), torch.float32).to(torch.bfloat16).to(dev)
        z1, z2 = one.encode(clip), banded.encode(clip)
        v1, v2 = one.decode(z1), banded.decode(z1)
        z3 = banded.encode(clip, bands=1)                       # explicit single-rank call on a banded tokenizer
        torch.cuda.synchronize()

        def cmp(a, b):
            a, b = a.float(), b.float()
            return (float((a - b).norm() / b.norm()), float((a == b).float().mean()), tuple(a.shape))
        q.put((rank, cmp(z2, z1), cmp(v2, v1), bool(torch.equal(z3, z1))))
    finally:
        dist.destroy_process_group()


def test_tokenizer_row_bands_equal_single_rank(gpu):
    """every frame cut into 2 bands of image rows (halo-row exchange before the spatial convs, GroupNorm
    statistics summed over the ranks, K/V gathered for the mid-block attention).  Same arithmetic per element; the GroupNorm
    sums are accumulated in fp64, which makes them independent of the split - so the result is the single-rank one, bit for bit."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_tok_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank, (ez, same_z, shz), (ev, same_v, shv), alone_ok in sorted(q.get(timeout=10) for _ in range(world)):
        print(f"rank {rank}: latent rel-L2 {ez:.2e} ({same_z:.3f} identical), video rel-L2 {ev:.2e} ({same_v:.3f} identical)")
        assert shz == (1, 16, 2, 8, 12) and shv == (1, 3, 9, 64, 96)
        assert alone_ok
        assert ez == 0.0 and ev == 0.0 and same_z == 1.0 and same_v == 1.0


@pytest.mark.parametrize("exchange", ["a2a", "gather"])
def test_bench_under_torchrun_one_rank_rccl(gpu, exchange):
    """bench.py as a rank of torch.distributed.run (1 rank on this box's one GPU, RCCL backend): process-group start-up, the
    exchange probe, both sequence-parallel exchanges through real RCCL calls, the exchange-exposure timers and the N > 1 fields
    of the JSON line.  2 of the 28 blocks; the transport over xGMI itself needs the driver's 8-GPU node."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, DRN_SP_EXCHANGE=exchange, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--blocks", "2", "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline", "--no-tokenizer", "--no-cfg"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["rccl_ranks"] == 1 and rec["value"] > 0 and rec["host_enqueue_ms_empty_queue"] > 0
    ex = rec["exchange"]
    want = {"a2a": {"a2a k|v", "a2a q", "a2a o (return)"}, "gather": {"gather k|v"}}[exchange]
    assert ex["kind"] == exchange and set(ex["exposed_ms_per_layer"]) == want, ex
    print(exchange, ex)


@pytest.mark.parametrize("exchange", ["a2a", "gather"])
def test_bench_two_ranks_rehearsal_over_gloo(gpu, exchange):
    """`python bench.py --gpus 2` exactly as the driver calls it (self-launching: child torch.distributed.run, two ranks), with
    DRN_BENCH_BACKEND=gloo so that both ranks can share this box's one GPU: the whole N > 1 path of bench.py runs - token-band
    sharded engine at the headline clip's shapes (2 of the 28 blocks), the return all-to-all in two parts, barrier + max-over-ranks
    timing, rank 0's single JSON line - only the transport is host-staged gloo instead of RCCL over xGMI."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, DRN_SP_EXCHANGE=exchange, DRN_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--blocks", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-tokenizer", "--no-cfg"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["steps"] == 2 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert "gloo" in rec["transport"] and "exchange_fallback" not in rec
    assert rec["config"]["tokens"] == 18432 and ("all-to-all" if exchange == "a2a" else "all-gather") in rec["config"]["parallelism"]
    print(exchange, rec["ms_per_step"], rec["config"]["parallelism"])
