"""Per-rank compute of the sequence-parallel DiT forward, measured on ONE GPU: the exchanges are replaced by no-ops, so the
kernels run on rank 0's shapes (token band of S/world rows, heads/world heads after the all-to-all); the peers' data is
faked by device copies of this rank's own slabs (same value distribution; the copies cost ~0.02 ms per exchange).
What it shows: the compute floor of `bench.py --gpus N` and which kernels lose efficiency at the per-rank shapes.

    python tools/rankbench.py --world 8 [--exchange a2a|gather] [--steps 4]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def tokenizer_band(pkg, args, dev):
    """One rank's row band of the tokenizer (rank 1: both neighbours exist); gathers are faked by device copies."""
    world = args.world
    sw = pkg.synthetic_weights
    pkg.parallel.allgather_stack = lambda local, group=None: local.unsqueeze(0).expand(world, *local.shape).contiguous()
    vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=dev), device=dev)
    clip = sw.synth_tensor("rb.rgb", (1, 3, args.frames, args.height, args.width), torch.float32, device=dev).to(torch.bfloat16)
    res = {}
    for w_ in (1, world):
        vae.model.rank, vae.model.world, vae.model.pg = (min(1, w_ - 1), w_, None)
        z = vae.encode(clip)
        vae.decode(z)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        z = vae.encode(clip)
        e[1].record()
        vae.decode(z)
        e[2].record()
        torch.cuda.synchronize()
        res[w_] = (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]))
    print(f"tokenizer {args.frames}f x {args.height} x {args.width}: one GPU encode {res[1][0]:.2f} ms decode {res[1][1]:.2f} ms; "
          f"one band of {world}: encode {res[world][0]:.2f} ms decode {res[world][1]:.2f} ms (exchanges faked by device copies)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--exchange", default="a2a")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--frames", type=int, default=57)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--blocks", type=int, default=28)
    ap.add_argument("--tokenizer", action="store_true", help="time this rank's band of the tokenizer instead of the DiT")
    ap.add_argument("--clips", type=int, default=1, help="clips stepped as one sharded batch (the node's 5 G-buffer passes)")
    args = ap.parse_args()
    os.environ["DRN_SP_EXCHANGE"] = args.exchange
    pkg = load_package()
    eng, N = pkg.dit_engine, pkg.native
    world = args.world
    eng.group_info = lambda pg=None: (0, world)

    def fake_alltoall(send, recv, pg=None, async_op=False):
        recv.copy_(send)

    def fake_allgather(full, plan, pg=None, async_op=False):
        if full.shape[0] == plan.S and plan.world > 1:
            full.view(plan.world, plan.rows, -1)[1:].copy_(plan.band(full).unsqueeze(0).expand(plan.world - 1, -1, -1))

    def fake_bands(send, recv, lo, hi, pg=None, async_op=False):
        if hi > lo:
            recv[lo:hi].copy_(send[lo:hi])

    eng.alltoall_rows_ = fake_alltoall
    eng.alltoall_bands_ = fake_bands
    eng.allgather_rows_ = fake_allgather
    dev = torch.device("cuda", 0)
    if args.tokenizer:
        return tokenizer_band(pkg, args, dev)
    cfg = pkg.diffusion_renderer_config.get_inverse_renderer_config(args.height, args.width, args.frames)
    net = dict(cfg["net"], num_blocks=args.blocks)
    sw = pkg.synthetic_weights
    dit = eng.HipDiT(net, sw.synth_state_dict(net, torch.bfloat16, device=dev), device=dev, process_group=object())
    F_, h, w = (args.frames - 1) // 8 + 1, args.height // 8, args.width // 8
    x = sw.synth_tensor("rb.x", (args.clips, 16, F_, h, w), torch.float32, device=dev, scale=2.0).to(torch.bfloat16)
    cond = sw.synth_tensor("rb.c", (args.clips, 16, F_, h, w), torch.float32, device=dev, scale=1.0).to(torch.bfloat16)
    sig = [80.0 * 0.8 ** i for i in range(args.steps + 1)]
    dit.prepare_timesteps(sig)
    dit(x, sig[0], cond, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in sig[1:]:
        dit(x, s_, cond, 3)
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host time to ENQUEUE a forward (must stay below the GPU time)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dit(x, sig[1], cond, 3)
    one_ms = (time.perf_counter() - t0) * 1e3                      # empty queue: pure host cost of one forward
    torch.cuda.synchronize()
    print(f"host cost of one forward on an empty queue: {one_ms:.2f} ms")
    timer = N.KernelTimer(sample_every=3)
    N.set_timer(timer)
    dit(x, sig[1], cond, 3)
    torch.cuda.synchronize()
    N.set_timer(None)
    print(f"world={world} exchange={dit.exchange} clips={args.clips}: {ms:.2f} ms per forward on this rank's shapes (no communication); "
          f"host enqueue {host_ms:.2f} ms per forward")
    for name, d in timer.summary().items():
        n = d["launches_seen"]
        print(f"  {name:10s} {n:4d} launches  avg {d['ms_avg']:.3f} ms  -> {d['ms_avg'] * n:7.2f} ms per forward, "
              f"{d['flops'] / d['ms_total'] / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
