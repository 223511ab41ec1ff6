"""G-buffer passes stepped one by one vs as one batch, full-width 28-block DiT, synthetic weights.

    python tools/passbench.py --frames 1 --height 256 --width 256          # BASELINE cfg 1
    python tools/passbench.py --frames 9 --height 512 --width 512          # cfg 2

Prints steps/s of one denoising step for P = 1 and P = 5 passes at guidance 0 and the CFG pair (2 clips) at guidance > 0.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=28)
    args = ap.parse_args()
    pkg = load_package()
    dev = torch.device("cuda", 0)
    cfg = pkg.diffusion_renderer_config.get_inverse_renderer_config(args.height, args.width, args.frames)
    net = dict(cfg["net"], num_blocks=args.blocks)
    sw = pkg.synthetic_weights
    dit = pkg.dit_engine.HipDiT(net, sw.synth_state_dict(net, torch.bfloat16, device=dev), device=dev)
    F_, h, w = (args.frames - 1) // 8 + 1, args.height // 8, args.width // 8
    S = F_ * (h // 2) * (w // 2)
    sigmas = [80.0 * 0.8 ** i for i in range(args.steps + 2)]
    dit.prepare_timesteps(sigmas)

    def run(B, cis):
        x = sw.synth_tensor("pb.x", (B, 16, F_, h, w), torch.float32, device=dev, scale=2.0).to(torch.bfloat16)
        cond = sw.synth_tensor("pb.c", (1, 16, F_, h, w), torch.float32, device=dev, scale=1.0).to(torch.bfloat16)
        for s_ in sigmas[:2]:
            dit(x, s_, cond, cis)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in sigmas[2:]:
            dit(x, s_, cond, cis)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    t1 = run(1, [3])
    t2 = run(2, [3, 0])
    t5 = run(5, [0, 1, 2, 3, 4])
    print(f"S={S} tokens per clip, {args.blocks} blocks")
    print(f"  1 clip : {t1:8.2f} ms per forward")
    print(f"  2 clips: {t2:8.2f} ms per batched forward = {t2 / 2:7.2f} ms per clip  (CFG pair: x{2 * t1 / t2:.2f} vs two forwards)")
    print(f"  5 clips: {t5:8.2f} ms per batched forward = {t5 / 5:7.2f} ms per clip  (5 passes: x{5 * t1 / t5:.2f} vs five forwards)")


if __name__ == "__main__":
    main()
