#!/usr/bin/env python3
"""Per-layer timing of the tokenizer's implicit-GEMM convolutions at the headline clip (57 f x 576 x 1024): which shapes
carry the time, and at what MFMA / HBM rate each runs.   python tools/convbench.py [--frames 57 --height 576 --width 1024]"""
import argparse
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=57)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    args = ap.parse_args()
    pkg = load_package()
    N, NV, sw = pkg.native, pkg.native_vae, pkg.synthetic_weights
    dev = torch.device("cuda")
    shapes = []
    real = NV.conv3d

    def traced(x, w, bias, N_out, k, *a, **kw):
        shapes.append((x.T, x.H, x.W, x.C, N_out, tuple(k), kw.get("stride", a[0] if a else (1, 1, 1))))
        return real(x, w, bias, N_out, k, *a, **kw)

    vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=dev), device=dev)
    clip = sw.synth_tensor("bench.rgb", (1, 3, args.frames, args.height, args.width), torch.float32, device=dev).to(torch.bfloat16)
    vae.decode(vae.encode(clip))
    torch.cuda.synchronize()
    # patch every module-level reference to conv3d (the tokenizer imports the module, not the function)
    NV.conv3d = traced
    vt = N.KernelTimer(names=("conv",))
    N.set_timer(vt)
    z = vae.encode(clip)
    n_enc = len(vt.records)
    vae.decode(z)
    torch.cuda.synchronize()
    N.set_timer(None)
    NV.conv3d = real
    rows = OrderedDict()
    conv_recs = [r for r in vt.records if r[0] == "conv"]
    for i, (name, s, e, fl, by) in enumerate(conv_recs):
        shp = shapes[i] if len(shapes) == len(conv_recs) else None
        key = (("enc" if i < n_enc else "dec"), shp, fl, by)
        d = rows.setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += s.elapsed_time(e)
    tot = sum(d[1] for d in rows.values())
    print(f"{len(conv_recs)} conv launches, {tot:.2f} ms")
    for (leg, shp, fl, by), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{leg} {str(shp):58s} x{n:3d} {ms:7.3f} ms ({ms / tot * 100:4.1f} %)  {ms / n * 1e3:8.1f} us each  "
              f"{fl / (ms / n * 1e-3) / 1e12:7.1f} TF/s  {by / (ms / n * 1e-3) / 1e9:7.1f} GB/s")


if __name__ == "__main__":
    main()
