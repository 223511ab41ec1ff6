#!/usr/bin/env python3
"""Tokenizer leg alone at the headline clip: encode / decode ms (HIP events) over a few rounds."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
dev = torch.device("cuda")
sw = pkg.synthetic_weights
T, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (57, 576, 1024))]
vae = pkg.CleanVAE.CleanVAE(state_dict=sw.synth_vae_state_dict(device=dev), device=dev)
clip = sw.synth_tensor("bench.rgb", (1, 3, T, H, W), torch.float32, device=dev).to(torch.bfloat16)
vae.decode(vae.encode(clip))
torch.cuda.synchronize()
for r in range(4):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    z = vae.encode(clip)
    e[1].record()
    y = vae.decode(z)
    e[2].record()
    torch.cuda.synchronize()
    print(f"round {r}: encode {e[0].elapsed_time(e[1]):.2f} ms  decode {e[1].elapsed_time(e[2]):.2f} ms", flush=True)
