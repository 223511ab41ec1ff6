#!/usr/bin/env python3
"""Library yardstick for the self-attention at the cfg-3 shape: torch.nn.functional.scaled_dot_product_attention on ROCm (whatever
backend torch picks: flash / efficient / math), same random data distribution as tools/kbench.py attn.  Not on the product path."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda")
H, S, d = 32, 18432, 128
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(1, H, S, d, generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
fl = 4.0 * H * S * S * d
for name, ctx in (("default dispatch", None),):
    try:
        for _ in range(2):
            o = F.scaled_dot_product_attention(q, k, v)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                o = F.scaled_dot_product_attention(q, k, v)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        ts.sort()
        print(f"library SDPA ({name}) [1,{H},{S},{d}] bf16: median {ts[2]:.3f} ms  {fl / ts[2] / 1e9:.1f} TF/s   min {ts[0]:.3f} ms")
    except Exception as e:                      # noqa: BLE001
        print(f"library SDPA ({name}) failed: {type(e).__name__}: {e}")
