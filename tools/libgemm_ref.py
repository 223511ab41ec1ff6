"""Library yardstick for DESIGN.md: torch.nn.functional.linear (hipBLASLt / rocBLAS, plain GEMM without epilogue) on the DiT's
linear shapes at cfg 3, full clip and 8-way token band.  Not used by the product path.

    python tools/libgemm_ref.py
"""
import torch
dev = torch.device("cuda")
S, D = 18432, 4096
for (M, N, K) in [(S, 3*D, D), (S, D, D), (S, 4*D, D), (S, D, 4*D), (2304, 4*D, D), (2304, D, 4*D)]:
    a = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    for _ in range(3):
        torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            torch.nn.functional.linear(a, w)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    t = sorted(ts)[len(ts) // 2]
    print(f"library GEMM (torch.nn.functional.linear -> hipBLASLt/rocBLAS) [{M}x{K}]x[{N}x{K}]: {t:.3f} ms  {2.0*M*N*K/t/1e9:.1f} TF/s", flush=True)
