import sys, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); N = pkg.native
dev = torch.device('cuda')
S, D = 18432, 4096
g = torch.Generator().manual_seed(0)
a = [torch.randn(S, D, generator=g).to(torch.bfloat16).to(dev) for _ in range(3)]
w = [(torch.randn(4 * D, D, generator=g) / 64).to(torch.bfloat16).to(dev) for _ in range(3)]
out = torch.empty(S, 4 * D, dtype=torch.bfloat16, device=dev)
for epi, name in ((N.EPI_NONE, "plain"), (N.EPI_GELU, "gelu"), (N.EPI_NONE, "plain"), (N.EPI_GELU, "gelu")):
    for i in range(3): N.gemm(a[i], w[i], out=out, epilogue=epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(12): N.gemm(a[r % 3], w[r % 3], out=out, epilogue=epi)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 12
    print(f"mlp-up {name}: {ms:.3f} ms  {2*S*4*D*D/ms/1e9:.1f} TF/s")
