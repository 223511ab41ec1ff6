import sys, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); V = pkg.native_vae
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
q = torch.randn(9216, 512, generator=g).to(torch.bfloat16).to(dev); k = torch.randn(9216, 512, generator=g).to(torch.bfloat16).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('implicit-GEMM 128^2 fp32 scores: %.1f us' % t(lambda: V.dense_gemm(q, k, out_f32=True, alpha=0.044)))
print('256^2 tile fp32 scores:          %.1f us' % t(lambda: V.scores_f32(q, k)))
s = V.scores_f32(q, k)
print('softmax scaled: %.1f us' % t(lambda: V.softmax_rows(s, 9216, 9216, 0.044)))
