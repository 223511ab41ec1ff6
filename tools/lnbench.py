import sys, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); N = pkg.native; lib = N.load_library()
dev = torch.device('cuda')
rows, D = 18432, 4096
x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
sh = torch.randn(1, D, device=dev).to(torch.bfloat16); sc = torch.randn(1, D, device=dev).to(torch.bfloat16); ad = torch.randn(1, D, device=dev).to(torch.bfloat16)
h = torch.empty_like(x)
for which in (0, 1, 0, 1):
    lib.drn_ln_force_kernel(which)
    for add in (None, ad):
        for _ in range(3): N.ln_modulate(x, sh, sc, out=h, add_vec=add)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): N.ln_modulate(x, sh, sc, out=h, add_vec=add)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nbytes = rows * D * 2 * (3 if add is not None else 2)
        print(f"kernel {'4 waves/row' if which else '1 wave/row '} add={add is not None}: {ms*1e3:7.1f} us  {nbytes/ms/1e6:7.1f} GB/s")
lib.drn_ln_force_kernel(-1)
