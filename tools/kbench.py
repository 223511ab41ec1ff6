#!/usr/bin/env python3
"""Micro-benchmarks of the two MFMA kernels at the cfg-3 shapes (S=18432, D=4096, 32 heads), random data.
    python tools/kbench.py [attn] [gemm] [--lib path/to/libdrn_variant.so ...]
Several --lib arguments are timed interleaved in ONE process (round-robin), as the CDNA guide's rule 24 asks."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["attn", "gemm"])
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--S", type=int, default=18432)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--splits", type=int, default=1, help="attention: split-KV chunks (drn_attention_splitkv_bf16)")
    ap.add_argument("--tiles", default="-1", help="gemm: comma list of forced tile kernels (-1 auto, 0 128^2, 1 256^2, 2 144x256)")
    ap.add_argument("--Sk", type=int, default=0, help="keys for attention (default: S); S is then the local query/token count")
    ap.add_argument("--cold", type=int, default=1, help="gemm: rotate over this many copies of the weights (and activations) so that "
                    "they come from HBM, not from the 256 MB Infinity Cache, as inside the model (4-6 copies)")
    ap.add_argument("--splitk", type=int, default=0, help="gemm: 1 = take the split-K path where drn_gemm_splitk_choice says so (few tokens)")
    ap.add_argument("--shapes", default="", help="attention: comma list of kernel bodies to time in one process (0 = 32x32x16, 1 = 16x16x32)")
    ap.add_argument("--diag", default="", help="attention: library built with -DATT_DIAG=1; prints the per-segment cycle shares")
    args = ap.parse_args()
    pkg = load_package()
    N = pkg.native
    libs = args.lib or [N.library_path()]
    handles = []
    for path in libs:
        lib = ctypes.CDLL(os.path.abspath(path))
        for name, at in N.SIGNATURES.items():
            if hasattr(lib, name):
                getattr(lib, name).argtypes = at
        handles.append(lib)
    dev = torch.device("cuda")
    S, D, H = args.S, 4096, 32
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)

    st = torch.cuda.current_stream().cuda_stream
    cases = []
    if "attn" in args.what:
        Sk = args.Sk or S
        qkv = rnd(max(S, Sk), 3 * D)
        o = torch.empty(S, D, dtype=torch.bfloat16, device=dev)
        fl = 4.0 * S * Sk * D

        ws = None
        if args.splits > 1:
            handles[0].drn_attention_splitkv_workspace_bytes.restype = ctypes.c_int64
            ws = torch.empty(handles[0].drn_attention_splitkv_workspace_bytes(1, H, S, args.splits), dtype=torch.uint8, device=dev)

        def run_attn(lib):
            if ws is not None:
                rc = lib.drn_attention_splitkv_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, o.data_ptr(),
                                                    1, H, S, Sk, 3 * D, 3 * D, 3 * D, D, 0, 0, 0, 0, 128 ** -0.5, args.splits,
                                                    ws.data_ptr(), st)
                assert rc == 0, rc
                return
            rc = lib.drn_attention_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, o.data_ptr(), 1, H, S, Sk,
                                        3 * D, 3 * D, 3 * D, D, 0, 0, 0, 0, 128 ** -0.5, st)
            assert rc == 0, rc
        if args.shapes:
            for sh in [int(v) for v in args.shapes.split(",")]:
                def run_shape(lib, sh=sh):
                    lib.drn_attention_force_shape16(sh)
                    run_attn(lib)
                    lib.drn_attention_force_shape16(-1)
                cases.append(("attention Sq=%d Sk=%d body %s" % (S, Sk, ("32x32x16", "16x16x32")[sh]), run_shape, fl))
        else:
            cases.append(("attention Sq=%d Sk=%d" % (S, Sk), run_attn, fl))
    ws_box = [None]
    if "gemm" in args.what:
        a = rnd(S, D)
        for (Nn, K, epi, nm) in [(3 * D, D, 0, "qkv"), (D, D, 2, "out+gate"), (4 * D, D, 1, "mlp1+gelu"), (D, 4 * D, 2, "mlp2+gate")]:
            A = a if K == D else rnd(S, K, scale=0.3)
            Wt = rnd(Nn, K, scale=K ** -0.5)
            As = [A] + [A.clone() for _ in range(args.cold - 1)]
            Ws = [Wt] + [Wt.clone() for _ in range(args.cold - 1)]
            C = torch.empty(S, Nn, dtype=torch.bfloat16, device=dev)
            R = rnd(S, Nn) if epi == 2 else None
            gate = rnd(1, Nn) if epi == 2 else None
            turn = [0]

            tiles = [int(t) for t in args.tiles.split(",")]
            for tile in tiles:
              def run_gemm(lib, As=As, Ws=Ws, C=C, R=R, gate=gate, Nn=Nn, K=K, epi=epi, tile=tile, turn=turn):
                turn[0] = (turn[0] + 1) % len(As)
                A, Wt = As[turn[0]], Ws[turn[0]]
                lib.drn_gemm_force_tile(tile)
                splits = lib.drn_gemm_splitk_choice(S, Nn, K) if args.splitk else 1
                if splits > 1:                                  # few tokens: the path native.gemm takes (slices + epilogue kernel)
                    lib.drn_gemm_splitk_workspace_bytes.restype = ctypes.c_int64
                    nb = lib.drn_gemm_splitk_workspace_bytes(S, Nn, splits)
                    if ws_box[0] is None or ws_box[0].numel() < nb:
                        ws_box[0] = torch.empty(nb, dtype=torch.uint8, device=dev)
                    rc = lib.drn_gemm_bf16_splitk(A.data_ptr(), Wt.data_ptr(), C.data_ptr(), S, Nn, K, K, K, Nn, epi,
                                                  gate.data_ptr() if gate is not None else None,
                                                  R.data_ptr() if R is not None else None, Nn, S, splits, ws_box[0].data_ptr(), st)
                else:
                    rc = lib.drn_gemm_bf16(A.data_ptr(), Wt.data_ptr(), C.data_ptr(), S, Nn, K, K, K, Nn, epi,
                                           gate.data_ptr() if gate is not None else None,
                                           R.data_ptr() if R is not None else None, Nn, S, st)
                lib.drn_gemm_force_tile(-1)
                assert rc == 0, rc
              cases.append((f"gemm {nm} [{S}x{K}]x[{Nn}x{K}] tile {tile}", run_gemm, 2.0 * S * Nn * K))
    if args.diag and "attn" in args.what:
        lib = ctypes.CDLL(os.path.abspath(args.diag))
        for name, at in N.SIGNATURES.items():
            if hasattr(lib, name):
                getattr(lib, name).argtypes = at
        Sk = args.Sk or S
        nwg = ((S + 255) // 256) * H
        dbg = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
        for _ in range(40):                        # ~0.2 s of back-to-back launches: the clock settles under load
            rc = lib.drn_attention_splitkv_bf16(qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, o.data_ptr(), 1, H, S, Sk,
                                                3 * D, 3 * D, 3 * D, D, 0, 0, 0, 0, 128 ** -0.5, 1, dbg.data_ptr(), st)
            assert rc == 0, rc
        torch.cuda.synchronize()
        d = dbg.view(nwg, 8, 8).double().cpu()
        names = ["M work", "DMA wait", "barrier after M", "S work", "barrier after S"]
        for g, gname in ((slice(0, 4), "G0 (waves 0-3)"), (slice(4, 8), "G1 (waves 4-7)")):
            x = d[:, g, :].reshape(-1, 8)
            tiles = x[:, 7].median().item()
            tot = x[:, 5].median().item()
            print(f"{gname}: loop {tot:.0f} cycles = {tot / tiles:.0f} per tile over {tiles:.0f} tiles; in-kernel clock "
                  f"{(x[:, 5] / x[:, 6]).median().item() * 100:.0f} MHz")
            for i, nm in enumerate(names):
                print(f"    {nm:18s} {x[:, i].median().item() / tiles:8.0f} cycles per tile  ({x[:, i].median().item() / tot * 100:5.1f} %)")
    for name, fn, fl in cases:
        times = [[] for _ in handles]
        ref_out = None
        for i, lib in enumerate(handles):
            if name.startswith("attention"):
                o.zero_()
            fn(lib)
            if name.startswith("attention") and len(handles) > 1:          # variants keep the summation order: same bits expected
                torch.cuda.synchronize()
                if ref_out is None:
                    ref_out = o.clone()
                else:
                    print(f"    {os.path.basename(libs[i])}: identical to {os.path.basename(libs[0])}: {torch.equal(o, ref_out)}  "
                          f"max|diff| {(o.float() - ref_out.float()).abs().max().item():.3e}", flush=True)
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for i, lib in enumerate(handles):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn(lib)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / 3)
        for i, path in enumerate(libs):
            t = sorted(times[i])
            med, mn = t[len(t) // 2], t[0]
            print(f"{name:46s} {os.path.basename(path):28s} median {med:8.3f} ms  {fl/med/1e9:8.1f} TF/s   min {mn:8.3f} ms {fl/mn/1e9:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
