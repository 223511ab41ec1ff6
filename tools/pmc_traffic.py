"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, MI355X_MICROARCH.md 'rocprofv3 PMC slots') of
the bench command into per-kernel L2-miss traffic per launch -> profiles/rNN_pmc_traffic.json (read by bench.py's roofline).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cfg
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cfg
    python tools/pmc_traffic.py gpurun_out/pmc_f/.../f_counter_collection.csv gpurun_out/pmc_w/.../w_counter_collection.csv [out.json]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, hence corrected bytes =
2 * FETCH + WRITE (the guide's correction)."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = {"gemm": ("gemm256s_kernel", "gemm256w_kernel", "gemm_tall_kernel", "gemm144_kernel", "gemm_bf16_kernel"), "attention": ("attention16_fwd_kernel", "attention_fwd_kernel", "attention_combine_kernel"),
            "conv": ("conv_igemm_kernel", "conv256s_kernel")}


def fold(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return tot, cnt


def main():
    fpath, wpath = sys.argv[1], sys.argv[2]
    ft, fc = fold(fpath, "FETCH_SIZE")
    wt, wc = fold(wpath, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(ft) | set(wt), key=lambda n: -(2 * ft.get(n, 0) + wt.get(n, 0))):
        n = max(fc.get(name, 0), wc.get(name, 0), 1)
        fk, wk = ft.get(name, 0.0) / max(fc.get(name, 1), 1), wt.get(name, 0.0) / max(wc.get(name, 1), 1)
        kernels[name] = {"launches": n, "fetch_KiB_per_launch_raw": round(fk, 1), "write_KiB_per_launch": round(wk, 1),
                         "hbm_bytes_per_launch_corrected": int((2 * fk + wk) * 1024)}
    families = {}
    for fam, names in FAMILIES.items():
        tb, tl = 0.0, 0
        for k, v in kernels.items():
            if any(s in k for s in names) and "combine" not in k:
                tb += v["hbm_bytes_per_launch_corrected"] * v["launches"]
                tl += v["launches"]
        if tl:
            # per wrapper call (what bench.py's roofline calls a launch): a forward makes 114 GEMM / 28 attention calls and the
            # profiled command runs 2 forwards; a GEMM call may be two kernels (whole-round + tail)
            # per wrapper call: a GEMM call may be two kernels (whole-round + tail) -> 4 calls per block (+ embed / final), one
            # attention call per block; PMC_FORWARDS forwards were profiled over PMC_BLOCKS blocks each; convs per kernel launch
            fw, blocks = int(os.environ.get("PMC_FORWARDS", "2")), int(os.environ.get("PMC_BLOCKS", "28"))
            calls = {"gemm": (4 * blocks + 2) * fw, "attention": blocks * fw}.get(fam, tl)
            families[fam] = {"kernel_launches": tl, "calls": calls, "bytes_per_launch_corrected": int(tb / calls)}
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 1 "
                   "--no-cpu-baseline --no-cfg`; per-launch averages over all launches of the kernel; corrected = "
                   "2*FETCH_SIZE + WRITE_SIZE in bytes (gfx950 FETCH_SIZE counts 128-B requests as 64 B); a GEMM wrapper call "
                   "may be two kernel launches (whole-round + tail)",
           "kernels": {k: v for k, v in list(kernels.items())[:24]}, "families": families}
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst, families)


if __name__ == "__main__":
    main()
