"""Fold a rocprofv3 counter pass of the bench command into MFMA utilisation per kernel -> profiles/rNN_pmc_mfma.json.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --kernel-trace \
        --output-format csv -d gpurun_out/prof/m -o m -- python3 bench.py --blocks 4 --steps 1 --warmup 1 ...
    python tools/pmc_mfma.py <m_counter_collection.csv> [<m_kernel_trace.csv>] out.json

What the counters are (rocprofiler-sdk counter_defs.yaml, gfx950): SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipe of a SIMD is
busy, summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE = cycles the graphics engine is busy, reported as the SUM over the 8 XCDs
(MI355X_MICROARCH.md, DVFS give-back); SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = bf16 MFMA FLOPs executed.
    MfmaUtil   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)           (the SDK's own derived metric, same formula)
    flops      = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 per launch  (compare with the algorithmic 2MNK / 4 S^2 d H)
MfmaUtil is the share of CYCLES the matrix pipes were busy at whatever clock the chip held; the roofline fraction bench.py reports
divides by the 2.4 GHz peak, so roofline_frac ~= MfmaUtil x (held clock / 2.4 GHz)."""
import csv
import json
import sys
from collections import defaultdict

SIMDS, XCDS = 1024, 8


def main():
    cpath = sys.argv[1]
    tpath = sys.argv[2] if len(sys.argv) > 3 else None
    dst = sys.argv[-1]
    per = defaultdict(lambda: defaultdict(float))       # kernel -> counter -> sum over dispatches
    cnt = defaultdict(lambda: defaultdict(int))
    with open(cpath, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            per[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[name][row["Counter_Name"]] += 1
    dur = defaultdict(float)
    ndur = defaultdict(int)
    if tpath:
        with open(tpath, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0]
                dur[name] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                ndur[name] += 1
    out = {}
    for name, c in per.items():
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        n = max(cnt[name].values())
        rec = {"launches": n, "mfma_busy_cycles_per_launch": round(busy / n), "gui_active_cycles_per_launch_per_xcd": round(gui / n / XCDS),
               "mfma_util": round(busy / (gui / XCDS * SIMDS), 4)}
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in c:
            rec["mfma_tflop_per_launch"] = round(c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / n / 1e12, 4)
        if "SQ_BUSY_CYCLES" in c:
            rec["sq_busy_cycles_per_launch"] = round(c["SQ_BUSY_CYCLES"] / n)
        if ndur.get(name):
            ms = dur[name] / ndur[name] * 1e-6
            rec["avg_ms_under_counters"] = round(ms, 4)
            rec["clock_ghz_under_counters"] = round(gui / n / XCDS / (ms * 1e-3) / 1e9, 3)
        out[name] = rec
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"]))
    res = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES of the bench "
                   "command (4 of 28 blocks: per-launch figures do not depend on the block count); mfma_util = busy / "
                   "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = the SDK's MfmaUtil; counter collection serialises kernels and lowers "
                   "the clock a little (profiled runs are a few % slower)", "kernels": out}
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in list(out.items())[:8]:
        print(k, v)


if __name__ == "__main__":
    main()
