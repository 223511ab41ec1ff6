"""Fold a rocprofv3 SQ counter pass into per-kernel wave-state shares.
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ... --output-format csv -d DIR -o x -- python3 tools/kbench.py gemm --rounds 1
    python tools/pmc_sq.py <x_counter_collection.csv>
Prints every counter per launch and as a share of SQ_WAVE_CYCLES (wave-resident cycles summed over the chip)."""
import csv
import sys
from collections import defaultdict

per = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(sys.argv[1], newline="") as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"].split("(")[0]
        per[name][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[name][row["Counter_Name"]] += 1
for name, c in per.items():
    if "at::" in name or "rocclr" in name:
        continue
    n = max(cnt[name].values())
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    parts = []
    for k in sorted(c):
        v = c[k] / n
        parts.append(f"{k} {v:.4g}" + (f" ({c[k] / wc:.3f})" if wc and k != "SQ_WAVE_CYCLES" else ""))
    print(f"{name}  launches {n}:  " + "  ".join(parts))
