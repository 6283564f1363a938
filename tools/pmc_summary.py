"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel dispatch (first step only)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:34], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
n = 0
for k, v in by.items():
    if "gemm" not in k[1]:
        continue
    w = v.get("SQ_WAVES", 1.0)
    wc = v.get("SQ_WAVE_CYCLES", 1.0)
    print(k[0], k[1][12:34], k[2], " ".join(
        f"{name}={val / w:.0f}/wave" if name.startswith("SQ_INSTS") or name == "SQ_VALU_MFMA_BUSY_CYCLES"
        else (f"{name}={val / wc:.2f}" if name.startswith(("SQ_WAIT", "SQ_ACTIVE")) else f"{name}={val:.0f}")
        for name, val in sorted(v.items()) if name not in ("SQ_WAVES",)))
    n += 1
    if n >= int(sys.argv[2]) if len(sys.argv) > 2 else 11:
        break
