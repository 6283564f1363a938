"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel dispatch.
usage: python tools/pmc_summary.py <counter_collection.csv> [max_rows] [name_filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
flt = sys.argv[3] if len(sys.argv) > 3 else ""
by = collections.OrderedDict()
for r in rows:
    by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Grid_Size"], r.get("Workgroup_Size", "")), {})[r["Counter_Name"]] = float(r["Counter_Value"])
n = 0
for k, v in by.items():
    if flt and flt not in k[1]:
        continue
    w = v.get("SQ_WAVES", 0.0) or 1.0
    name = k[1].replace("void ", "").replace("ta3n::", "").replace("(anonymous namespace)::", "")[:28]
    cols = []
    for cname, val in sorted(v.items()):
        if cname == "SQ_WAVES":
            cols.append(f"waves={val:.0f}")
        elif cname.startswith("SQ_"):
            cols.append(f"{cname[3:]}={val / w:.0f}/w")
        else:
            cols.append(f"{cname}={val:.0f}")
    print(k[0], name, k[2], k[3], " ".join(cols))
    n += 1
    if n >= limit:
        break
