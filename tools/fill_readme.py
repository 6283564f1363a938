"""Fill README.md's results table (the @NAME@ placeholders, or the previous numbers between the table markers) from committed bench lines:
usage: python tools/fill_readme.py profiles/r06_bench_driver_protocol.json [more lines of the same protocol ...]
Every cell shows the range over the given lines."""
import json
import re
import sys

lines = [json.loads([l for l in open(f) if l.startswith("{")][-1]) for f in sys.argv[1:]]


def rng(vals, fmt):
    lo, hi = min(vals), max(vals)
    return fmt(lo) if fmt(lo) == fmt(hi) else f"{fmt(lo)}–{fmt(hi)}"


def ms(vals):
    return rng(vals, lambda v: f"{v:.4f}" if v < 1 else f"{v:.1f}")


def vps(vals):
    return rng(vals, lambda v: f"{v / 1e6:.2f} M" if v >= 1e6 else f"{v / 1e3:.1f} k")


rows = {
    "FRESH": ([d["ms_per_step_fresh_batch"] for d in lines], [d["value_fresh_batch"] for d in lines]),
    "HEAD": ([d["ms_per_step"] for d in lines], [d["value"] for d in lines]),
    "F32": ([d["other_arithmetic"]["ms_per_step"] for d in lines], [d["other_arithmetic"]["value"] for d in lines]),
    "X3": ([d["roofline"]["split_arithmetic"]["ms_per_step"] for d in lines], [d["roofline"]["split_arithmetic"]["value"] for d in lines]),
    "C0": ([d["configs"]["configs[0]"]["ms_per_step"] for d in lines], [d["configs"]["configs[0]"]["value"] for d in lines]),
    "C3": ([d["configs"]["configs[3]"]["ms_per_step"] for d in lines], [d["configs"]["configs[3]"]["value"] for d in lines]),
    "C4": ([d["configs"]["configs[4]"]["ms_per_step"] for d in lines], [d["configs"]["configs[4]"]["value"] for d in lines]),
    "CPU": ([d["cpu_baseline"]["ms_per_step"] for d in lines], [d["cpu_baseline"]["value"] for d in lines]),
}
for key, name in (("BN", "headline+AdaBN"), ("MCD", "headline+MCD"), ("DAN", "headline+DAN"), ("JAN", "headline+JAN")):
    got = [d["variants"][name] for d in lines if "ms_per_step" in (d.get("variants") or {}).get(name, {})]
    if got:
        rows[key] = ([g["ms_per_step"] for g in got], [g["value"] for g in got])
s = open("README.md").read()
for k, (m, v) in rows.items():
    s = s.replace(f"@{k}_MS@", ms(m)).replace(f"@{k}_V@", vps(v))
open("README.md", "w").write(s)
print({k: (ms(m), vps(v)) for k, (m, v) in rows.items()})
