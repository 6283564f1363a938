for c in 1 3 4 5; do python bench.py --config $c --steps 30 --warmup 5 --skip-cpu-baseline 2>/dev/null | tail -1 > /tmp/l.json; python - <<PY
import json
d=json.loads(open('/tmp/l.json').read()); r=d["roofline"]
print("config $c", d["dtype"], round(1e3*d["ms_per_step"],1), "us", r["bound"], round(r["frac"],3), "whole_step", round(r["whole_step"]["frac"],3), d["config"]["launch"][:40])
PY
done
