for m in 0 2 4 6 12 36 28 60 1; do
  for l7 in 2124 2222; do
    r=$(timeout 200 python bench.py --steps 300 --warmup 30 --skip-cpu-baseline --single-dtype --wgrads-late $m --phase-tiles 3124,3124,2118,2118,2118,2118,2118,2124,2122,2124,3124,3214,2118,2124,2222,$l7 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pp=d['roofline']['per_phase_us']
print(round(d['ms_per_step']*1e3,1), [p[3] for p in pp if p[0]==0])")
    echo "late=$m L7=$l7: $r"
  done
done
