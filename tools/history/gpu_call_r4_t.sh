#!/bin/bash
# round 4, call T: traffic re-measured on the final sources (plan description gained per-launch flops), then the bench lines with the per-launch table
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 240 python $R/tools/measure_traffic.py $O > $O/traffic_stdout.txt 2>&1 < /dev/null
cp $O/gemm_traffic.json $R/profiles/gemm_traffic.json
cd $R
timeout 150 python bench.py --steps 20 --warmup 5 > $O/bench_driver_protocol.json 2>> $O/bench.err < /dev/null
timeout 200 python bench.py > $O/bench.json 2>> $O/bench.err < /dev/null
tail -n 3 $O/traffic_stdout.txt
python - <<'PY'
import json
for f in ("bench_driver_protocol", "bench"):
    d = json.loads(open(f"gpurun_out/prof_r04t/{f}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print(f, d["ms_per_step"], d["value"], "fresh", r.get("traffic_source", {}).get("fresh"), {k: round(v["ms_per_step"], 4) for k, v in d["configs"].items()})
    print(" headline launches", [(l["tile"], l["us"], l["tflops"]) for l in r.get("gemm_launches", [])])
    print(" configs[3] launches", [(l["tile"], l["blocks_per_wave"], l["us"], l["tflops"], l["frac_of_mfma_peak"]) for l in d["configs"]["configs[3]"].get("gemm_launches", [])])
PY
tail -n 5 $O/bench.err
