#!/bin/bash
# After `gpurun -- bash tools/collect_profiles_r06.sh`: copy the merged gpurun_out/prof_r06/ files to their committed names under profiles/ and
# refresh README.md's table from the two driver-protocol lines.  Refuses when the traffic measurement is not of the tree's kernel sources.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/prof_r06
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from ta3n_amd.build import source_hash
h = json.load(open("gpurun_out/prof_r06/gemm_traffic.json"))["source_hash"]
assert h == source_hash(), f"evidence is of sources {h}, the tree is {source_hash()}"
print("evidence and tree agree on the kernel sources:", h)
PY
grep -E "passed|failed" $O/gpu_tier.txt | tail -1
cp $O/gemm_traffic.json profiles/gemm_traffic.json
cp $O/bench_driver_protocol_1.json profiles/r06_bench_driver_protocol.json
cp $O/bench_driver_protocol_2.json profiles/r06_bench_driver_protocol_2.json
cp $O/bench.json profiles/r06_bench.json
cp $O/bench_kernel_stats.csv profiles/r06_bench_kernel_stats.csv
cp $O/gaps_driver_protocol.txt profiles/r06_kernel_trace_gaps_driver_protocol.txt
cp $O/bench_selftest.json profiles/r06_ddp_selftest_exchange_probe.json
cp $O/bench_two_ranks_shared_gpu.json profiles/r06_bench_two_ranks_shared_gpu.json
cp $O/pmc_bf16.txt profiles/r06_pmc_fused_step_bf16.txt
cp $O/pmc_f32.txt profiles/r06_pmc_fused_step_f32.txt
cp $O/pmc_f32x3.txt profiles/r06_pmc_fused_step_f32x3.txt
cp $O/pmc_config4_bf16.txt profiles/r06_pmc_fused_step_config4_bf16.txt
(echo "# Round 6, final evidence call (tools/collect_profiles_r06.sh) on the shipped kernel sources: tools/pmc_config.py per-launch counters, headline bf16 and configs[3] bf16, default tile order"; cat $O/pmc_per_launch.txt) > profiles/r06_pmc_per_launch_final.txt
(tail -12 $O/gpu_tier.txt | cut -c1-250; echo; grep -E "^gpu tier|^bench --gpus" $O/summary.txt) > profiles/r06_gpu_tier_final_tail.txt
python - <<'PY'
p = "README.md"
s = open(p).read()
rows = [("FRESH", "**headline as a training loop runs it"), ("HEAD", "| the same with one resident batch"), ("F32", "| same, fp32 MFMA"), ("X3", "| same, fp32-grade split"),
        ("BN", "| headline + `use_bn AdaBN`"), ("C0", "| TemPooling source-only"), ("C3", "| 512+512 videos"), ("C4", "| two-stream 1024-d"), ("CPU", "| CPU path on the box")]
out = []
for ln in s.split("\n"):
    for k, pre in rows:
        if pre in ln and ln.count("|") >= 4:
            cells = ln.split("|")
            cells[-3], cells[-2] = f" @{k}_MS@ ", f" @{k}_V@ "
            ln = "|".join(cells)
    if "| headline + `ens_DA MCD`" in ln and ln.count("|") >= 4:
        cells = ln.split("|")
        cells[-3], cells[-2] = " @MCD_MS@ / @DAN_MS@ / @JAN_MS@ ", " @MCD_V@ / @DAN_V@ / @JAN_V@ "
        ln = "|".join(cells)
    out.append(ln)
open(p, "w").write("\n".join(out))
PY
python tools/fill_readme.py profiles/r06_bench_driver_protocol.json profiles/r06_bench_driver_protocol_2.json
