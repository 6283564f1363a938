#!/bin/bash
# Round 6, GPU call 1: the full -m gpu tier on the round's first tree (TA3N_HEADS_FIX switch and the VPW = 4 heads instantiation deleted;
# peer / sharded tests moved into the tier), those moved tests 10 x in a row, the driver-protocol bench line, the N > 1 code path in a
# 1-rank group with the exchange probe, and the launcher's refusal of --gpus 2 on a 1-GPU box.
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out; O=gpurun_out/r6s1; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tier.txt 2>&1; echo "gpu tier rc=$?" | tee -a $O/summary.txt; tail -15 $O/gpu_tier.txt >> $O/summary.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python -m pytest tests/test_gpu_peer.py tests/test_gpu_ddp_engine.py tests/test_gpu_rccl.py -m gpu -x -q > $O/moved.$i.txt 2>&1
  echo "moved tests rep $i rc=$? $(tail -1 $O/moved.$i.txt | cut -c1-100)" | tee -a $O/summary.txt
done
python bench.py --steps 20 --warmup 5 > $O/bench_protocol.json 2> $O/bench_protocol.err; echo "bench rc=$?" | tee -a $O/summary.txt
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "bench --gpus 2 on this box rc=$? (must be non-zero)" | tee -a $O/summary.txt; tail -2 $O/bench_gpus2.err >> $O/summary.txt
TA3N_DDP_SELFTEST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --steps 100 --warmup 10 --skip-cpu-baseline > $O/bench_selftest.json 2> $O/bench_selftest.err; echo "selftest bench rc=$?" | tee -a $O/summary.txt
python - <<'PY' | tee -a gpurun_out/r6s1/summary.txt
import json
for f in ("bench_protocol", "bench_selftest"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r6s1/{f}.json") if l.startswith("{")][-1])
        print(f, d["ms_per_step"], d["value"], d.get("value_fresh_batch"), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"),
              (d.get("cpu_baseline") or {}).get("probe_ms_per_step_by_threads"))
        print("   exchange", d["config"].get("exchange"), json.dumps(d["config"].get("exchange_probe"))[:900])
        print("   per_phase", d["roofline"]["per_phase_us"])
        print("   other", d.get("other_arithmetic", {}).get("ms_per_step"), {k: v.get("ms_per_step") for k, v in (d.get("configs") or {}).items()})
    except Exception as ex:
        print(f, "unreadable", ex)
PY
