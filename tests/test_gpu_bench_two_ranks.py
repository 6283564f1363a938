"""The WHOLE N > 1 path of bench.py on hardware ("make N > 1 impossible to get wrong"): bare `python bench.py --gpus 2`
starts its own two ranks, they pass the rank / device checks, probe every gradient exchange on the real message (MAX over ranks, the
same choice on both), run the timed region on the fastest and rank 0 prints ONE line with n_gpus 2 and an exchange spanning 2 ranks.
The boxes this is built on have ONE GPU, and RCCL does not put two ranks on one device: TA3N_BENCH_SHARED_GPU=1 lets the two ranks share
cuda:0 over gloo - everything but the RCCL calls themselves is the code an 8-GPU run executes (those are covered in a 1-rank group by
tests/test_gpu_rccl.py and by the bench's TA3N_DDP_SELFTEST line)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_gpus_2_runs_two_ranks_probes_the_exchanges_and_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TA3N_BENCH_SHARED_GPU="1", TA3N_PEER_TIMEOUT_S="10", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--skip-cpu-baseline",
                        "--single-dtype", "--no-other-configs"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, r.stdout[-2000:]      # nothing but the line on stdout (RCCL's banner etc. go to stderr)
    d = lines[0]
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["rccl_ranks"] == 2 and "shared_gpu_test" in d
    assert d["config"]["global_batch"] == 2 * (128 + 74) and d["value"] > 0 and d["config"]["finite"] is True
    probe = d["config"]["exchange_probe"]
    cands = probe["candidates"]
    assert set(cands) >= {"none", "allreduce", "allreduce_overlapped", "sharded", "peer"}
    assert "ms_per_step" in cands["allreduce"] and "ms_per_step" in cands["none"]          # the default exchange always measures
    timed = {k: v["ms_per_step"] for k, v in cands.items() if k != "none" and "ms_per_step" in v and "rejected" not in v}
    assert probe["chosen"] == min(timed, key=timed.get) == d["config"]["exchange"]            # the timed region ran on the fastest
    for k, v in cands.items():                                                                   # a candidate that could not run says why
        assert "ms_per_step" in v or "unavailable" in v or "error" in v or "not_probed" in v, (k, v)
    assert d["config"]["collective"]["exposed_us_per_step"] is not None
    assert "launching" in r.stderr and "torch.distributed.run" in r.stderr
