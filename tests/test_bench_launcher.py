"""bench.py --gpus N must be impossible to get wrong (the reference needs no launcher: main.py:79).

* bare `python bench.py --gpus 2` (no RANK in the environment) starts its own two ranks under torch.distributed.run and both
  come up (TA3N_BENCH_LAUNCH_TEST=1: gloo handshake on CPU instead of the engine - no GPU in this container);
* on a box with fewer GPUs than ranks it exits non-zero with a message, before spawning anything;
* a job whose WORLD_SIZE differs from --gpus (the old silent 1-rank run) is an error, in both directions."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    env.update(kw)
    return env


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{"):
            out.append(json.loads(ln))
    return out


def test_bare_invocation_starts_its_own_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=_env(TA3N_BENCH_LAUNCH_TEST="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout           # rank 0 prints ONE line
    line = lines[0]
    assert line["launch_test"] is True and line["value"] is None      # cannot be mistaken for a measurement
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["config"]["parallelism"] == "dp2"
    assert "torch.distributed.run" in r.stderr                        # the launcher says what it starts


def test_fewer_gpus_than_ranks_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "visible GPU" in r.stderr
    assert _json_lines(r.stdout) == []


def test_world_size_must_equal_gpus():
    # a caller's launcher that brought up ONE rank for --gpus 2 (what round 5's bench ran silently as n_gpus: 1)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=_env(TA3N_BENCH_LAUNCH_TEST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    # ... and two ranks for --gpus 1
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=_env(TA3N_BENCH_LAUNCH_TEST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_single_rank_launch_test_line():
    r = subprocess.run([sys.executable, BENCH], cwd=ROOT, env=_env(TA3N_BENCH_LAUNCH_TEST="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]["n_gpus"] == 1
