"""oracle/reference_runner.py: the staged reference (oracle/_ref/py, git-ignored, made from /root/reference by the committed recipe)
runs its own main.train on CPU - what bench.py reports as cpu_baseline.kind == "reference".  Skipped where nothing is staged."""
import json
import os
import subprocess
import sys

import pytest

from oracle import reference_runner as rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage_recipe_copies_byte_identical_files():
    if not os.path.isdir(rr.REFERENCE):
        pytest.skip("no reference checkout (the GPU box): the stage travels with the snapshot")
    assert rr.stage() and rr.available()
    import hashlib
    for f, h in rr.staged_sha256().items():
        assert hashlib.sha256(open(os.path.join(rr.REFERENCE, f), "rb").read()).hexdigest()[:16] == h, f
    assert set(rr.staged_sha256()) == set(rr.FILES)
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    assert tracked == "", "the staged reference must never be committed"


@pytest.mark.parametrize("config", [1, 2])
def test_reference_train_step_runs_from_the_stage(config):
    if not rr.available():
        pytest.skip("nothing staged under oracle/_ref/py")
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-m", "oracle.reference_runner", "--config", str(config), "--threads", "4", "--seconds", "0.3"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("REFERENCE_JSON ")), None)
    assert line is not None, r.stderr[-1500:]
    res = json.loads(line[len("REFERENCE_JSON "):])
    assert res["steps"] >= 1 and res["ms_per_step"] > 0 and set(res["sha256"]) == set(rr.FILES)
