"""oracle/reference_runner.py: an explicitly staged, sha256-pinned copy of the reference (oracle/_ref/py, git-ignored, opt-in - never
made by build()) runs its own main.train on CPU - what bench.py reports as cpu_baseline.reference.  Skipped where nothing is staged."""
import json
import os
import shutil
import subprocess
import sys

import pytest

from oracle import reference_runner as rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage_recipe_copies_only_files_that_match_their_pins(tmp_path, monkeypatch):
    if not os.path.isdir(rr.REFERENCE):
        pytest.skip("no reference checkout (the GPU box)")
    monkeypatch.setattr(rr, "STAGE", str(tmp_path / "py"))
    assert not rr.available()
    assert rr.stage() and rr.available()
    assert set(rr.staged_sha256()) == set(rr.FILES)
    import hashlib
    for f in rr.FILES:
        assert hashlib.sha256(open(os.path.join(rr.REFERENCE, f), "rb").read()).hexdigest() == rr.REFERENCE_SHA256[f], f
    # a staged file that was touched afterwards is no longer "available": nothing of the stage would be executed
    victim = tmp_path / "py" / "loss.py"
    os.chmod(victim, 0o644)
    victim.write_text(victim.read_text() + "\n# edited\n")
    assert not rr.available() and "loss.py" not in rr.staged_sha256()
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    assert tracked == "", "the staged reference must never be committed"


def test_stage_refuses_a_checkout_that_differs_and_never_raises(tmp_path, monkeypatch, capsys):
    if not os.path.isdir(rr.REFERENCE):
        pytest.skip("no reference checkout (the GPU box)")
    fake = tmp_path / "ref"
    shutil.copytree(rr.REFERENCE, fake, ignore=shutil.ignore_patterns(".git", "dataset", "*.md", "webpage"))
    with open(fake / "models.py", "a") as fh:
        fh.write("\nimport os  # not the pinned file\n")
    monkeypatch.setattr(rr, "REFERENCE", str(fake))
    monkeypatch.setattr(rr, "STAGE", str(tmp_path / "py"))
    assert rr.stage() is False and not rr.available()
    assert "does not match the pinned sha256" in capsys.readouterr().err
    assert not os.path.exists(tmp_path / "py" / "models.py")
    monkeypatch.setattr(rr, "REFERENCE", str(tmp_path / "nowhere"))      # no checkout at all: False, no exception
    assert rr.stage() is False


def test_build_does_not_stage():
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "reference_runner" not in src.split("def smoke")[0].replace("oracle.reference_runner --stage", "")


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
