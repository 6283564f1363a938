#!/bin/bash
# Stage the reference's two PROGRAM files (main.py, test_models.py - never its modules) into .ref_stage/, a git-ignored scratch
# directory that travels to the GPU box with the gpurun snapshot the way the built .so does, so that
# tests/test_main_dropin.py::test_reference_main_py_trains_on_the_gpu can run the reference's own, unmodified program on the
# MI355X through compat/ (VERDICT r03 item 2).  Nothing under .ref_stage/ is committed; run again after a clean checkout.
set -e
cd "$(dirname "$0")/.."
REF=${TA3N_REFERENCE_DIR:-/root/reference}
mkdir -p .ref_stage
cp "$REF/main.py" "$REF/test_models.py" .ref_stage/
( cd .ref_stage && sha256sum main.py test_models.py > SHA256SUMS )
( cd "$REF" && sha256sum main.py test_models.py ) | diff - .ref_stage/SHA256SUMS && echo "staged (byte-identical to $REF):" && cat .ref_stage/SHA256SUMS
