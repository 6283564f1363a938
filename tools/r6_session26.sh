#!/bin/bash
# Round 6, GPU call 26: the bench lines once more (variants headline+MCD / +DAN / +JAN timed with the collector off, lr 1e-3) - what tools/install_final_evidence.sh installs.
bash tools/r6_session17.sh
