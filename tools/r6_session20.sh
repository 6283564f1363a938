#!/bin/bash
# Round 6, GPU call 20: the final-evidence collection on the tree with the XCD-aware BatchNorm launches (full tier, kernel stats + trace, PMC traffic, bench lines), then the
# three bench lines once more (steadier cpu_baseline) - what tools/install_final_evidence.sh installs.
bash tools/collect_profiles_r06.sh
bash tools/r6_session17.sh
