#!/bin/bash
# Round 6, GPU call 25: final evidence on the final sources - full -m gpu tier, kernel stats + trace, PMC traffic, bench lines (collect_profiles_r06.sh), the three bench
# lines once more with the steadier cpu_baseline (r6_session17.sh), the experiments tier on the experiments build (r6_session18.sh).
bash tools/collect_profiles_r06.sh
bash tools/r6_session17.sh
bash tools/r6_session18.sh
