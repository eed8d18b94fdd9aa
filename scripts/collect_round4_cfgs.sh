#!/bin/bash
# round 4: counter evidence for the configs that had none.  GPU box, repo root.
T=${1:-r4a}
export PROF_TIMEOUT=150
bash scripts/collect_profiles_cfg.sh ${T}_c4 python scripts/profile_cfg.py c4 1000000
bash scripts/collect_profiles_cfg.sh ${T}_c2_1e6 python scripts/profile_cfg.py c2 1000000
bash scripts/collect_profiles_cfg.sh ${T}_c2_1e5 python scripts/profile_cfg.py c2 100000
bash scripts/collect_profiles_cfg.sh ${T}_c5 python scripts/profile_cfg.py c5 1000000
bash scripts/collect_profiles_cfg.sh ${T}_c5b python scripts/profile_cfg.py c5b 1000000
