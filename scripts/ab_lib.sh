#!/bin/bash
# A/B of library variants on one box: scripts/ab_lib.sh <out> <suffix> [<suffix> ...]   ("" = the product library)
# per variant: K11 step timing (scripts/time_pfgru.py) and the RAD-A2C bench leg
cd $GRAFT_REPO_ROOT
out=$1; shift
for sfx in "$@"; do
  lib=$GRAFT_REPO_ROOT/radiation_ppo_amd/lib/librs_hip${sfx}.so
  echo "== variant '${sfx}'" >> $out
  RS_LIB_PATH=$lib python scripts/time_pfgru.py 2>/dev/null >> $out
  RS_LIB_PATH=$lib python bench.py --steps 3 --warmup 1 --configs a2c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; c=d['configs']['row_f2_rada2c']
print(round(c['value']), round(c['ms_per_step'],1), c['phase_ms'], c['update_split_ms'])" >> $out
done
