#!/bin/bash
# A/B of the RAD-A2C bench leg on one box: RS_NO_DRAW_PREFETCH=1 (draws inline) vs default (next iteration's draws on a side stream)
cd $GRAFT_REPO_ROOT
for v in 1 "" 1 ""; do
  RS_NO_DRAW_PREFETCH=$v python bench.py --steps 3 --warmup 1 --configs a2c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; c=d['configs']['row_f2_rada2c']
print('noprefetch=$v', round(c['value']), round(c['ms_per_step'],1), c['phase_ms'], c['update_split_ms'])"
done
