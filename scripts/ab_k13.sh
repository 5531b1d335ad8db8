#!/bin/bash
# A/B of K13 (rs_pfgru_train) on one box: scripts/ab_k13.sh <out> <suffix> ...   ("" = the product library); 16 384 full-length episodes
cd $GRAFT_REPO_ROOT
out=$1; shift
for sfx in "$@"; do
  echo "== variant '${sfx}'" >> $out
  K13_EPISODES=16384 RS_LIB_PATH=$GRAFT_REPO_ROOT/radiation_ppo_amd/lib/librs_hip${sfx}.so python scripts/time_k13.py 2>/dev/null >> $out
done
