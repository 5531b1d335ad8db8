#!/bin/bash
# A/B of the CNN trunk kernels on one box: lib/librs_hip_cnnbase.so (rs_cnn.hip of the previous commit, built by hand) vs the current library
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "== base"; RS_LIB_PATH=$PWD/radiation_ppo_amd/lib/librs_hip_cnnbase.so python scripts/time_cnn.py 2>&1 | grep "cin="
  echo "== new";  python scripts/time_cnn.py 2>&1 | grep "cin="
done
