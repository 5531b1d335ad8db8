#!/bin/bash
# SQ counters of K9 / K10 (separate rocprofv3 --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_sq_cnn
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/a -o a --output-format csv -- python3 scripts/time_cnn.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM -d $OUT/b -o b --output-format csv -- python3 scripts/time_cnn.py > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES -d $OUT/c -o c --output-format csv -- python3 scripts/time_cnn.py > $OUT/c.log 2>&1
for p in a b c; do python3 scripts/pmc_sq.py $OUT/$p rs_cnn_fwd_kernel\<6\> rs_cnn_bwd_kernel\<6\> rs_cnn_fwd_kernel\<4\> rs_cnn_bwd_kernel\<4\>; done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
