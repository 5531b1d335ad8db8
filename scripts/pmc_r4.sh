#!/bin/bash
# two separate counter passes (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in their own runs, with --kernel-trace only)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_pmc
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 scripts/prof_r3_kernels.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 scripts/prof_r3_kernels.py > $OUT/write.log 2>&1
python3 scripts/pmc_traffic.py $OUT/fetch $OUT/write $OUT/r04_pmc_traffic.json $OUT/fetch.log > $OUT/summary.txt 2>&1
tail -3 $OUT/fetch.log
cat $OUT/summary.txt
# keep only the small files
find $OUT -name "*.csv" -size +8M -delete
