#!/bin/bash
# Round-4 profile set on one box: rocprofv3 --kernel-trace --stats of (1) the default bench command's headline leg, (2) config 4 with a
# shortened update, (3) the RAD-A2C iteration; then the two PMC traffic passes (scripts/pmc_r4.sh).  Summaries -> gpurun_out/r4_prof/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4_prof
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/head -o p --output-format csv -- python3 bench.py --steps 10 --warmup 2 --configs none --no-cpu-baseline > $OUT/head.log 2>&1
python3 scripts/summarize_rocprof.py $(find $OUT/head -name "*kernel_stats.csv" | head -1) $OUT/r04_kernel_stats.csv 40
rocprofv3 --kernel-trace --stats -d $OUT/c4 -o p --output-format csv -- python3 scripts/prof_c4.py > $OUT/c4.log 2>&1
python3 scripts/summarize_rocprof.py $(find $OUT/c4 -name "*kernel_stats.csv" | head -1) $OUT/r04_config4_kernel_stats.csv 45
rocprofv3 --kernel-trace --stats -d $OUT/a2c -o p --output-format csv -- python3 scripts/prof_a2c.py > $OUT/a2c.log 2>&1
python3 scripts/summarize_rocprof.py $(find $OUT/a2c -name "*kernel_stats.csv" | head -1) $OUT/r04_rada2c_kernel_stats.csv 45
tail -2 $OUT/head.log | cut -c1-300; tail -2 $OUT/c4.log; tail -2 $OUT/a2c.log
find $OUT -name "*.csv" -size +4M -delete
find $OUT -name "*.db" -delete
bash scripts/pmc_r4.sh > $OUT/pmc.log 2>&1; tail -5 $OUT/pmc.log
