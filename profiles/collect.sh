#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh <tag>          e.g. r1
# 1. kernel trace + stats of the default bench command (graph-replayed timed region + op-by-op profiled pass)
# 2. PMC FETCH_SIZE and 3. PMC WRITE_SIZE in their own passes (--kernel-trace only, as the pool requires), op-by-op launches
# Every pass runs under `timeout`: a profiler pass that hangs must not take the box with it.
# Results land in gpurun_out/prof_<tag>/ ; the summaries are copied to profiles/ by hand afterwards.
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-decode --no-graph --steps 3 --warmup 1 > /dev/null 2> $OUT/fetch.log
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-decode --no-graph --steps 3 --warmup 1 > /dev/null 2> $OUT/write.log
cd $ROOT
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python3 profiles/summarize_pmc.py $F $W $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cp $S $OUT/kernel_stats.csv
# keep the merged output small: drop the raw traces (the per-dispatch csv files are tens of MB)
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*.db' -delete
ls -la $OUT
head -12 $OUT/kernel_stats.csv
cat $OUT/hbm_traffic.txt
