#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run from the repo root through gpurun):
#   bash profiles/collect.sh <tag>          e.g. r2
# 1. kernel trace + stats of the default bench command (graph-replayed timed region + op-by-op profiled pass)
# 2./3. PMC FETCH_SIZE and WRITE_SIZE in their own passes (--kernel-trace only, as the pool requires), op-by-op launches,
#       config 2 (the headline) and config 5 (--wide-only) in the same passes
# 4. PMC matrix-core pass: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_{BF16,F16,F32}
# Every pass runs under `timeout`: a profiler pass that hangs must not take the box with it.
# Results land in gpurun_out/prof_<tag>/ ; the summaries are copied to profiles/ by hand afterwards.
set -u
TAG=${1:-r6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCARGS="--no-cpu-baseline --no-decode --no-graph --no-wide --steps 3 --warmup 1"
WIDEARGS="--wide-only --steps 2 --warmup 1"
timeout 480 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
for C in fetch:FETCH_SIZE write:WRITE_SIZE; do
  D=${C%%:*}; P=${C##*:}
  timeout 420 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$D -o bench -- python3 $ROOT/bench.py $PMCARGS > /dev/null 2> $OUT/$D.log
  timeout 420 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/${D}_wide -o bench -- python3 $ROOT/bench.py $WIDEARGS > /dev/null 2> $OUT/${D}_wide.log
done
MF="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32"
timeout 420 rocprofv3 --kernel-trace --pmc $MF --output-format csv -d $OUT/mfma -o bench -- python3 $ROOT/bench.py $PMCARGS > /dev/null 2> $OUT/mfma.log
timeout 420 rocprofv3 --kernel-trace --pmc $MF --output-format csv -d $OUT/mfma_wide -o bench -- python3 $ROOT/bench.py $WIDEARGS > /dev/null 2> $OUT/mfma_wide.log
cd $ROOT
one() { find $OUT/$1 -name '*counter_collection.csv' | head -1; }
python3 profiles/summarize_pmc.py traffic $(one fetch) $(one write) $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt
python3 profiles/summarize_pmc.py traffic $(one fetch_wide) $(one write_wide) $OUT/cfg5_hbm_traffic.json > $OUT/cfg5_hbm_traffic.txt
python3 profiles/summarize_pmc.py mfma $(one mfma) $OUT/mfma_busy.json > $OUT/mfma_busy.txt
python3 profiles/summarize_pmc.py mfma $(one mfma_wide) $OUT/cfg5_mfma_busy.json > $OUT/cfg5_mfma_busy.txt
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cp $S $OUT/kernel_stats.csv
DS=$(find $OUT/stats -name '*domain_stats.csv' | head -1)
cp $DS $OUT/domain_stats.csv
# keep the merged output small: drop the raw traces (the per-dispatch csv files are tens of MB)
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*.db' -delete
ls -la $OUT
head -14 $OUT/kernel_stats.csv
cat $OUT/hbm_traffic.txt $OUT/cfg5_hbm_traffic.txt $OUT/mfma_busy.txt $OUT/cfg5_mfma_busy.txt
