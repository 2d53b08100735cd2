#!/bin/bash
# usage: tools/profile_c2.sh TAG -- rocprofv3 passes over the C2 bench (run on the GPU box through gpurun).
#   pass 1: --kernel-trace --stats                  -> gpurun_out/prof_TAG_trace/
#   pass 2: --pmc FETCH_SIZE   (own run, no traces) -> gpurun_out/prof_TAG_fetch/
#   pass 3: --pmc WRITE_SIZE   (own run, no traces) -> gpurun_out/prof_TAG_write/
# tools/summarize_profile.py turns them into the files kept under profiles/.
TAG=$1
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_trace.log 2>&1 || exit $?
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_${TAG}_fetch.log 2>&1 || exit $?
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_${TAG}_write.log 2>&1 || exit $?
# keep only what the summary needs (the per-dispatch counter files are large)
find $OUT/prof_${TAG}_trace $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write -name '*.csv' -size +20M -delete
ls -R $OUT/prof_${TAG}_trace $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write | head -30
