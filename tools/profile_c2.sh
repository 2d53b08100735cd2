#!/bin/bash
# usage: tools/profile_c2.sh TAG -- rocprofv3 passes over the C2 bench (run on the GPU box through gpurun).
#   pass 1: --kernel-trace --stats                  -> gpurun_out/prof_TAG_trace/
#   pass 2: --pmc FETCH_SIZE   (own run, no traces) -> gpurun_out/prof_TAG_fetch/
#   pass 3: --pmc WRITE_SIZE   (own run, no traces) -> gpurun_out/prof_TAG_write/
#   pass 4: --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum (L2 hit rate, atomic requests at L2) -> prof_TAG_l2/
#   pass 5: --pmc TCC_EA0_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum TCP_TOTAL_ATOMIC_WITH_RET_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum -> prof_TAG_atomic/
# BENCH_ARGS: extra arguments for bench.py (e.g. --error 0.01).  tools/summarize_profile.py turns them into the files kept under profiles/.
TAG=$1
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-after --no-error-record $BENCH_ARGS > $OUT/prof_${TAG}_trace.log 2>&1 || exit $?
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-after --no-error-record $BENCH_ARGS > $OUT/prof_${TAG}_fetch.log 2>&1 || exit $?
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_${TAG}_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-after --no-error-record $BENCH_ARGS > $OUT/prof_${TAG}_write.log 2>&1 || exit $?
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d $OUT/prof_${TAG}_l2 -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-after --no-error-record $BENCH_ARGS > $OUT/prof_${TAG}_l2.log 2>&1 || exit $?
timeout -s KILL 300 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum TCP_TOTAL_ATOMIC_WITH_RET_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum --output-format csv -d $OUT/prof_${TAG}_atomic -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-after --no-error-record $BENCH_ARGS > $OUT/prof_${TAG}_atomic.log 2>&1 || exit $?
# keep only what the summary needs (the per-dispatch counter files are large)
find $OUT/prof_${TAG}_trace $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write $OUT/prof_${TAG}_l2 $OUT/prof_${TAG}_atomic -name '*.csv' -size +30M -delete
ls -R $OUT/prof_${TAG}_trace $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write | head -30
