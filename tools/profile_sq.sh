#!/bin/bash
# usage: [SQ_COUNTERS="..."] tools/profile_sq.sh TAG -- one rocprofv3 --pmc pass of SQ counters over one C2 bench step (own run, no traces)
TAG=$1
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out
timeout -s KILL 240 rocprofv3 --pmc ${SQ_COUNTERS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS} --output-format csv -d $OUT/prof_${TAG}_sq -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-after --no-error-record $BENCH_ARGS > $OUT/prof_${TAG}_sq.log 2>&1
echo "rc=$?"
tail -n 3 $OUT/prof_${TAG}_sq.log | cut -c1-300
find $OUT/prof_${TAG}_sq -name '*.csv' -size +30M -delete
ls -R $OUT/prof_${TAG}_sq | head
