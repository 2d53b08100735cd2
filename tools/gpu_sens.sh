#!/bin/bash
# usage: tools/gpu_sens.sh TAG -- GPU tests, then the C2 bench error-free, with 1 % and 0.2 % errors, and C5's shape (no CPU baseline)
TAG=$1
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/pytest_$TAG.log 2>&1
rc=$?
tail -n 5 gpurun_out/pytest_$TAG.log | cut -c1-220
if [ $rc -ne 0 ]; then exit $rc; fi
RG_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit $?
RG_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-after --error 0.01 > gpurun_out/bench_${TAG}_e01.json 2> gpurun_out/bench_${TAG}_e01.err || exit $?
RG_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-after --error 0.002 > gpurun_out/bench_${TAG}_e002.json 2> gpurun_out/bench_${TAG}_e002.err || exit $?
RG_DEBUG=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-after --workload C5_3.1Gbp_30x_150bp_k31_1pct --scale 0.03 > gpurun_out/bench_${TAG}_c5.json 2> gpurun_out/bench_${TAG}_c5.err || exit $?
grep -h "record sample" gpurun_out/bench_${TAG}*.err | sort | uniq -c
