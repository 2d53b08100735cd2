#!/bin/bash
# usage: tools/gpu_check.sh TAG  -- GPU parity tests, then the C2 bench (no CPU baseline); outputs under gpurun_out/
TAG=$1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/pytest_$TAG.log 2>&1
rc=$?
tail -n 5 gpurun_out/pytest_$TAG.log | cut -c1-220
if [ $rc -ne 0 ]; then exit $rc; fi
RG_DEBUG=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit $?
tail -n 2 gpurun_out/bench_$TAG.err
