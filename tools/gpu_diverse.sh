#!/bin/bash
# usage: tools/gpu_diverse.sh TAG -- parity tests, then the error-rich benches (C2 with 1 % / 0.5 % errors, C5's shape, C5's share per GPU)
TAG=$1
timeout -k 10 500 python -u -m pytest tests/test_parity_gpu.py tests/test_golden_vectors.py tests/test_baseline_configs_gpu.py -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/pytest_$TAG.log 2>&1
rc=$?
tail -n 3 gpurun_out/pytest_$TAG.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
run() { local n=$1; shift; RG_DEBUG=1 timeout -k 10 600 python bench.py --no-cpu-baseline --no-after "$@" > gpurun_out/bench_${TAG}_$n.json 2> gpurun_out/bench_${TAG}_$n.err || exit 2; }
run e01 --steps 2 --warmup 1 --error 0.01
run e005 --steps 2 --warmup 1 --error 0.005
run c5 --steps 2 --warmup 1 --workload C5_3.1Gbp_30x_150bp_k31_1pct --scale 0.03
run c5share --steps 1 --warmup 0 --workload C5_3.1Gbp_30x_150bp_k31_1pct --scale 0.125 --emulate-world 8
