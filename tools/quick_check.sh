#!/bin/bash
# usage: tools/quick_check.sh TAG -- parity + fuzz tests, then the C2 bench error-free and with 1 % errors (development loop on the GPU box)
TAG=$1
timeout -k 10 400 python -u -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_golden_vectors.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-after --no-error-record --error 0.01 > gpurun_out/qc_${TAG}_err.json 2>/dev/null || exit 1
python tools/show_bench.py gpurun_out/qc_${TAG}_err.json > gpurun_out/qc_${TAG}_err.txt; sed -n 1,5p gpurun_out/qc_${TAG}_err.txt
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-after --no-error-record > gpurun_out/qc_${TAG}.json 2>/dev/null || exit 1
python tools/show_bench.py gpurun_out/qc_${TAG}.json > gpurun_out/qc_${TAG}.txt; sed -n 1,5p gpurun_out/qc_${TAG}.txt
