#!/bin/bash
# usage: tools/exp_err_matrix.sh -- C2 with 1 % errors at forced bucket bits, with the 256- and the 512-slot aggregation tables
for bits in 22 23 24; do for big in 0 0x4000000; do
  fl=$(( (bits << 8) | big ))
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-after --no-error-record --error 0.01 --sweep $fl > gpurun_out/errm_${bits}_${big}.json 2> gpurun_out/errm_${bits}_${big}.err || exit $?
  echo "bits $bits big $big"; python tools/show_bench.py gpurun_out/errm_${bits}_${big}.json | head -8
done; done
