#!/bin/bash
# usage: tools/exp_bits.sh  -- C2 at several error rates (development)
for cfg in "0 0.002"; do
  set -- $cfg
  out=gpurun_out/bits_$1_$2.json
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --sweep $1 --error $2 > $out 2> gpurun_out/bits.err || exit $?
  echo "== sweep $1 error $2"; python tools/show_bench.py $out | head -12
done
