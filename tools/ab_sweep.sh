#!/bin/bash
# usage: tools/ab_sweep.sh FLAGS...  -- C2 bench with rg_config.reserved0 = FLAGS, same box, one after the other
for f in "$@"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sweep $f > gpurun_out/ab_sweep_$f.json 2> gpurun_out/ab_sweep_$f.err || exit $?
  echo "== reserved0 $f"; python tools/show_bench.py gpurun_out/ab_sweep_$f.json
done
