#!/bin/bash
# usage: tools/ab_variants.sh NAME...  -- C2 bench with variants/NAME.so in place of the library, same box, one after the other
for v in "$@"; do
  cp variants/$v.so ray_amd/libraygraph.so || exit 1
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || exit $?
  echo "== $v"; python tools/show_bench.py gpurun_out/ab_$v.json
done
