#!/bin/bash
# usage: tools/bench_err.sh TAG ERR...  -- C2 bench with substitution errors (sensitivity runs), RG_DEBUG on
TAG=$1; shift
for e in "$@"; do
  RG_DEBUG=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --error $e > gpurun_out/bench_${TAG}_err$e.json 2> gpurun_out/bench_${TAG}_err$e.err || exit $?
  echo "== error $e"; python tools/show_bench.py gpurun_out/bench_${TAG}_err$e.json; grep -E "log probe|partitioned sweep" gpurun_out/bench_${TAG}_err$e.err | tail -n 2 | cut -c1-300
done
