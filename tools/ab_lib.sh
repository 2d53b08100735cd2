#!/bin/bash
# usage: tools/ab_lib.sh TAG [bench arguments] -- the C2 bench with ray_amd/libraygraph.so and with ray_amd/libraygraph_prev.so (a build of
# an earlier commit put there by hand), alternating, twice, on one box: differences between boxes are larger than most changes
TAG=$1; shift
cp ray_amd/libraygraph.so /tmp/rg_cur.so || exit 1
for round in 1 2; do
  for which in cur prev; do
    if [ $which = cur ]; then cp /tmp/rg_cur.so ray_amd/libraygraph.so; else cp ray_amd/libraygraph_prev.so ray_amd/libraygraph.so; fi
    timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-after --no-error-record "$@" > gpurun_out/abl_${TAG}_$which$round.json 2> gpurun_out/abl_${TAG}_$which$round.err || { cp /tmp/rg_cur.so ray_amd/libraygraph.so; exit 1; }
    echo "== $which $round"; python tools/show_bench.py gpurun_out/abl_${TAG}_$which$round.json | sed -n 1,5p; grep -o '"host_syncs_per_step": [0-9.]*' gpurun_out/abl_${TAG}_$which$round.json
  done
done
cp /tmp/rg_cur.so ray_amd/libraygraph.so
