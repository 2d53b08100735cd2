#!/bin/bash
# usage: tools/ab_env.sh TAG VAR=VALUE ... -- the C2 bench (3 steps) without and with the given environment variables, alternating, twice
TAG=$1; shift
for round in 1 2; do
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-after --no-error-record > gpurun_out/ab_${TAG}_base$round.json 2> gpurun_out/ab_${TAG}_base$round.err || exit $?
  env "$@" timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-after --no-error-record > gpurun_out/ab_${TAG}_exp$round.json 2> gpurun_out/ab_${TAG}_exp$round.err || exit $?
done
for f in gpurun_out/ab_${TAG}_base1.json gpurun_out/ab_${TAG}_exp1.json gpurun_out/ab_${TAG}_base2.json gpurun_out/ab_${TAG}_exp2.json; do echo $f; python tools/show_bench.py $f | head -6; done
