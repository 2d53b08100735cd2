#!/bin/bash
# usage: tools/ab_sweep2.sh TAG FLAGS -- the C2 bench without and with rg_debug_set FLAGS, alternating, twice
TAG=$1; FL=$2
for round in 1 2; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-after --no-error-record > gpurun_out/ab_${TAG}_base$round.json 2>/dev/null || exit $?
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-after --no-error-record --sweep $FL > gpurun_out/ab_${TAG}_exp$round.json 2>/dev/null || exit $?
done
for f in base1 exp1 base2 exp2; do python -c "
import json,sys;d=json.load(open('gpurun_out/ab_${TAG}_$f.json'));print('$f',round(d['ms_per_step'],2),d['host_syncs_per_step'],round(d['kernels_ms_per_step'],2))"; done
