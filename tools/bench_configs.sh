#!/bin/bash
# usage: tools/bench_configs.sh TAG -- the other BASELINE configurations at sizes one GPU holds (timing evidence; parity is tested scaled down)
TAG=$1
run() {  # name workload scale
  RG_DEBUG=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $2 --scale $3 > gpurun_out/bench_${TAG}_$1.json 2> gpurun_out/bench_${TAG}_$1.err || exit $?
  echo "== $1 ($2 scale $3)"; python tools/show_bench.py gpurun_out/bench_${TAG}_$1.json | head -n 8
}
run C1 C1_ecoli_50x_100bp_k31 1.0
run C3 C3_500Mbp_40x_150bp_k63 0.2
run C5 C5_3.1Gbp_30x_150bp_k31_1pct 0.03
