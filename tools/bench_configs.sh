#!/bin/bash
# usage: tools/bench_configs.sh TAG -- the other BASELINE configurations at sizes one GPU holds, and C2 with substitution errors
# (timing evidence; parity is tested scaled down)
TAG=$1
run() {  # name, then bench.py arguments
  local name=$1; shift
  RG_DEBUG=1 timeout -k 10 500 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-error-record "$@" > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err || exit $?
  echo "== $name ($*)"; python tools/show_bench.py gpurun_out/bench_${TAG}_$name.json | sed -n 1,7p
}
run C1_whole --workload C1_ecoli_50x_100bp_k31
run C2_error_0.002 --error 0.002
run C2_error_0.01 --error 0.01
run C3_shape_scale0.2 --workload C3_500Mbp_40x_150bp_k63 --scale 0.2
run C4_scale0.25 --workload C4_meta_200sp_lognormal_30x_150bp_k31 --scale 0.25
run C5_shape_scale0.03 --workload C5_3.1Gbp_30x_150bp_k31_1pct --scale 0.03
run C4_whole --workload C4_meta_200sp_lognormal_30x_150bp_k31
run C5_true_share_scale0.125_one_gpu --workload C5_3.1Gbp_30x_150bp_k31_1pct --scale 0.125 --emulate-world 8 --no-after
