#!/bin/bash
# usage: tools/rehearse_gloo.sh N SCALE ROUTE -- bench.py --gpus N rehearsed on ONE GPU (gloo, host-staged exchange)
N=$1; SCALE=$2; ROUTE=$3
RAYGRAPH_BENCH_BACKEND=gloo RG_DEBUG=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus $N --steps 2 --warmup 1 --scale $SCALE --route $ROUTE > gpurun_out/gloo_${N}_${ROUTE}.json 2> gpurun_out/gloo_${N}_${ROUTE}.err
rc=$?
tail -n 1 gpurun_out/gloo_${N}_${ROUTE}.json | cut -c1-600
grep -c "kept as a log" gpurun_out/gloo_${N}_${ROUTE}.err
exit $rc
