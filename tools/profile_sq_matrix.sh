#!/bin/bash
# usage: tools/profile_sq_matrix.sh -- SQ counter passes (issue, then LDS) over one C2 step, error-free and with 1 % errors
set -e
for err in 0 0.01; do
  BENCH_ARGS="--error $err" tools/profile_sq.sh sq_a_$err > /dev/null
  python tools/summarize_sq.py sq_a_$err gpurun_out/sq_a_$err.json > /dev/null
  BENCH_ARGS="--error $err" SQ_COUNTERS="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" tools/profile_sq.sh sq_b_$err > /dev/null
  python tools/summarize_sq.py sq_b_$err gpurun_out/sq_b_$err.json > /dev/null
done
python - <<'P'
import json
for err in ("0", "0.01"):
    a = json.load(open("gpurun_out/sq_a_%s.json" % err)); b = json.load(open("gpurun_out/sq_b_%s.json" % err))
    for k in ("k_aggregate", "k_superkmers_lanes_s", "k_scatter"):
        d = dict(a.get(k, {})); d.update(b.get(k, {}))
        print(err, k, {x: ("%.3g" % v) for x, v in sorted(d.items())})
P
