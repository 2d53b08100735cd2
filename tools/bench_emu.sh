#!/bin/bash
# usage: tools/bench_emu.sh TAG N...  -- what one of N ranks does, on one GPU (bench.py --emulate-world), RG_DEBUG on
TAG=$1; shift
for n in "$@"; do
  RG_DEBUG=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --emulate-world $n > gpurun_out/bench_${TAG}_emu$n.json 2> gpurun_out/bench_${TAG}_emu$n.err || exit $?
  echo "== emulated rank of $n"; python tools/show_bench.py gpurun_out/bench_${TAG}_emu$n.json; grep -E "kept as a log" gpurun_out/bench_${TAG}_emu$n.err | tail -n 1 | cut -c1-300
done
