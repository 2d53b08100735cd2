#!/bin/bash
# usage: tools/gpu_one.sh TAG PYTEST_ARGS...  -- selected GPU tests with RG_DEBUG on; log under gpurun_out/
TAG=$1; shift
RG_DEBUG=1 timeout -k 10 600 python -u -m pytest -m gpu -x -q --timeout 240 --timeout-method thread -s "$@" > gpurun_out/pytest_$TAG.log 2>&1
rc=$?
grep -E "log probe|passed|failed|Error|error" gpurun_out/pytest_$TAG.log | tail -n 30 | cut -c1-250
exit $rc
