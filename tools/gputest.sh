#!/bin/bash
# On the GPU box: the GPU test tier with its log kept under gpurun_out/ and the summary line printed last
#   gpurun --timeout 1500 -- 'bash tools/gputest.sh [pytest args]'
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=6 "$@" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5
exit $rc
