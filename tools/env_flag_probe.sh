#!/bin/bash
# Which HIP runtime switches move the per-step time of the C3 step kernel inside a graph replay?
# (developer probe; the product sets none of them)   gpurun -- 'bash tools/env_flag_probe.sh > gpurun_out/env_flag_probe.log 2>&1'
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout 120 python tools/kbench.py c3 2>&1 | grep "us/step"; }
run A=0
run DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_USE_FGS_KERNARG=0
run AMD_OPT_FLUSH=0
run GPU_FLUSH_ON_EXECUTION=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_DYNAMIC_QUEUES=0
run A=0
for lib in "$@"; do run DRONESIM_LIB=$lib; done
