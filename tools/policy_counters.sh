#!/bin/bash
# L2 hit rate / matrix-pipe busy counters of the batched policy kernels:  gpurun -- 'bash tools/policy_counters.sh r2 c3 bf16x3'
set -u
TAG=${1:-r2}; SPEC=${2:-c3}; PREC=${3:-bf16x3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for P in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
    i=$((i+1)); d=$OUT/pc_${TAG}_${SPEC}_${PREC}_p$i; rm -rf $d
    (cd $ROOT && PB_KINDS=${PB_KINDS:-softmax16,gaussian,critic} PB_PREC=$PREC timeout 280 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/pbench.py $SPEC > $OUT/pc_${TAG}_${SPEC}_${PREC}_p$i.log 2>&1)
done
cd $ROOT
python - "$OUT" "$TAG" "$SPEC" "$PREC" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag, spec, prec = sys.argv[1:5]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, f"pc_{tag}_{spec}_{prec}_p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp3" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in sorted(d.items()):
        print(f"{k},{c},{len(v)},{sum(v)/len(v):.1f}")
PY
