#!/bin/bash
# SQ issue / stall counters of the step kernel (two rocprofv3 --pmc passes, 8 SQ slots each):
#   gpurun --timeout 900 -- 'bash tools/sq_counters.sh r2a c3'
# writes gpurun_out/<tag>_sq_<spec>.csv (per-kernel averages of every counter)
set -u
TAG=${1:-r2}; SPEC=${2:-c3}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
P2="SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
    i=$((i+1)); d=$OUT/sq_${TAG}_${SPEC}_p$i; rm -rf $d
    (cd $ROOT && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/pmc_run.py $SPEC 30 > $OUT/sq_${TAG}_${SPEC}_p$i.log 2>&1)
done
cd $ROOT
python - "$OUT" "$TAG" "$SPEC" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag, spec = sys.argv[1:4]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, f"sq_{tag}_{spec}_p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "drone_kernel" in r["Kernel_Name"] or "reset_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, f"{tag}_sq_{spec}.csv"), "w") as f:
    f.write("kernel,counter,launches,mean\n")
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            f.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.1f}\n")
print(open(os.path.join(out, f"{tag}_sq_{spec}.csv")).read())
PY
