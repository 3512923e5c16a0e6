#!/bin/bash
# usage (GPU box): tools/prof_sq.sh <python script + args, relative to the repo> <kernel-substring> <tag> [counter set: a|b]
#   -> SQ counters of one kernel (mean per dispatch). Two sets because the SQ block holds ~8 counters per pass.
script=$1; k=$2; tag=${3:-x}; set=${4:-a}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/sq_${tag}_$set
rm -rf $out; mkdir -p $out
if [ "$set" = "a" ]; then ctrs="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU"
else ctrs="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; fi
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $root/$script > $out/run.log 2>&1
python3 - "$out" "$k" <<'PY'
import csv, glob, sys, collections
out, key = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if key in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0] + "|grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
