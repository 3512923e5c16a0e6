#!/bin/bash
# usage (GPU box): tools/prof_pmc_sq.sh <victim> <kernel-substring> -> SQ stall breakdown of one kernel in tools/bench_victims.py <victim>
v=${1:-ssg}; k=${2:-group_linear_max}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_sq_$v
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 $root/tools/bench_victims.py $v > $out/run.log 2>&1
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
