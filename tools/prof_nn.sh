#!/bin/bash
# usage (GPU box): tools/prof_nn.sh <tag> -> gpurun_out/prof_nn_<tag>/kernel_stats.csv : per-kernel durations of tools/bench_nn.py
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_nn_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/bench_nn.py > $out/bench.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
t=$(ls $out/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0][:60]
    key = (name, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    v = sorted(acc[k])
    print(f"{k[0]:60s} grid {k[1]:>7s} {k[2]:>4s} {k[3]:>3s} calls {len(v):4d} median_us {v[len(v)//2]:9.2f} min {v[0]:9.2f}")
PY
