#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh <tag>  -> gpurun_out/prof_<tag>/ kernel stats of the default bench run
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --steps 100 --warmup 10 --no-sweep --no-cpu-baseline > $out/bench.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
PY
