#!/bin/bash
# usage (GPU box): tools/prof_any.sh <tag> <python script> [args] -> gpurun_out/prof_<tag>.txt (kernel stats, top 25)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/$@ > $out/run.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
python3 - "$f" > $root/gpurun_out/prof_$tag.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
rm -rf $out
