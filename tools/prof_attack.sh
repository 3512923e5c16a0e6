#!/bin/bash
# usage (GPU box): tools/prof_attack.sh <geoa3|knn> -> kernel stats of tools/bench_attacks.py <attack>
v=${1:-geoa3}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_attack_$v
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/bench_attacks.py $v > $out/run.log 2>&1
f=$(ls $out/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6, "launches", sum(int(r['Calls']) for r in rows))
for r in rows[:30]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.2f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
PY
