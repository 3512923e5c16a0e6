#!/bin/bash
# usage (GPU box): tools/prof_trace.sh <geoa3|knn|cw_curvenet|geoa3_curvenet> [marker] -> one iteration's kernel timeline
v=${1:-knn}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/trace_$v
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/tools/bench_attacks.py $v > $out/run.log 2>&1
f=$(ls -S $out/*/*kernel_trace.csv | head -1)
python3 $root/tools/trace_iter.py $f ${2:-adam_clip} > $root/gpurun_out/trace_$v.txt
rm -rf $out
