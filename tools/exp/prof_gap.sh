#!/bin/bash
# EXPERIMENT (GPU box): kernel + memory-copy + HIP runtime traces of the CW loop on CurveNet, to see what sits in the
# idle gap at the start of every iteration.
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/prof_gap
mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $out -- python3 $root/tools/bench_attacks.py cw_curvenet > $out/run.log 2>&1
ls -la $out/*/ | head -20
