#!/bin/bash
# usage (GPU box): tools/prof_attack_quick.sh  -> gpurun_out/quick_attacks.log : ms per iteration of the cfg3 / cfg4 / cfg5
# attack loops (tools/bench_attacks.py), no profiler
root=${GRAFT_REPO_ROOT:-$PWD}
python3 $root/tools/bench_attacks.py > $root/gpurun_out/quick_attacks.log 2>&1
tail -20 $root/gpurun_out/quick_attacks.log
