# GPU box: the whole GPU suite, the bench line, the attack loops and their kernel statistics, into gpurun_out/final_*.
set -x
python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; rc=$?; tail -3 gpurun_out/final_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || exit 1
tail -c 600 gpurun_out/final_bench.json
python tools/bench_attacks.py geoa3 knn cw_curvenet geoa3_curvenet > gpurun_out/final_attacks.log 2>&1 || exit 1
tail -1 gpurun_out/final_attacks.log
for v in geoa3 knn cw_curvenet; do
  bash tools/prof_attack.sh $v > gpurun_out/final_pa_$v.txt 2>&1 || exit 1
  head -3 gpurun_out/final_pa_$v.txt
done
