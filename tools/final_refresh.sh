set -x
python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; tail -3 gpurun_out/final_tests.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; tail -c 600 gpurun_out/final_bench.json
python tools/bench_attacks.py geoa3 knn cw_curvenet geoa3_curvenet > gpurun_out/final_attacks.log 2>&1; tail -1 gpurun_out/final_attacks.log
for v in geoa3 knn cw_curvenet; do bash tools/prof_attack.sh $v > gpurun_out/final_pa_$v.txt 2>&1; head -3 gpurun_out/final_pa_$v.txt; done
