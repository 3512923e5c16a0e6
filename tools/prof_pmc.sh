#!/bin/bash
# usage (GPU box): tools/prof_pmc.sh <tag> -> gpurun_out/pmc_<tag>.json : per-kernel FETCH_SIZE / WRITE_SIZE (KiB, mean
# per dispatch) of the default bench run. Two SEPARATE passes (the TCC block cannot hold both counters at once), with
# --kernel-trace only (no sys/runtime trace domains next to --pmc on this pool).
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_$ctr
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 $root/bench.py --steps 40 --warmup 5 --no-sweep --no-cpu-baseline > $out/bench.log 2>&1
done
python3 - "$root" "$tag" <<'PY'
import csv, glob, json, sys, collections
root, tag = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/gpurun_out/pmc_{tag}_{ctr}/*/*counter_collection.csv")
    acc = collections.defaultdict(list)
    # key = kernel name | total work-items | run: dispatches of one kernel name are split into runs wherever its grid
    # changes, because two problem sizes can share a total grid (bench.py's N=4096 and N=2048 Chamfer points do) and
    # dynamic LDS does not show in LDS_Block_Size
    last_grid, run_of = {}, collections.defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != ctr:
            continue
        name = r["Kernel_Name"].split("(")[0]
        if "nn_" in name and last_grid.get(name) not in (None, r["Grid_Size"]):   # (only the search kernels share grids)
            run_of[name] += 1
        last_grid[name] = r["Grid_Size"]
        acc[f"{name}|grid={r['Grid_Size']}|run={run_of[name]}"].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][ctr] = sum(v) / len(v)
        res[k]["dispatches"] = len(v)
json.dump(res, open(f"{root}/gpurun_out/pmc_{tag}.json", "w"), indent=1, sort_keys=True)
for k in sorted(res):
    if "pc3d" in k:
        print(k, res[k])
PY
