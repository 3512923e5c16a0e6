#!/bin/bash
# usage (GPU box): tools/prof_fetch.sh <geoa3|knn|cw_curvenet|...> -> per-kernel mean FETCH_SIZE / WRITE_SIZE (raw KiB per dispatch) and
# time of tools/bench_attacks.py <attack>: which kernels pull more from HBM / the fabric than they need (XCD-blind grids, re-reads)
v=${1:-geoa3}
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/fetch_${v}_$ctr
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 $root/tools/bench_attacks.py $v > $out/run.log 2>&1
done
python3 - "$root" "$v" <<'PY'
import csv, glob, sys, collections
root, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = f"{root}/gpurun_out/fetch_{v}_{ctr}"
    tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(glob.glob(f"{d}/*/*kernel_trace.csv")[0]))}
    for r in csv.DictReader(open(glob.glob(f"{d}/*/*counter_collection.csv")[0])):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"].split("(")[0][:70] + "|grid=" + r["Grid_Size"]
        acc[k][ctr].append(float(r["Counter_Value"]))
        t = tr.get(r["Dispatch_Id"])
        if t and ctr == "FETCH_SIZE":
            acc[k]["us"].append((int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3)
rows = []
for k, d in acc.items():
    n = len(d["FETCH_SIZE"])
    f = sum(d["FETCH_SIZE"]) / max(1, n) * 2 * 1024 / 1e6       # MB, x2: 16-byte-per-lane reads on gfx950 (upper bound for narrower loads)
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024 / 1e6
    us = sum(d["us"]) / max(1, len(d["us"]))
    rows.append((us * n, k, n, us, f, w))
rows.sort(reverse=True)
print("kernel | calls | us | fetch MB (x2) | write MB | (fetch + write) GB/s")
for tot, k, n, us, f, w in rows[:40]:
    print(f"{k:95s} {n:6d} {us:8.1f} {f:9.1f} {w:9.1f} {((f + w) / us * 1e3) if us else 0:8.0f}")
PY
