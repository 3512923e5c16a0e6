#!/bin/bash
# cumulative device time of the tower-backward kernel by phase (debug aid: PC3D_BWD_STOP)
cd /tmp && export TMPDIR=/tmp
for s in 1 2 3 4 5 0; do
  export PC3D_BWD_STOP=$s
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwd_$s -- python3 $GRAFT_REPO_ROOT/tools/bench_pointmlp.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_bwd_$s/*/*kernel_trace.csv')[0]
v=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'bwd_kernel' in r['Kernel_Name'] and r['Grid_Size_X' if 'Grid_Size_X' in r else 'Grid_Size']=='8192']
print('stop=$s', len(v), 'avg_us', round(sum(v)/len(v),1), 'min', round(min(v),1))
PY
done
