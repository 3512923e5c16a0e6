"""Print ONE iteration of an attack loop from a rocprofv3 --kernel-trace CSV: start (us after the previous iteration's
update launch), duration, queue, grid, kernel name — to read the critical path and the gaps. usage: trace_iter.py <csv>
[marker substring, default adam_clip] [which iteration from the end, default 12]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_clip"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]["End_Timestamp"])
print("launches", b - a, "iteration wall us", (int(rows[b]["End_Timestamp"]) - t0) / 1e3)
busy = 0.0
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"].replace("void ", "").replace("pc3d::", "").replace("at::native::", "")[:86]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    print("%9.1f %8.1f  q%-3s g%-10d %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], g, n))
print("sum of kernel durations us", busy)
