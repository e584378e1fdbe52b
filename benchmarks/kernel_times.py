"""Summarise a rocprofv3 kernel trace (sqlite .db or *_kernel_trace.csv) per kernel name and grid."""
import csv, sqlite3, sys, collections
path = sys.argv[1]
rows = collections.defaultdict(list)
if path.endswith(".db"):
    db = sqlite3.connect(path)
    for name, s, e, gx, gy, gz in db.execute("select name, start, end, grid_x, grid_y, grid_z from kernels"):
        rows[(name, gx, gy, gz)].append(e - s)
else:
    for r in csv.DictReader(open(path)):
        rows[(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
flt = sys.argv[2] if len(sys.argv) > 2 else "lyc"
for (name, gx, gy, gz), d in sorted(rows.items()):
    if flt in name:
        d.sort()
        print(f"{name[:64]:64s} grid=({gx//256 if gx%256==0 else gx},{gy},{gz}) n={len(d):4d} med={d[len(d)//2]/1e3:8.1f}us min={d[0]/1e3:8.1f}")
