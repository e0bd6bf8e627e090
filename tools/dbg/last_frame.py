"""Print the launches of the last frame in a rocprofv3 kernel trace (start ms, duration ms, kernel).  usage: last_frame.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.3f %8.3f  %s  grid %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Kernel_Name"].split("(")[0].replace("rtx::", "")[:50], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
