"""Developer tool: print the tail of a rocprofv3 kernel trace as a timeline (relative start, duration, queue, thread, kernel, grid).
usage: python tools/timeline.py <kernel_trace.csv> [n_rows] [skip_from_end]"""
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) - n - skip: len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
def short(k):
    k = re.sub(r"\(.*", "", k)
    k = k.replace("dflo::", "").replace("void ", "")
    return k[:58]
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %8.1f  q%-2s s%-3s t%-5s %-58s %s" % (s / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Stream_Id"], r["Thread_Id"], short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
