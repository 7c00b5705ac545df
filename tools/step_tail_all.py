"""Kernels AND memory copies in the last `us` microseconds before an optimizer launch (dev tool).
    python tools/step_tail_all.py <trace dir> [us=600] [step from the end=3]"""
import csv, glob, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from prof_summary import klass

d = sys.argv[1]
span = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        gx, wx = int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s %s" % (r.get("Queue_Id", "?"), klass(r["Kernel_Name"], gx, wx)),
                   r["Kernel_Name"]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s %s bytes" % (r.get("Direction", "?"), r.get("Bytes", "?")), "copy"))
ev.sort(key=lambda e: e[0])
adam = [e for e in ev if "adam_flat_k" in e[3]]
t1 = adam[-back][1]
for s, e, what, _ in ev:
    if t1 - span * 1e3 <= s <= t1 + 50e3:
        print("%9.3f .. %9.3f us  %s" % ((s - t1) / 1e3, (e - t1) / 1e3, what))
