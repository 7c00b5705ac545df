"""The last N kernels of one training step, all queues, from a rocprofv3 --kernel-trace csv (dev tool): start / end relative
to the step's last kernel, queue, kernel class.   python tools/step_tail.py <trace dir> [N=40] [step from the end=3]"""
import csv, glob, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from prof_summary import klass

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
               int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
ev.sort(key=lambda e: e[1])
adam = [i for i, e in enumerate(ev) if "adam_flat_k" in e[2]]
i1 = adam[-back]
t1 = ev[i1][1]
for s, e, name, q, gx, wx in ev[max(0, i1 - n):i1 + 8]:
    print("%9.3f .. %9.3f us  q%-3s %4d wg  %s" % ((s - t1) / 1e3, (e - t1) / 1e3, q, gx // max(wx, 1), klass(name, gx, wx)))
