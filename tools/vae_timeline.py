"""Coarse per-queue timeline of ONE training step from a rocprofv3 --kernel-trace csv (dev tool).

    python tools/vae_timeline.py <trace dir> [step index from the end, default 3]

Kernels of a queue that follow each other with gaps < 40 us are merged into runs; every run prints as
[start .. end ms] launches, busy ms, dominant kernel classes.  Shows which chain the step is waiting for."""
import collections, csv, glob, re, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from prof_summary import klass

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
               int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
ev.sort()
opt = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2] or "adam_flat_k" in e[2]]
ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] - i > 8]
i0, i1 = ends[-back - 1], ends[-back]
t0 = ev[i0][1]
win = ev[i0 + 1:i1 + 1]
print("step window %.3f ms, %d launches" % ((ev[i1][1] - t0) / 1e6, len(win)))
for q in sorted(set(e[3] for e in win)):
    runs, cur = [], None
    for s, e, name, qq, gx, wx in win:
        if qq != q:
            continue
        k = klass(name, gx, wx)
        if cur is None or s - cur["end"] > 40000:
            cur = {"start": s, "end": e, "n": 0, "busy": 0, "k": collections.Counter()}
            runs.append(cur)
        cur["end"] = max(cur["end"], e)
        cur["n"] += 1
        cur["busy"] += e - s
        cur["k"][k] += e - s
    print("queue %s:" % q)
    for r in runs:
        top = ", ".join("%s %.0f%%" % (k, 100 * v / max(r["busy"], 1)) for k, v in r["k"].most_common(3))
        print("   [%7.3f .. %7.3f] %4d launches, busy %6.3f ms  %s" % ((r["start"] - t0) / 1e6, (r["end"] - t0) / 1e6, r["n"],
                                                                       r["busy"] / 1e6, top))
