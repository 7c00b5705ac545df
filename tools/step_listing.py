"""Every launch of ONE steady-state step, in start order, from a rocprofv3 --kernel-trace csv (dev tool):
queue, start (us after the step's first launch), duration, idle time of ITS queue in front of it, grid / workgroup, name.

    python tools/step_listing.py /tmp/prof_dir [step index, default 30] [queue id to list, default all]
"""
import csv
import glob
import re
import sys

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'),
             int(r.get('Grid_Size_X', 0) or 0) * max(1, int(r.get('Grid_Size_Y', 1) or 1)) * max(1, int(r.get('Grid_Size_Z', 1) or 1)),
             int(r.get('Workgroup_Size_X', 0) or 0) * max(1, int(r.get('Workgroup_Size_Y', 1) or 1)),
             int(r.get('LDS_Block_Size', 0) or 0)) for r in rows)
mt = [i for i, e in enumerate(ev) if 'multi_tensor_apply' in e[2] or 'adam_flat_k' in e[2]]
ends = [i for k, i in enumerate(mt) if k + 1 == len(mt) or mt[k + 1] - i > 8]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
only = sys.argv[3] if len(sys.argv) > 3 else None
i0, i1 = ends[n - 1], ends[n]
win = ev[i0 + 1:i1 + 1]
t0 = win[0][0]
queues = sorted(set(e[3] for e in win))
last_end = {}
print("step %d: %d launches, window %.3f ms; queues %s" % (n, len(win), (win[-1][1] - t0) / 1e6, queues))
print("%5s %9s %8s %8s %7s %6s %7s  %s" % ("queue", "start us", "dur us", "idle us", "wgs", "wg", "lds", "kernel"))
for s, e, name, q, grid, wg, lds in win:
    idle = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = max(e, last_end.get(q, 0))
    if only is not None and q != only:
        continue
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    short = re.sub(r'\(.*', '', short)[:70]
    print("%5s %9.1f %8.1f %8.1f %7d %6d %7d  %s" % (queues.index(q), (s - t0) / 1e3, (e - s) / 1e3, idle, grid // max(wg, 1), wg, lds, short))
