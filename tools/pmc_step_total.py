"""HBM bytes per STEP (and per kernel class) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a bench.py run.

    python tools/pmc_step_total.py <fetch_dir> <write_dir> [marker kernel regex = adam_flat_k] [top = 14]

A step is delimited by the kernel that runs exactly once per step (the flat Adam update); the totals over the whole run are
divided by its launch count, so warm-up and timed steps weigh alike (they are the same launches).  gfx950 corrections as in
tools/pmc_summary.py (MI355X_MICROARCH.md, HBM section): KiB -> bytes, FETCH_SIZE doubled.  Last line: one JSON object
(what profiles/pmc_traffic.json keeps)."""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
    out = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        m = re.search(r'([\w:]+)(<|\()', name)
        k = m.group(1) if m else name[:40]
        out[k][0] += float(r['Counter_Value'])
        out[k][1] += 1
    return out


fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
marker = re.compile(sys.argv[3] if len(sys.argv) > 3 else r'adam_flat_k')
top = int(sys.argv[4]) if len(sys.argv) > 4 else 14
# the two passes are separate runs and need not execute the same number of steps (bench.py measures a region again when the
# allocator went to the device inside it): each pass is normalised by ITS OWN count of marker launches
steps_f = sum(v[1] for k, v in fe.items() if marker.search(k))
steps_w = sum(v[1] for k, v in wr.items() if marker.search(k))
if steps_f == 0 or steps_w == 0:
    sys.exit("no launch of the marker kernel in the trace")
rows = []
for k in set(fe) | set(wr):
    rd = fe.get(k, [0.0, 0])[0] * 1024 * 2 / steps_f
    wt = wr.get(k, [0.0, 0])[0] * 1024 / steps_w
    n = fe.get(k, [0, 0])[1] / steps_f if k in fe else wr[k][1] / steps_w
    rows.append((rd + wt, k, n, rd, wt))
rows.sort(reverse=True)
tot_r, tot_w = sum(r[3] for r in rows), sum(r[4] for r in rows)
print("%d / %d steps in the FETCH / WRITE pass (launches of the marker kernel); traffic per step at the L2 <-> fabric boundary "
      "(FETCH_SIZE / WRITE_SIZE: Infinity-Cache hits included): read %.1f MB + written %.1f MB = %.1f MB" % (
          steps_f, steps_w, tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6))
print("%-34s %10s %12s %12s %12s" % ("kernel", "calls/step", "read MB", "write MB", "total MB/step"))
for tot, k, n, rd, wt in rows[:top]:
    print("%-34s %10.1f %12.2f %12.2f %12.2f" % (k[:34], n, rd / 1e6, wt / 1e6, tot / 1e6))
print(json.dumps({"steps_fetch_pass": steps_f, "steps_write_pass": steps_w, "bytes_per_step": round(tot_r + tot_w),
                  "read_bytes_per_step": round(tot_r), "write_bytes_per_step": round(tot_w),
                  "by_kernel_bytes_per_step": {k: round(tot) for tot, k, n, rd, wt in rows[:top]}}))
