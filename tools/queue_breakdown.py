"""Per-queue kernel time of one steady-state bench step from a rocprofv3 --kernel-trace csv (dev tool)."""
import collections, csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'))
            for r in csv.DictReader(open(f)))
# the optimizer's kernels close a step: the last one of every run of consecutive multi_tensor_apply launches is a boundary
mt = [i for i, e in enumerate(ev) if 'multi_tensor_apply' in e[2]]
ends = [i for k, i in enumerate(mt) if k + 1 == len(mt) or mt[k + 1] - i > 8]
# argv[2]: which step (counted from the start of the run; default 30 = inside bench.py's timed region: 16 warm-up + 30 timed)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
i0, i1 = ends[n - 1], ends[n]
print("step %d of %d, window %.3f ms" % (n, len(ends), (ev[i1][1] - ev[i0][1]) / 1e6))
win = ev[i0 + 1:i1 + 1]
for q in sorted(set(e[3] for e in win)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, name, qq in win:
        if qq != q:
            continue
        m = re.search(r'(\w+)(<|\()', name.replace('(anonymous namespace)::', '').replace('void ', ''))
        k = m.group(1) if m else name[:40]
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    print("queue %s: %d kernels, %.1f us busy" % (q, sum(v[0] for v in agg.values()), tot))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-34s x%3d %8.1f us" % (k[:34], v[0], v[1]))
