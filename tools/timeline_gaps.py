"""GPU timeline of the bench step from a rocprofv3 --kernel-trace csv: busy time per stream, idle gaps (dev tool)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'), r.get('Stream_Id', '0')))
ev.sort()
# steady-state window: between the last two fused-adam kernels
adam = [i for i, e in enumerate(ev) if 'multi_tensor_apply' in e[2] and 'adam' in e[2].lower()]
if len(adam) < 3:
    adam = [i for i, e in enumerate(ev) if 'multi_tensor_apply' in e[2]]
i0, i1 = adam[-3], adam[-2]
win = ev[i0 + 1:i1 + 1]
t0, t1 = ev[i0][1], ev[i1][1]
print("step window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(win)))
queues = sorted(set(e[3] for e in win))
for q in queues:
    w = [e for e in win if e[3] == q]
    print("queue %s: %d kernels, busy %.3f ms" % (q, len(w), sum(e[1] - e[0] for e in w) / 1e6))
# union busy over all queues
cur_s, cur_e, busy = None, None, 0
for s, e, *_ in sorted(win):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("union busy %.3f ms, idle %.3f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
# largest idle gaps (no kernel running on any queue)
gaps = []
end = t0
for s, e, name, q, st in sorted(win):
    if s > end:
        gaps.append((s - end, name, q))
    end = max(end, e)
gaps.sort(reverse=True)
print("gap histogram: >20us %d, 5-20us %d, 2-5us %d, <2us %d; sum %.3f ms" % (
    sum(g[0] > 20000 for g in gaps), sum(5000 < g[0] <= 20000 for g in gaps), sum(2000 < g[0] <= 5000 for g in gaps),
    sum(g[0] <= 2000 for g in gaps), sum(g[0] for g in gaps) / 1e6))
for g in gaps[:12]:
    print("  gap %.1f us before %s (queue %s)" % (g[0] / 1e3, g[1][:70], g[2]))
# phase split on the main queue: time from first to last depth kernel of each kind
mainq = max(queues, key=lambda q: sum(1 for e in win if e[3] == q))
for key in ("gru_fwd", "gru_bwd", "gemm_kernel", "lstm_fwd", "lstm_bwd"):
    w = [e for e in win if key in e[2]]
    if w:
        print("%-12s n=%4d busy %.3f ms, span %.3f..%.3f ms" % (key, len(w), sum(e[1] - e[0] for e in w) / 1e6,
                                                              (w[0][0] - t0) / 1e6, (w[-1][1] - t0) / 1e6))
