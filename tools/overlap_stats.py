"""How much do kernels from different queues overlap? (rocprofv3 --kernel-trace csv; dev tool)"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')) for r in rows)
ev = ev[len(ev) // 2:]                      # steady state
t0, t1 = ev[0][0], max(e[1] for e in ev)
by_q = {}
for s, e, n, q in ev:
    by_q.setdefault(q, []).append((s, e))
tot = 0
for q, lst in sorted(by_q.items()):
    busy = sum(e - s for s, e in lst)
    tot += busy
    print("queue %s: %d kernels, busy %.2f ms" % (q, len(lst), busy / 1e6))
cur_s = cur_e = None
union = 0
for s, e, *_ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("window %.2f ms, sum busy %.2f ms, union busy %.2f ms, overlap factor %.2f" % ((t1 - t0) / 1e6, tot / 1e6, union / 1e6, tot / union))
