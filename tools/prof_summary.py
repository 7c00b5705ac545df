"""Digest of a rocprofv3 --kernel-trace directory (dev tool; the summaries under profiles/ are its output).

    python tools/prof_summary.py <trace dir> [--steps K] [--json out.json] [--top N]

Per kernel CLASS (template arguments and namespaces stripped), with the depth-step kernels split by grid size -- the atom
level (>= 128 row tiles) is a different regime from the small tree-side / decode-step launches and an average over both
says nothing about either -- calls, mean / median duration and share of the kernel time.  With ``--steps K`` the last K
complete training steps are cut out (the optimizer's kernel closes a step) and everything is reported PER STEP, including
the number of launches per step and the busy time per queue.
"""
import argparse, collections, csv, glob, json, re, statistics, sys

DEPTH = ("gru_fwd_a", "gru_bwd_a", "gru_fwd_b", "gru_bwd_b", "lstm_fwd_a", "lstm_bwd_a", "lstm_fwd_b", "lstm_bwd_b")


def klass(name, grid_x, wg_x):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.search(r"([A-Za-z_][\w:]*)\s*(<|\()", n)
    k = m.group(1) if m else n[:48]
    k = k.split("::")[-1]
    if k in DEPTH:
        row_tiles = grid_x // max(wg_x, 1)
        k += "[atom level]" if row_tiles >= 128 else "[small levels]"
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--json", default=None)
    ap.add_argument("--top", type=int, default=28)
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    f = sorted(glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True))[-1]
    ev = []
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                   int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
    ev.sort()
    per_step = 1
    window_ms = None
    if a.steps:
        opt = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2] or "adam_flat_k" in e[2]]
        ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] - i > 8]
        if len(ends) < a.steps + 1:
            sys.exit("only %d optimizer steps in the trace" % len(ends))
        i0, i1 = ends[-a.steps - 1], ends[-1]
        window_ms = (ev[i1][1] - ev[i0][1]) / 1e6
        ev = ev[i0 + 1:i1 + 1]
        per_step = a.steps
    agg = collections.defaultdict(list)
    queues = collections.defaultdict(lambda: [0, 0.0])
    for s, e, name, q, gx, wx in ev:
        agg[klass(name, gx, wx)].append((e - s) / 1e3)
        queues[q][0] += 1
        queues[q][1] += (e - s) / 1e3
    tot = sum(sum(v) for v in agg.values())
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    head = "%s%d launches" % (a.label + ": " if a.label else "", len(ev))
    if a.steps:
        head += " in the last %d steps = %.1f per step; step window %.3f ms; kernel time %.3f ms per step" % (
            a.steps, len(ev) / per_step, window_ms / per_step, tot / 1e3 / per_step)
    print(head)
    print("%-44s %9s %9s %9s %10s %6s" % ("kernel class", "calls" + ("/step" if a.steps else ""), "mean us", "median us",
                                            "ms" + ("/step" if a.steps else ""), "share"))
    for k, v in rows[:a.top]:
        print("%-44s %9.1f %9.2f %9.2f %10.3f %5.1f%%" % (k[:44], len(v) / per_step, statistics.mean(v), statistics.median(v),
                                                          sum(v) / 1e3 / per_step, 100 * sum(v) / tot))
    rest = rows[a.top:]
    if rest:
        print("%-44s %9.1f %9s %9s %10.3f %5.1f%%" % ("(%d more classes)" % len(rest), sum(len(v) for _, v in rest) / per_step, "", "",
                                                       sum(sum(v) for _, v in rest) / 1e3 / per_step,
                                                       100 * sum(sum(v) for _, v in rest) / tot))
    for q in sorted(queues):
        print("queue %s: %.1f launches%s, %.3f ms busy" % (q, queues[q][0] / per_step, " per step" if a.steps else "",
                                                           queues[q][1] / 1e3 / per_step))
    if a.json:
        out = {"launches_per_step": round(len(ev) / per_step, 1), "kernel_ms_per_step": round(tot / 1e3 / per_step, 3),
               "step_window_ms": round(window_ms / per_step, 3) if window_ms else None,
               "classes": {k: {"calls_per_step": round(len(v) / per_step, 1), "mean_us": round(statistics.mean(v), 2),
                               "ms_per_step": round(sum(v) / 1e3 / per_step, 3)} for k, v in rows[:16]}}
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
