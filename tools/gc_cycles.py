"""Which reference cycles does a piece of code leave for Python's cyclic collector?

    from tools.gc_cycles import cycles_of
    report = cycles_of(lambda: step(0), repeat=4)

Runs `fn` `repeat` times under gc.DEBUG_SAVEALL and reports the strongly connected components (size > 1, or self loops) of
the saved garbage: the objects that only the collector could free -- everything else in the garbage merely hangs off them.
Used on the VAE step (bench.py --vae-profile gc): cyclic garbage delays the release of device tensors to the next collection
and costs host time on the thread that issues the launches."""
import collections
import gc


def _name(o):
    t = type(o)
    n = t.__module__ + "." + t.__qualname__
    if t is dict:
        n += "{" + ",".join(sorted(map(str, list(o.keys())[:5]))) + "}"
    elif t is tuple or t is list:
        n += "[%d]" % len(o)
    elif isinstance(o, type):
        n += "<%s>" % o.__name__
    return n


def components(objs):
    ids = {id(o): i for i, o in enumerate(objs)}
    succ = [[ids[id(r)] for r in gc.get_referents(o) if id(r) in ids] for o in objs]
    n = len(objs)
    index, low, on, order, comps = [-1] * n, [0] * n, [False] * n, [], []
    counter = 0
    for root in range(n):
        if index[root] != -1:
            continue
        work = [(root, 0)]
        while work:
            v, k = work.pop()
            if k == 0:
                index[v] = low[v] = counter
                counter += 1
                order.append(v)
                on[v] = True
            descended = False
            for j in range(k, len(succ[v])):
                w = succ[v][j]
                if index[w] == -1:
                    work.append((v, j + 1))
                    work.append((w, 0))
                    descended = True
                    break
                if on[w]:
                    low[v] = min(low[v], index[w])
            if descended:
                continue
            if low[v] == index[v]:
                comp = []
                while True:
                    w = order.pop()
                    on[w] = False
                    comp.append(w)
                    if w == v:
                        break
                if len(comp) > 1 or v in succ[v]:
                    comps.append([objs[i] for i in comp])
            if work:
                p = work[-1][0]
                low[p] = min(low[p], low[v])
    return comps


def cycles_of(fn, repeat=4, top=12):
    fn()
    gc.collect()
    gc.set_debug(gc.DEBUG_SAVEALL)
    del gc.garbage[:]
    try:
        for _ in range(repeat):
            fn()
        gc.collect()
    finally:
        gc.set_debug(0)
    garbage = list(gc.garbage)
    del gc.garbage[:]
    comps = components(garbage)
    shapes = collections.Counter()
    for c in comps:
        shapes[tuple(sorted(collections.Counter(_name(o) for o in c).items()))] += 1
    lines = ["%d objects only the collector could free, in %d cycles (%d objects saved in all)"
             % (sum(len(c) for c in comps), len(comps), len(garbage))]
    for shape, k in shapes.most_common(top):
        lines.append("  %3d x  %s" % (k, "  ".join("%s*%d" % kv for kv in shape)))
    return lines
