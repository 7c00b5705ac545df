"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), gfx950 corrections applied.

MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the
bytes of a wide coalesced (16 B/lane) streaming read -> doubled here; WRITE_SIZE is exact for 16 B/lane stores.
usage: python tools/pmc_summary.py <fetch_dir> <write_dir>
"""
import collections, csv, glob, re, sys


def load(d, counter):
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[-1]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        m = re.search(r'(\w+)(<|\()', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))
        name = m.group(1) if m else r['Kernel_Name'][:40]
        out[(name, r['Grid_Size'])].append(float(r['Counter_Value']))
    return out


fe, wr = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
rows = []
for k in fe:
    f = sum(fe[k]) / len(fe[k]) * 1024 * 2      # KiB -> B, x2 gfx950 correction for wide reads
    w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1) * 1024
    rows.append((f + w, k, len(fe[k]), f, w))
print("%-28s %10s %6s %12s %12s %12s" % ("kernel", "grid", "calls", "read MB", "write MB", "total MB"))
for tot, k, n, f, w in sorted(rows, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-28s %10s %6d %12.3f %12.3f %12.3f" % (k[0][:28], k[1], n, f / 1e6, w / 1e6, tot / 1e6))

# per-kernel average over ALL launches (the same population bench.py's event timing averages over)
import json, os
agg = collections.defaultdict(lambda: [0.0, 0])
for tot, k, n, f, w in rows:
    agg[k[0]][0] += tot * n
    agg[k[0]][1] += n
out = {name: round(v[0] / v[1]) for name, v in agg.items() if re.match(r'(gru|lstm)_(fwd|bwd)_[ab]$', name)}
print(json.dumps(out))
# the atom-level launches only (the launches of a depth kernel that move at least half of that kernel's largest
# per-launch traffic: the atom graph has ~5x the messages of the tree-side levels): what bench.py's roofline.traffic quotes
top = collections.defaultdict(float)
for tot, k, n, f, w in rows:
    top[k[0]] = max(top[k[0]], tot)
big = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for tot, k, n, f, w in rows:
    if re.match(r'(gru|lstm)_(fwd|bwd)_[ab]$', k[0]) and tot >= 0.5 * top[k[0]]:
        b = big[k[0]]
        b[0] += tot * n; b[1] += f * n; b[2] += w * n; b[3] += n
print(json.dumps({"atom_level": {k: {"bytes_per_launch": round(v[0] / v[3]), "read_bytes": round(v[1] / v[3]),
                                     "write_bytes": round(v[2] / v[3]), "launches": v[3]} for k, v in big.items()}}))
