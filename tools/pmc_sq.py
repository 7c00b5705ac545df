"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass (dev tool): python tools/pmc_sq.py <dir> [name filter]"""
import collections, csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True))[-1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if flt not in r['Kernel_Name']:
        continue
    key = (r['Kernel_Name'].replace('(anonymous namespace)::', '')[:46], r['Grid_Size'])
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, cs in acc.items():
    print(key[0], 'grid', key[1])
    for c, v in sorted(cs.items()):
        print("   %-32s n=%3d avg %.4g" % (c, len(v), sum(v) / len(v)))
