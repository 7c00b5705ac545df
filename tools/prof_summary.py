import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:n]:
    print("%-78s calls=%6s avg_us=%9.2f tot_ms=%8.2f %5.1f%%" % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
print("total kernel ms", tot / 1e6)
