"""per-kernel average duration and following gap in the graph-replay windows before / after the marker burst"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
burst = [i for i, r in enumerate(rows) if 'vectorized_elementwise' in r[2] or 'elementwise_kernel' in r[2]]
# the marker burst = longest run of consecutive tiny elementwise kernels
best, cur, start = (0, 0), 0, 0
for i in range(1, len(rows)):
    same = ('elementwise' in rows[i][2] and 'elementwise' in rows[i - 1][2])
    if same:
        cur += 1
        if cur > best[0]: best = (cur, i)
    else:
        cur = 0
b_end = best[1]; b_start = b_end - best[0]
print('marker burst rows %d..%d of %d' % (b_start, b_end, len(rows)))
def stats(lo, hi, tag):
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for i in range(lo, hi - 1):
        s, e, n = rows[i]
        agg[n[:40]][0] += e - s; agg[n[:40]][1] += 1; agg[n[:40]][2] += max(0, rows[i + 1][0] - e)
    print(tag)
    for n, (t, c, g) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:5]:
        print('   %-42s n=%6d avg %.2f us, avg gap after %.2f us' % (n, c, t / c / 1e3, g / c / 1e3))
nwin = 19000
stats(b_start - nwin, b_start, 'BEFORE the burst (last %d kernels)' % nwin)
stats(b_end + 2000, b_end + 2000 + nwin, 'AFTER the burst')
