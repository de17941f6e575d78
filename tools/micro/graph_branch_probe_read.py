import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
fills = [i for i, r in enumerate(rows) if 'FillFunctor' in r[2]]
# a replay ends with a run of fills; take the last 3 replays
ends = [i for i in fills if i + 1 >= len(rows) or 'FillFunctor' not in rows[i + 1][2]]
for k in range(-3, 0):
    lo, hi = ends[k - 1] + 1, ends[k]
    t0 = rows[lo][0]
    first, last = {}, {}
    for s, e, n in rows[lo:hi + 1]:
        tag = 'add' if 'CUDAFunctor_add' in n or 'add' in n.lower() and 'Fill' not in n else 'mul' if 'mul' in n.lower() or 'MulFunctor' in n else 'sin' if 'sin' in n.lower() else 'fill' if 'Fill' in n else n[:20]
        first.setdefault(tag, (s - t0) / 1e3)
        last[tag] = (e - t0) / 1e3
    print('replay %d: prev fill end -> first kernel %.1f us; ' % (k, (t0 - rows[lo - 1][1]) / 1e3) + ', '.join('%s first %.1f last-end %.1f' % (t, first[t], last[t]) for t in first))
