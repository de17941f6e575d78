"""Per-kernel time line of the LAST steady training step of a rocprofv3 kernel trace (multi-stream aware):
   start / end offset in us from the end of the previous step's k_clip_adam, queue id, workgroups, kernel name;
   followed by a summary: wall, busy (union), time with >= 2 kernels in flight, launches.
   usage: python tools/timeline.py <kernel_trace.csv> [out.txt]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        try:
            wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1))
        except Exception:
            wgs = -1
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), wgs))
rows.sort()
ad = [i for i, r in enumerate(rows) if r[2].startswith('k_clip_adam')]
lo, hi = ad[-2], ad[-1]
t0 = rows[lo][1]
sel = rows[lo + 1:hi + 1]
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
qs = {}
for s, e, n, q, w in sel:
    qs.setdefault(q, len(qs))
for s, e, n, q, w in sel:
    out.write("%9.1f %9.1f  %7.1f us  q%d  wgs %6d  %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qs[q], w, n[:80]))
ev = []
for s, e, n, q, w in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, over, busy = 0, None, 0, 0
for t, d in ev:
    if last is not None:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
    depth += d
    last = t
out.write("step wall %.3f ms, busy(union) %.3f ms, >=2 kernels in flight %.3f ms, %d launches, queues %d\n" % (
    (rows[hi][1] - t0) / 1e6, busy / 1e6, over / 1e6, len(sel), len(qs)))
