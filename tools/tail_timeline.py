"""Time line of the last traced step from a rocprofv3 kernel_trace.csv: start (us after the anchor kernel's END), duration,
queue / stream id, grid, name — `python tools/tail_timeline.py trace.csv [anchor-kernel-prefix] [min-us]`.  The anchor defaults to
the reverse-pass kernel: what is listed is the tail of the step (weight gradients, encoder backward, optimiser)."""
import csv, sys
path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_bwd_persist16'
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        q = r.get('Stream_Id') or r.get('Queue_Id') or '?'
        wg = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)
        grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], q, grid // max(wg, 1)))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith('k_clip_adam')]
lo, hi = adam[-2], adam[-1]
step = rows[lo + 1:hi + 1]
anc = [r for r in step if r[2].startswith(anchor)]
t0 = anc[-1][1] if anc else step[0][0]
print("step: %.1f us from the first launch to the end of k_clip_adam; anchor %s ends at %.1f us" % (
    (step[-1][1] - step[0][0]) / 1e3, anchor, (t0 - step[0][0]) / 1e3))
print("%10s %9s %6s %6s  %s" % ("start us", "dur us", "queue", "wgs", "kernel"))
for s, e, n, q, g in step:
    if e <= t0 or (e - s) / 1e3 < min_us:
        continue
    print("%10.1f %9.1f %6s %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, g, n[:90]))
