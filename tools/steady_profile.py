"""Aggregate a rocprofv3 kernel_trace.csv over the last N optimiser steps (steady state only)."""
import csv, sys, collections
path, nsteps = sys.argv[1], int(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith('k_clip_adam')]
lo, hi = adam[-nsteps - 1], adam[-1]
sel = rows[lo + 1:hi + 1]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    agg[n][0] += e - s
    agg[n][1] += 1
wall = (rows[hi][1] - rows[lo][1]) / nsteps / 1e6
busy = sum(v[0] for v in agg.values()) / nsteps / 1e6
print("steady state over %d steps: wall %.3f ms/step, GPU busy %.3f ms/step, %d launches/step" % (nsteps, wall, busy, len(sel) // nsteps))
print("%-64s %8s %10s %9s" % ("kernel", "calls/st", "ms/step", "avg us"))
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-64s %8.1f %10.3f %9.2f" % (n[:64], c / nsteps, t / nsteps / 1e6, t / c / 1e3))

# ---- idle-gap attribution: GPU idle time between consecutive kernels, charged to the kernel that follows
# Gaps of 0.2 ms and more in the middle of the decoder loop are the tracer's own stalls (rocprofv3 drains its buffer every
# few dozen dispatches on this stack; they appear with eager launches and with graph replays alike and vanish without the
# tracer: bench.py's wall clock per step is BELOW the GPU-busy time measured here).  They are reported separately.
gaps = collections.defaultdict(lambda: [0, 0])
big = collections.defaultdict(lambda: [0, 0])
prev_end = None
stall_ns, stall_n = 0, 0
for s_, e_, n_ in sel:
    if prev_end is not None and s_ > prev_end:
        g = s_ - prev_end
        if g >= 200000:
            stall_ns += g
            stall_n += 1
            big[n_[:50]][0] += g
            big[n_[:50]][1] += 1
        elif g > 3000:
            gaps[n_[:50]][0] += g
            gaps[n_[:50]][1] += 1
    prev_end = max(prev_end or 0, e_)
tot = sum(v[0] for v in gaps.values()) / nsteps / 1e6
print("tracer stalls (gaps >= 200 us): %.1f per step, %.3f ms/step; wall without them %.3f ms/step" % (
    stall_n / nsteps, stall_ns / nsteps / 1e6, wall - stall_ns / nsteps / 1e6))
for n_, (t_, c_) in sorted(big.items(), key=lambda kv: -kv[1][0])[:6]:
    print("   gap >= 200 us before %-50s %5.1f /step %8.3f ms/step" % (n_, c_ / nsteps, t_ / nsteps / 1e6))
print("idle gaps 3 us .. 200 us: %.3f ms/step; top followers:" % tot)
for n_, (t_, c_) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print("   %-50s %7.1f gaps/st %8.3f ms/step  avg %7.1f us" % (n_, c_ / nsteps, t_ / nsteps / 1e6, t_ / c_ / 1e3))

# ---- concurrency: time during which >= 2 kernels were running (multi-stream overlap)
ev = []
for s_, e_, n_ in sel:
    ev.append((s_, 1)); ev.append((e_, -1))
ev.sort()
depth, last, over = 0, None, 0
for t_, d_ in ev:
    if depth >= 2 and last is not None:
        over += t_ - last
    depth += d_
    last = t_
print("time with >= 2 kernels in flight: %.3f ms/step" % (over / nsteps / 1e6))

# ---- machine-readable per-kernel in-situ durations (bench.py's roofline rows read this from profiles/)
if len(sys.argv) > 4:
    import json, re
    def short(n):
        m = re.match(r'(?:void )?([A-Za-z_0-9]+)', n)
        return m.group(1) if m else n
    dur = collections.defaultdict(lambda: [0, 0])
    for n, (t, c) in agg.items():
        k = short(n)
        dur[k][0] += t
        dur[k][1] += c
    json.dump({k: round(t / c / 1e3, 3) for k, (t, c) in sorted(dur.items())}, open(sys.argv[4], 'w'), indent=1)

# ---- optional: the launch sequence of the LAST steady step (env T2V_PROFILE_SEQ=<file>): order, name, grid, duration, gap
import os
if os.environ.get('T2V_PROFILE_SEQ'):
    full = []
    with open(path) as f:
        for r in csv.DictReader(f):
            full.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                         r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?'), r.get('Workgroup_Size_X', '?')))
    full.sort()
    ad = [i for i, r in enumerate(full) if r[2].startswith('k_clip_adam')]
    with open(os.environ['T2V_PROFILE_SEQ'], 'w') as out:
        prev = None
        for s_, e_, n_, gx, gy, gz, wx in full[ad[-2] + 1:ad[-1] + 1]:
            try:
                wgs = int(gx) * int(gy) * int(gz) // max(1, int(wx))
            except ValueError:
                wgs = -1
            out.write("%8.1f us  gap %6.1f  wgs %6d x %4s  %s\n" % ((e_ - s_) / 1e3, (s_ - prev) / 1e3 if prev else 0.0, wgs, wx, n_[:90]))
            prev = e_
