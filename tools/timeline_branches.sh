cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof2 -o r -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-decode > /tmp/p2.log 2>&1
KT=$(find /tmp/prof2 -name '*kernel_trace.csv' | head -1)
python - "$KT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_clip_adam')]
lo, hi = adam[-2], adam[-1]
t0 = int(rows[lo]['End_Timestamp'])
for r in rows[lo + 1:hi + 1]:
    n = r['Kernel_Name']
    if any(k in n for k in ('bilstm', 'gru', 'conv2d', 'conv5_fwd<6>', 'conv5_dw<96>')):
        print('%9.1f -> %9.1f us  q=%s stream=%s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
              r.get('Queue_Id'), r.get('Stream_Id'), n[:50]))
PY
