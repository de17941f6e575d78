"""longest individual kernel instances of the last optimiser step in a rocprofv3 kernel trace (with grid sizes)"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', '?')))
rows.sort()
adam = [i for i, r in enumerate(rows) if r[2].startswith('k_clip_adam')]
sel = rows[adam[-2] + 1:adam[-1] + 1]
skip = ('k_lstm_fwd256', 'k_lstm_bwd256', 'k_attn_cell_bwd', 'k_attn_fwd')
sel = [r for r in sel if not any(s in r[2] for s in skip)]
sel.sort(key=lambda r: r[0] - r[1])
for s, e, n, gx, wx in sel[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('%8.1f us  grid %8s  %s' % ((e - s) / 1e3, gx, n[:90]))
