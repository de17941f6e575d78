# usage: [PMC_FILTER=name1,name2] bash tools/r06_pmc_py.sh "<counters>" <script.py> [args]  -> per-kernel mean counter values of a python script (separate PMC-only pass)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
CTRS="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pypmc
timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pypmc -o p -- python $R/"$@" > /tmp/pypmc.log 2>&1
echo "rocprofv3 rc=$?"; tail -2 /tmp/pypmc.log | cut -c1-200
F=$(find /tmp/pypmc -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].split('(')[0][-40:]
    key = (n, r.get('Grid_Size', ''))
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    import os
    if not any(x in k[0] for x in os.environ.get('PMC_FILTER', 'x3,gemm,conv5').split(',')):
        continue
    e = {n: sum(v) / len(v) for n, v in c.items()}
    line = '%-40s grid %9s n=%3d ' % (k[0], k[1], len(next(iter(c.values()))))
    for n, v in sorted(e.items()):
        line += ' %s=%.4g' % (n.replace('SQ_', ''), v)
    if 'GRBM_GUI_ACTIVE' in e and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
        line += '  | mfma_busy=%.3f' % (e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * e['GRBM_GUI_ACTIVE'] / 8.0))
    print(line)
PY
