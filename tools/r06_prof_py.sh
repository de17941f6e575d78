# usage: bash tools/r06_prof_py.sh <script.py> [args]  -> per-kernel durations (grouped by name and grid) of a python script under rocprofv3
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pyprof
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pyprof -o p -- python $R/"$@" > /tmp/pyprof.log 2>&1; tail -5 /tmp/pyprof.log
echo "rocprofv3 rc=$?"
T=$(find /tmp/pyprof -name '*kernel_trace.csv' | head -1)
python - "$T" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0]
    key = (n[-48:], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print('%-50s grid %7s x %5s x %3s  n=%4d  avg %8.1f us  min %8.1f' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), min(v)))
PY
