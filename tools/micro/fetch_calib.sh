# usage (GPU box): bash tools/micro/fetch_calib.sh <tag>  -> gpurun_out/<tag>_fetch_calib.txt
TAG=${1:-r04}
REPO=$(pwd); mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fcal
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fcal -o p -- $REPO/tools/micro/fetch_calib > /tmp/fcal.log 2>&1
F=$(find /tmp/fcal -name '*counter_collection.csv' | head -1)
python - "$F" > $REPO/gpurun_out/${TAG}_fetch_calib.txt <<'PY'
import csv, sys
N = 256 << 20
print("FETCH_SIZE calibration (rocprofv3 --pmc FETCH_SIZE, tools/micro/fetch_calib.hip): every kernel reads 256 MiB of lines once")
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] != 'FETCH_SIZE':
        continue
    kb = float(r['Counter_Value'])
    print("  %-16s FETCH_SIZE %12.1f KB -> x1024 = %7.1f MiB, x2048 = %7.1f MiB  (lines touched: 256 MiB)" % (
        r['Kernel_Name'].split('(')[0], kb, kb * 1024 / 2**20, kb * 2048 / 2**20))
PY
cat $REPO/gpurun_out/${TAG}_fetch_calib.txt
