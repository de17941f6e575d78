cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /tmp/pmc1 -o p -- python /root/repo/tools/conv_bench.py > /tmp/pmc1.log 2>&1
tail -3 /tmp/pmc1.log
F=$(find /tmp/pmc1 -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'conv5' in r['Kernel_Name'] or 'conv_gemm' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:40], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
