# Separate PMC-only pass (no --stats / sys-trace): HBM bytes fetched per launch of the dominant kernels.
# FETCH_SIZE is reported in KB and, on gfx950, counts 64 B per 128-B request for wide coalesced streams:
# bytes = 2 * 1024 * FETCH_SIZE (MI355X_MICROARCH.md, HBM section).
# usage: bash tools/pmc_fetch_size.sh <tag> [--bf16]   -> gpurun_out/<tag>_pmc_fetch_size[_bf16].json
TAG=${1:-r03}
EXTRA=$2
SUF=""; [ "$EXTRA" = "--bf16" ] && SUF="_bf16"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcf$SUF
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcf$SUF -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-secondary --no-graph $EXTRA > /tmp/pmcf$SUF.log 2>&1
F=$(find /tmp/pmcf$SUF -name '*counter_collection.csv' | head -1)
mkdir -p $REPO/gpurun_out
python - "$F" "$EXTRA" > $REPO/gpurun_out/${TAG}_pmc_fetch_size$SUF.json <<'PY'
import csv, sys, json, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] != 'FETCH_SIZE':
        continue
    n = r['Kernel_Name']
    for key in ('k_bwd_persist16', 'k_dec_train_persist16', 'k_q16_fill', 'k_p16_fill', 'k_gemm_x3p', 'k_x3_split', 'k_conv5_x3', 'k_cx3_split', 'k_gemm_bf16_big', 'k_conv5_fwd_bf16', 'k_achain_bwd', 'k_dec_train_persist', 'k_lstm_fwd256', 'k_lstm_bwd256', 'k_attn_fwd', 'k_attn_cell_bwd', 'k_clip_adam', 'k_attn_wgrad_part', 'k_conv5_fwd<5>', 'k_conv5_dw', 'k_gemm_f32_big', 'k_pb_factors', 'k_pb_cellpre', 'k_pb_fill', 'k_bilstm_fwd', 'k_bilstm_bwd'):
        if key in n:
            agg[key].append(float(r['Counter_Value']))
            break
alg = {'k_dec_train_persist16': 886500000, 'k_bwd_persist16': 1352500000, 'k_lstm_fwd256': 67108864, 'k_lstm_bwd256': 67108864, 'k_clip_adam': 462000000, 'k_achain_bwd': 507200000, 'k_dec_train_persist': 335500000}
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-secondary --no-graph %s (MI355X, separate PMC-only pass, tools/pmc_fetch_size.sh; calibration of the x2048 correction for 4-byte / 16-byte sc1 polled loads: profiles/r04_fetch_calib.txt)" % (sys.argv[2] if len(sys.argv) > 2 else ''),
       "unit_note": "FETCH_SIZE is reported in KB and, on gfx950, counts 64 B per 128-byte LINE fetched, whatever the width of the load that asked for it (tools/micro/fetch_calib.hip: 16-byte streams, 4-byte and 16-byte sc1 loads, one 4-byte sc1 word per line all read 2048 x FETCH_SIZE = bytes of lines touched): bytes = 2 * 1024 * FETCH_SIZE",
       "kernels": {}}
for k, v in agg.items():
    e = {"dispatches": len(v), "avg_FETCH_SIZE_KB": round(sum(v) / len(v), 1), "corrected_bytes_per_launch": int(2 * 1024 * sum(v) / len(v))}
    if k in alg:
        e["algorithmic_bytes_per_launch"] = alg[k]
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
PY
cat $REPO/gpurun_out/${TAG}_pmc_fetch_size$SUF.json | head -50
