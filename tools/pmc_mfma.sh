# Separate PMC-only pass (no --stats / sys-trace): matrix-core utilisation of the MFMA-bound kernels.
#   MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE
#   (SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs, MI355X_MICROARCH.md constants table).
# usage: bash tools/pmc_mfma.sh <tag> [--bf16]   -> gpurun_out/<tag>_pmc_mfma[_bf16].json
TAG=${1:-r03}
EXTRA=$2
SUF=""; [ "$EXTRA" = "--bf16" ] && SUF="_bf16"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcm$SUF
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/pmcm$SUF -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-secondary --no-graph $EXTRA > /tmp/pmcm$SUF.log 2>&1
tail -2 /tmp/pmcm$SUF.log | cut -c1-200
F=$(find /tmp/pmcm$SUF -name '*counter_collection.csv' | head -1)
mkdir -p $REPO/gpurun_out
python - "$F" "$EXTRA" > $REPO/gpurun_out/${TAG}_pmc_mfma$SUF.json <<'PY'
import csv, sys, json, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
# round 6: one row per kernel FUNCTION (template arguments and parameters stripped) — no substring keys: k_conv5_fwd_bf16k32 used to be
# counted under k_conv5_fwd_bf16, k_conv5_dw_bf16 under k_conv5_dw
want = re.compile(r'^(k_bwd_persist16|k_dec_train_persist16|k_gemm_\w+|k_conv5_\w+|k_x3_split|k_cx3_split_\w+|k_lstm_\w+|k_dec_train_persist|k_achain_bwd|k_attn_cell_bwd|k_bilstm_\w+|Cijk\w*)$')
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('void ', '')
    base = re.split(r'[<(]', n, 1)[0].strip()
    if want.match(base):
        agg[base][r['Counter_Name']].append(float(r['Counter_Value']))
out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-secondary --no-graph %s (MI355X, separate PMC-only pass, tools/pmc_mfma.sh)" % sys.argv[2],
       "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the CSV reports GRBM_GUI_ACTIVE summed over the 8 XCDs (cross-check: MOPS x 512 / cycles reproduces the event-timed TFLOP/s); mean over the dispatches of a kernel",
       "kernels": {}}
for k, c in agg.items():
    e = {"dispatches": len(next(iter(c.values())))}
    for name, v in c.items():
        e["avg_" + name] = round(sum(v) / len(v), 1)
    if e.get("avg_GRBM_GUI_ACTIVE") and "avg_SQ_VALU_MFMA_BUSY_CYCLES" in e:
        cyc = e["avg_GRBM_GUI_ACTIVE"] / 8.0        # the CSV value is the SUM over the 8 XCDs
        e["kernel_cycles"] = round(cyc, 1)
        e["mfma_busy_frac"] = round(e["avg_SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        fl = (e.get("avg_SQ_INSTS_VALU_MFMA_MOPS_F32", 0) + e.get("avg_SQ_INSTS_VALU_MFMA_MOPS_BF16", 0)) * 512
        e["mfma_flop_per_dispatch"] = int(fl)
        e["tflops_at_2.4GHz"] = round(fl / (cyc / 2.4e9) / 1e12, 1)
        e["mfma_dtype"] = "bf16" if e.get("avg_SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) > e.get("avg_SQ_INSTS_VALU_MFMA_MOPS_F32", 0) else "f32"
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
PY
head -c 1500 $REPO/gpurun_out/${TAG}_pmc_mfma$SUF.json
