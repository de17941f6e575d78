cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
for i in 1 2; do
for pr in "0 0" "-1 0" "-2 0"; do set -- $pr
echo "MAIN=$1 SIDE=$2"
T2V_MAIN_PRIO=$1 T2V_SIDE_PRIO=$2 timeout 300 python bench.py --bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c88-200
T2V_MAIN_PRIO=$1 T2V_SIDE_PRIO=$2 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c88-200
done; done
