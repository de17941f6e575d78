bash tools/timeline.sh r04b
T2V_FWD_ORDER=side_first bash tools/timeline.sh r04c
bash tools/timeline.sh r04d --no-graph
python tools/ab_branch_overlap.py 2>&1 | tail -4
