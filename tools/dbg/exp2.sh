REPO=$(pwd)
for n in 40 120; do
python $REPO/tools/micro/graph_branch_probe.py $n abc 64 2>/dev/null | grep "N="
python $REPO/tools/micro/graph_branch_probe.py $n abc 64 serial 2>/dev/null | grep "N="
done
