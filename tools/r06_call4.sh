cd $GRAFT_REPO_ROOT
for sk in 0 4 1; do echo "T2V_X3_SKIP=$sk (1: no split passes, 2: no GEMM, 4: split once per shape+mode (23 calls each))"; T2V_X3_SKIP=$sk timeout 120 python tools/dbg/x3_time.py 2 2>&1 | grep GEMM; done
