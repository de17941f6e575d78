cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 5 6 7; do echo "T2V_CX3_DBG=$d (1: no DMA in loop, 2: no MFMA, 4: no LDS reads)"; T2V_CX3_DBG=$d timeout 120 python tools/dbg/conv_x3_time.py 2>&1 | grep conv | head -1 | cut -c1-62; done
