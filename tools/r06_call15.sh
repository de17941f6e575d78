cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_bf16_gpu.py -q -m gpu -x 2>&1 | grep -E "^E |assert|Error|FAILED|passed|failed" | head -30 )
