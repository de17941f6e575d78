cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_conv_bn_gpu.py tests/test_refenc_gpu.py -q -m gpu 2>&1 | tail -3 )
for i in 1 2; do
for sp in 1 0; do
echo "T2V_BN_SPLIT=$sp"; ( T2V_BN_SPLIT=$sp timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-200 )
done; done
