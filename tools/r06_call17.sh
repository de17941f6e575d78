cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_conv_bn_gpu.py -q -m gpu 2>&1 | tail -3 )
for i in 1 2; do
for cx in 2 0; do
echo "T2V_CONV_X3=$cx (bf16 step)"; ( T2V_CONV_X3=$cx timeout 300 python bench.py --bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-190 )
done; done
