cd $GRAFT_REPO_ROOT
for i in 1 2; do
for mt in 64 32 16; do
echo "T2V_X3_MIN_TILES=$mt"; ( T2V_X3_MIN_TILES=$mt timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-190 )
done; done
T2V_X3_MIN_TILES=16 timeout 300 python tools/dbg/gemm_shapes.py 2>&1 | head -8
