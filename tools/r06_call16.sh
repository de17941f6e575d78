cd $GRAFT_REPO_ROOT
build() { ( cd tacotron2-vae_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $1 -c gemm.hip -o gemm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libt2vae_hip.so ); }
run() { timeout 300 python bench.py --bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-190; }
T2V_BF16_GEMM_PLANES=0 timeout 100 python tools/dbg/bf16_gemm_time.py | grep GEMM
for cfg in "-DGX_NG1=8" "-DGX_NG1=4" "-DGX_NG1=2" "-DGX_NG1=4 -DGX_NB=3"; do
  echo "== $cfg"; build "$cfg"; timeout 100 python tools/dbg/bf16_gemm_time.py | grep GEMM; run
done
echo "== planes off"; T2V_BF16_GEMM_PLANES=0 T2V_DW_GROUPED=0 run
