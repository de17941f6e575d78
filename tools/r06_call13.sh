cd $GRAFT_REPO_ROOT
build() { ( cd tacotron2-vae_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $1 -c gemm.hip -o gemm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libt2vae_hip.so ); }
run() { timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-190; }
for cfg in "-DGX_NB=3" "-DGX_NB=2 -DGX_LDS_PAD=512" "-DGX_NB=2" "-DGX_NB=3 -DGX_LDS_PAD=600" "-DGX_NB=3" "-DGX_NB=2 -DGX_LDS_PAD=512"; do
  echo "== $cfg"; build "$cfg"; run
done
build "-DGX_NB=2 -DGX_LDS_PAD=512"; timeout 100 python tools/dbg/x3_time.py 4 | grep GEMM
