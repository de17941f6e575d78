cd $GRAFT_REPO_ROOT
cd tacotron2-vae_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DT2V_X3_ABLATE -c gemm.hip -o gemm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libt2vae_hip.so && cd ../..
for ab in 0 1 2 3 4 5 6 7; do echo "ablate=$ab (1: no global loads, 2: no split/LDS stores, 4: no MFMA)"; T2V_X3_ABLATE=$ab python tools/dbg/x3_time.py 2 2>&1 | grep GEMM; done
echo all shapes; python tools/dbg/x3_time.py 2>&1 | grep GEMM
