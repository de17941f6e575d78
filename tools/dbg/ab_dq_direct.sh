cd $GRAFT_REPO_ROOT
for cfg in "-DPBA_DQ_DIRECT=0" "-DPBA_DQ_DIRECT=1"; do
 ( cd tacotron2-vae_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $cfg -c decoder_train_bwd_persist.hip -o decoder_train_bwd_persist.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libt2vae_hip.so )
 echo "== $cfg"; timeout 200 python tools/dbg/persist_bwd_prof.py 6 84 400 2>&1 | grep -E "^replay|attention_rnn role" | head -4
 timeout 300 python -m pytest tests/test_decoder_persist_train_gpu.py -q -m gpu -x 2>&1 | tail -1
 timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-secondary --eager-steps 0 2>&1 | tail -1 | cut -c1-190
done
