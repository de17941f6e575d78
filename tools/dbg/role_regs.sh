# register pressure of each role of k_achain_bwd<6> compiled on its own (1 = attention slices, 2 = attention_rnn 12+6 columns,
# 3 = attention_rnn 13+7 columns, 4 = decoder_rnn)
cd "$(dirname "$0")/../../tacotron2-vae_amd/csrc"
for r in 1 2 3 4; do
  echo -n "role $r: "
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPBA_ONLY=$r -Wno-unused-function -Rpass-analysis=kernel-resource-usage -c decoder_train_bwd_persist.hip -o /tmp/role.o 2>&1 | grep -A12 "k_achain_bwdILi6" | grep -E " VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize" | sed 's/\[-Rpass[^]]*\]//g; s/.*remark: *//' | tr '\n' ' '
  echo
done
