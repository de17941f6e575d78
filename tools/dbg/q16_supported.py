import sys
sys.path.insert(0, 'tacotron2-vae_amd')
import torch, t2v_hip as H
lib = H.load_library()
torch.zeros(1, device='cuda')
for B, T in ((16, 84), (7, 40), (3, 16), (16, 224), (16, 5)):
    print(B, T, 'fwd16', lib.t2v_decoder_train_persist16_supported(B, T), 'bwd16', lib.t2v_decoder_bwd_persist16_supported(B, T),
          'slices', lib.t2v_decoder_bwd_persist16_slices(T), lib.t2v_last_error())
