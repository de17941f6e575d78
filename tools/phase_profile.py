"""Print the in-kernel phase stamps of k_attn_fwd / k_attn_bwd (cycles between phase boundaries)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, t2v_hip, hparams as HP, model as M
lib = t2v_hip.load_library()
lib.t2v_set_phase_profile.argtypes = [C.c_void_p]
buf = torch.zeros(32, dtype=torch.int64, device='cuda')
lib.t2v_set_phase_profile(C.c_void_p(buf.data_ptr()))
hp = HP.create_hparams(); torch.manual_seed(0)
dec = M.Decoder(hp).cuda().train()
B, T_in, T = 6, 84, 50
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5).requires_grad_(True)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
for _ in range(2):
    mel, gate, al = dec(mem, mels, lens)
    (mel.sum() + gate.sum()).backward()
torch.cuda.synchronize()
v = buf.cpu().tolist()
print('attn_fwd phase cycles:', [v[i + 1] - v[i] for i in range(0, 5)], 'total', v[5] - v[0])
print('attn_fwd entry detail (loads issued, window barrier, location MFMAs, partials reduced, barrier):', [v[i] - v[0] for i in range(6, 11)])
print('attn_bwd phase cycles:', [v[16 + i + 1] - v[16 + i] for i in range(0, 5)], 'total', v[21] - v[16])
print('cell_bwd stamps relative to the attention workgroup start:', [v[24 + i] - v[16] for i in range(5)])
