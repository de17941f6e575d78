"""phase stamps of the persistent reverse pass (step T/2): L workgroup slots 0..6, T workgroup 0 slots 8..11"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import ctypes as C
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in sys.argv[1:4])
lib = H.load_library()
hp = HP.create_hparams(); torch.manual_seed(0)
M.drop_rate = 0.0
dec = M.Decoder(hp).cuda().train()
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5).requires_grad_(True)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
prof = torch.zeros(32, dtype=torch.int64, device='cuda')
H.DecoderCore.keep_last = True
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(3):
    mel, gate, al = dec(mem, mels, lens)
    loss = mel.sum() + gate.sum()
    if it == 2:
        lib.t2v_set_phase_profile(C.c_void_p(prof.data_ptr()))
    loss.backward()
    torch.cuda.synchronize()
lib.t2v_set_phase_profile(None)
for it in range(3):
    ev[0].record(); H.replay_persistent_backward(); ev[1].record(); torch.cuda.synchronize()
    print('replay: %.2f us per reverse step' % (ev[0].elapsed_time(ev[1]) * 1e3 / T))
pv = prof.cpu().tolist()
L = pv[0:7]; Tt = pv[8:12]
print('L role (cycles): gather dga %d, A-GEMV+sums %d, publish dctx %d, cell D + gather dgd + D-GEMV %d, dq gather+sync %d, Wq^T dq + cell A + publish %d | step %d'
      % (L[1] - L[0], L[2] - L[1], L[3] - L[2], L[4] - L[3], L[5] - L[4], L[6] - L[5], L[6] - L[0]))
print('T role (cycles): prefetch + wait dctx %d, softmax/tanh backward -> dq published %d, location backward + window partials %d | step %d'
      % (Tt[1] - Tt[0], Tt[2] - Tt[1], Tt[3] - Tt[2], Tt[3] - Tt[0]))
H.check_async_errors()
