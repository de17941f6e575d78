"""phase stamps of the persistent training forward (step T/2): L workgroup slots 0..5, T workgroup slots 8..12"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import ctypes as C
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in sys.argv[1:4])
lib = H.load_library()
hp = HP.create_hparams(); torch.manual_seed(0)
dec = M.Decoder(hp).cuda().train()
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
prof = torch.zeros(32, dtype=torch.int64, device='cuda')
H.DecoderCore.persistent = True
with torch.no_grad():
    dec(mem, mels, lens)
    lib.t2v_set_phase_profile(C.c_void_p(prof.data_ptr()))
    dec(mem, mels, lens)
    torch.cuda.synchronize()
    lib.t2v_set_phase_profile(None)
pv = prof.cpu().tolist()
L = pv[0:6]; Tt = pv[8:13]
print('L role (cycles): A(ctx half)+cell+publish %d, D-gemv+publish %d, gather h_att %d, gather h_dec %d, A(h_att half) %d, gather ctx %d  | step %d'
      % (L[1] - L[0], L[2] - L[1], L[3] - L[2], L[4] - L[3], pv[6] - L[4], L[5] - pv[6], L[5] - L[0]))
print('T role: loc + wait h_att %d, query %d, energies+store %d, EX gather %d, softmax %d, context+publish %d | step %d'
      % (Tt[1] - Tt[0], pv[13] - Tt[1], Tt[2] - pv[13], pv[14] - Tt[2], Tt[3] - pv[14], Tt[4] - Tt[3], Tt[4] - Tt[0]))
print('  inside A (cycles from step start): own gemv+row sums done %d, barrier passed %d, cell+publish done %d' % (pv[16] - L[0], pv[17] - L[0], L[1] - L[0]))
print('T start relative to L start (cycles, different CUs: indicative only): %d' % (Tt[0] - L[0]))
print('wall clock (10 ns units): L publishes h_att -> T sees it: %d ; T sees h_att -> T publishes ctx: %d ; T publishes ctx -> L has ctx: %d' % (pv[21] - pv[20], pv[22] - pv[21], pv[23] - pv[22]))
print('  L ctx poll rounds %d nap %d ; T h_att poll rounds %d nap %d' % (pv[24], pv[25], pv[26], pv[27]))
H.check_async_errors()
