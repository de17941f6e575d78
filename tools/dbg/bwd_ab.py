"""A/B of the decoder reverse pass in ONE process (boxes differ in clocks): launch-per-step vs one-launch persistent,
alternating; prints ms per DecoderCore.backward (whole backward of the Decoder module, incl. its GEMMs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (6, 84, 400)
hp = HP.create_hparams(); torch.manual_seed(0)
dec = M.Decoder(hp).cuda().train()
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5).requires_grad_(True)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {False: [], True: []}
for it in range(12):
    mode = bool(it & 1)
    H.DecoderCore.persistent_bwd = mode
    mel, gate, al = dec(mem, mels, lens)
    loss = mel.sum() + gate.sum()
    torch.cuda.synchronize()
    ev[0].record(); loss.backward(); ev[1].record(); torch.cuda.synchronize()
    if it >= 4:
        res[mode].append(ev[0].elapsed_time(ev[1]))
    assert H.DecoderCore.last_bwd_mode == ('persistent' if mode else 'launch-per-step'), H.DecoderCore.last_bwd_mode
for mode in (False, True):
    v = sorted(res[mode])
    print('%-16s backward of the Decoder module: min %.3f ms, median %.3f ms' % ('persistent' if mode else 'launch-per-step', v[0], v[len(v) // 2]))
H.check_async_errors()
