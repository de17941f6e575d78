"""Teacher-forced decoder forward + reverse pass per engine (one-launch persistent kernels / launch-per-step loops) at a
list of shapes: ms per call and us per time step, events around the Python calls (T_out = 400: the kernels dominate).
usage: [T2V_BF16=1] dec_engines_time.py B,T_in,T_out [B,T_in,T_out ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, t2v_hip as H, hparams as HP, model as M

shapes = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(6, 84, 400), (6, 300, 400), (6, 555, 400)]
bf16 = bool(os.environ.get('T2V_BF16'))
if bf16:
    H.set_bf16(True)
hp = HP.create_hparams("bf16_run=True" if bf16 else None); torch.manual_seed(0)
dec = M.Decoder(hp).cuda().train()
for B, T_in, T in shapes:
    mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5).requires_grad_(True)
    mels = torch.randn(B, 80, T, device='cuda')
    lens = torch.tensor([max(1, T_in - 5 * i) for i in range(B)], device='cuda')
    for engine in (True, False):
        H.DecoderCore.persistent = engine
        H.DecoderCore.persistent_bwd = engine
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        res = []
        for it in range(5):
            for q in dec.parameters():
                q.grad = None
            mem.grad = None
            ev[0].record()
            mel, gate, al = dec(mem, mels, lens)
            ev[1].record()
            (mel.sum() + gate.sum()).backward()
            ev[2].record()
            torch.cuda.synchronize()
            res.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
        H.check_async_errors()
        f = sorted(r[0] for r in res[1:])[2]; b = sorted(r[1] for r in res[1:])[2]
        print('%s B=%d T_in=%d T_out=%d %-15s fwd %.2f ms bwd %.2f ms (%.1f + %.1f us per step) modes %s / %s (%s, %s)' %
              ('bf16' if bf16 else 'fp32', B, T_in, T, 'persistent' if engine else 'launch-per-step', f, b, f * 1e3 / T, b * 1e3 / T,
               H.DecoderCore.last_mode, H.DecoderCore.last_bwd_mode, H.DecoderCore.last_kernel, H.DecoderCore.last_bwd_kernel), flush=True)
