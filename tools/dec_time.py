"""Decoder recurrence timing at the bench shape: us per time step of the forward loop and of the reverse loop
(events on the launch stream around the C loops only; no Python between the launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import ctypes as C
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (6, 84, 400)))
lib = H.load_library()
if os.environ.get('T2V_BF16'):
    H.set_bf16(True)      # bf16_run: bf16 weight streams in the two per-step LSTM kernels
hp = HP.create_hparams(); torch.manual_seed(0)
dec = M.Decoder(hp).cuda().train()
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5).requires_grad_(True)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
H.DecoderCore.keep_last = True
# wrap the two C loops with events
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
orig_f, orig_b = lib.t2v_decoder_train_fwd, lib.t2v_decoder_train_bwd
class Wrap(object):
    def __init__(self, fn, e0, e1): self.fn, self.e0, self.e1 = fn, e0, e1
    def __call__(self, *a):
        self.e0.record(); r = self.fn(*a); self.e1.record(); return r
lib.t2v_decoder_train_fwd = Wrap(orig_f, ev[0], ev[1])
lib.t2v_decoder_train_bwd = Wrap(orig_b, ev[2], ev[3])
res = []
for it in range(6):
    mel, gate, al = dec(mem, mels, lens)
    (mel.sum() + gate.sum()).backward()
    torch.cuda.synchronize()
    res.append((ev[0].elapsed_time(ev[1]) * 1e3 / T, ev[2].elapsed_time(ev[3]) * 1e3 / T))
f = sorted(r[0] for r in res[2:])[len(res[2:]) // 2]; b = sorted(r[1] for r in res[2:])[len(res[2:]) // 2]
print('B=%d T_in=%d T=%d: forward %.2f us/step, reverse %.2f us/step, sum %.2f' % (B, T_in, T, f, b, f + b))
H.check_async_errors()
