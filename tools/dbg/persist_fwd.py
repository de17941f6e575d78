"""Persistent teacher-forced forward (csrc/decoder_train_persist.hip) vs the launch-per-step forward: arena equality,
gradients through the (shared) backward, time per step.   python tools/dbg/persist_fwd.py B T_in T [p_drop] [lens]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in sys.argv[1:4])
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
ragged = len(sys.argv) > 5
lib = H.load_library()
print('supported:', lib.t2v_decoder_train_persist_supported(B, T_in), flush=True)
hp = HP.create_hparams(); torch.manual_seed(0)
dec = M.Decoder(hp).cuda().train()
dec.p_attention_dropout = dec.p_decoder_dropout = p
g = torch.Generator().manual_seed(1)
mem0 = (torch.randn(B, T_in, 512, generator=g) * 0.5).cuda()
mels = torch.randn(B, 80, T, generator=g).cuda()
lens = torch.tensor([max(1, T_in - 7 * i) for i in range(B)] if ragged else [T_in] * B).cuda()
H.DecoderCore.keep_last = True
names = ('gpre', 'memory', 'pm', 'lengths', 'XS', 'CA', 'CD', 'GA', 'GD', 'QP', 'AL', 'ACUM', 'S')
out = {}
for mode in (False, True):
    H.DecoderCore.persistent = mode
    dec._calls = 0
    mem = mem0.clone().requires_grad_(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    times = []
    for it in range(4):
        dec._calls = 0
        for q in dec.parameters():
            q.grad = None
        mem.grad = None
        torch.cuda.synchronize()
        ev[0].record()
        mel, gate, al = dec(mem, mels, lens)
        ev[1].record()
        torch.cuda.synchronize()
        times.append(ev[0].elapsed_time(ev[1]) * 1e3 / T)
        H.check_async_errors()
    print('mode', H.DecoderCore.last_mode, 'forward (incl. prenet / projection GEMMs) us/step:', ['%.2f' % t for t in times], flush=True)
    keep = H.DecoderCore.last_call[3]
    arena = {n: (keep[i].clone() if torch.is_tensor(keep[i]) else None) for i, n in enumerate(names)}
    (mel.sum() + 0.3 * gate.sum() + (mel * mel).sum() * 0.01).backward()
    torch.cuda.synchronize()
    H.check_async_errors()
    grads = {n: q.grad.clone() for n, q in dec.named_parameters() if q.grad is not None}
    grads['memory'] = mem.grad.clone()
    out[mode] = (mel.detach().clone(), gate.detach().clone(), al.detach().clone(), arena, grads)
a, b = out[False], out[True]
def cmp(x, y):
    d = (x - y).abs().max().item(); s = x.abs().max().item()
    return '%.3e (scale %.3e)' % (d, s)
print('mel', cmp(a[0], b[0]), 'gate', cmp(a[1], b[1]), 'align', cmp(a[2], b[2]))
for n in ('XS', 'CA', 'CD', 'GA', 'GD', 'AL', 'ACUM', 'S'):
    x, y = a[3][n], b[3][n]
    if n == 'XS':
        x, y = x[:T + 2], y[:T + 2]
    bad = torch.isnan(y).sum().item()
    print(' arena', n, cmp(x, y), 'nan in persistent:', bad)
worst = 0.0
for n in a[4]:
    d = (a[4][n] - b[4][n]).abs().max().item() / (a[4][n].abs().max().item() + 1e-30)
    worst = max(worst, d)
    if d > 1e-3:
        print(' grad', n, 'rel diff', d)
print('worst relative gradient difference', worst)
