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
M.drop_rate = 0.0          # Prenet dropout off: its masks are keyed by a per-call counter that differs between the two runs
zero = os.environ.get('ZERO', '')
with torch.no_grad():
    if 'hh' in zero: dec.attention_rnn.weight_hh.zero_()
    if 'ih' in zero: dec.attention_rnn.weight_ih[:, 256:].zero_()
    if 'pre' in zero: dec.attention_rnn.weight_ih[:, :256].zero_()
g = torch.Generator().manual_seed(1)
mem0 = (torch.randn(B, T_in, 512, generator=g) * 0.5).cuda()
mels = torch.randn(B, 80, T, generator=g).cuda()
lens = torch.tensor([max(1, T_in - 7 * i) for i in range(B)] if ragged else [T_in] * B).cuda()
H.DecoderCore.keep_last = True
names = ('gpre', 'memory', 'pm', 'lengths', 'XS', 'CA', 'CD', 'GA', 'GD', 'QP', 'AL', 'ACUM', 'S')
out = {}
for mode, bmode in ((False, False), (True, False), (True, True)):
    H.DecoderCore.persistent = mode
    H.DecoderCore.persistent_bwd = bmode
    dec._calls = 0
    mem = mem0.clone().requires_grad_(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
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
    ev[2].record()
    (mel.sum() + 0.3 * gate.sum() + (mel * mel).sum() * 0.01).backward()
    ev[3].record()
    torch.cuda.synchronize()
    print('   backward mode', H.DecoderCore.last_bwd_mode, 'us/step (incl. all the time-batched GEMMs of the node) %.2f' % (ev[2].elapsed_time(ev[3]) * 1e3 / T), flush=True)
    H.check_async_errors()
    grads = {n: q.grad.clone() for n, q in dec.named_parameters() if q.grad is not None}
    grads['memory'] = mem.grad.clone()
    out[(mode, bmode)] = (mel.detach().clone(), gate.detach().clone(), al.detach().clone(), arena, grads)
a, b = out[(False, False)], out[(True, False)]
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
xa, xb = a[3]['XS'], b[3]['XS']
for t in range(min(T + 2, 6)):
    print('  XS row', t, 'h_att %.3e ctx %.3e h_dec %.3e' % ((xa[t, :, :1024] - xb[t, :, :1024]).abs().max().item(),
          (xa[t, :, 1024:1536] - xb[t, :, 1024:1536]).abs().max().item(), (xa[t, :, 1536:] - xb[t, :, 1536:]).abs().max().item()),
          ' AL %.3e' % ((a[3]['AL'][t] - b[3]['AL'][t]).abs().max().item() if t <= T else -1),
          ' S %.3e' % ((a[3]['S'][t] - b[3]['S'][t]).abs().max().item() if t < T else -1),
          ' CA %.3e' % ((a[3]['CA'][t] - b[3]['CA'][t]).abs().max().item() if t <= T else -1),
          ' GA %.3e' % ((a[3]['GA'][t] - b[3]['GA'][t]).abs().max().item() if t < T else -1))
ga, gb = a[3]['GA'][1], b[3]['GA'][1]          # (B, 4096)
d = (ga - gb).abs()
print('GA[1] err by gate:', [float(d[:, r * 1024:(r + 1) * 1024].max()) for r in range(4)], 'by item:', [float(d[i].max()) for i in range(B)])
du = d.view(B, 4, 1024).amax((0, 1))
bad = torch.nonzero(du > 1e-4).flatten().tolist()
print('bad units: %d of 1024; first %s last %s' % (len(bad), bad[:24], bad[-8:]))
worst = 0.0
for n in a[4]:
    d = (a[4][n] - b[4][n]).abs().max().item() / (a[4][n].abs().max().item() + 1e-30)
    worst = max(worst, d)
    if d > 1e-3:
        print(' grad', n, 'rel diff', d)
print('worst relative gradient difference', worst)
c = out[(True, True)]
worst = 0.0
for n in a[4]:
    d = (a[4][n] - c[4][n]).abs().max().item() / (a[4][n].abs().max().item() + 1e-30)
    worst = max(worst, d)
    if d > 1e-3:
        print(' PERSISTENT-BWD grad', n, 'rel diff', d)
print('persistent backward: worst relative gradient difference', worst)
