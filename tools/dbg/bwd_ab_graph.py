"""A/B of the decoder reverse pass as HIP-graph replays in ONE process (boxes differ in clocks; eager timing is launch
bound): forward + backward of the Decoder module captured once per engine, replayed alternately."""
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
leaves = [mem] + [p for p in dec.parameters()]

def body(bwd):
    mel, gate, al = dec(mem, mels, lens)
    loss = mel.sum() + gate.sum()
    if bwd:
        return torch.autograd.grad(loss, leaves, allow_unused=True)
    return loss

graphs = {}
side = torch.cuda.Stream()
for key, (mode, bwd) in {'fwd': (False, False), 'launch-per-step': (False, True), 'persistent': (True, True)}.items():
    H.DecoderCore.persistent_bwd = mode
    with torch.cuda.stream(side):
        for _ in range(2):
            body(bwd)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = body(bwd)
    graphs[key] = (g, out)
    if bwd:
        assert H.DecoderCore.last_bwd_mode == key, H.DecoderCore.last_bwd_mode
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {k: [] for k in graphs}
for it in range(8):
    for k, (g, _) in graphs.items():
        ev[0].record(); g.replay(); ev[1].record(); torch.cuda.synchronize()
        if it >= 2:
            res[k].append(ev[0].elapsed_time(ev[1]))
med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
for k in ('launch-per-step', 'persistent'):
    print('%-16s fwd+bwd %.3f ms, backward alone %.3f ms (%.2f us per reverse step incl. its GEMMs)' % (k, med[k], med[k] - med['fwd'], (med[k] - med['fwd']) * 1e3 / T))
print('forward alone %.3f ms' % med['fwd'])
H.check_async_errors()
