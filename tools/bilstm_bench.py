"""Time the encoder BiLSTM forward / backward (persistent cooperative kernels) at the bench shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd'))
import torch, t2v_hip
B, T = 6, 84
torch.manual_seed(0)
lstm = torch.nn.LSTM(512, 256, 1, batch_first=True, bidirectional=True).cuda()
x = torch.randn(B, T, 512, device='cuda', requires_grad=True)
lens = torch.full((B,), T, device='cuda', dtype=torch.int32)
ps = [p.detach().clone().requires_grad_(True) for p in (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
      lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse)]
def fwd():
    return t2v_hip.BiLSTM.apply(x, lens, *ps, True)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
t_f = timeit(fwd)
wo = torch.randn(B, T, 512, device='cuda')
def fb():
    y = fwd(); (y * wo).sum().backward()
t_fb = timeit(fb)
print('BiLSTM B=%d T=%d: forward (incl. input GEMMs) %.0f us, forward+backward %.0f us' % (B, T, t_f, t_fb))
t2v_hip.check_async_errors()
