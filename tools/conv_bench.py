"""Time the Conv1d kernels (forward, data gradient, weight gradient) on the shapes of one train step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd'))
import torch, t2v_hip
from t2v_hip import _p, _stream, _check
lib = t2v_hip.load_library()
import ctypes as C
lib.t2v_set_phase_profile.argtypes = [C.c_void_p]
prof = torch.zeros(32, dtype=torch.int64, device='cuda')
shapes = [('postnet 80->512', 6, 80, 512, 400), ('postnet 512->512', 6, 512, 512, 400), ('postnet 512->80', 6, 512, 80, 400),
          ('encoder 512->512', 6, 512, 512, 84), ('postnet 512->512 B16', 16, 512, 512, 400)]
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, B, Cin, Cout, T in shapes:
    x = torch.randn(B, Cin, T, device='cuda'); w = torch.randn(Cout, Cin, 5, device='cuda') * 0.02
    y = torch.empty(B, Cout, T, device='cuda'); dy = torch.randn(B, Cout, T, device='cuda')
    dx = torch.empty_like(x); dw = torch.empty_like(w); wt = torch.empty_like(w)
    nscr = lib.t2v_conv1d_dw_scratch_floats(B, Cin, T, Cout, 5)
    scr = torch.empty(max(nscr, 1), device='cuda')
    nblk = lib.t2v_conv1d_stat_blocks(B, T, Cin, Cout, 5)
    part = torch.empty(nblk, Cout, 2, device='cuda')
    fl = 2.0 * B * T * Cin * Cout * 5
    lib.t2v_set_phase_profile(C.c_void_p(prof.data_ptr()))
    t_f = timeit(lambda: _check(lib.t2v_conv1d_fwd(_p(w), _p(x), None, _p(y), _p(part), B, Cin, T, Cout, 5, _stream()), 'fwd'))
    torch.cuda.synchronize(); pv = prof.cpu().tolist(); lib.t2v_set_phase_profile(None)
    cyc, real = pv[1] - pv[0], (pv[3] - pv[2]) / 100.0
    print('   fwd main loop of wg(0,0): %d shader cycles in %.1f us -> %.2f GHz; %.0f cycles per k-tile' % (cyc, real, cyc / real / 1e3, cyc / (Cin / 16)))
    t_x = timeit(lambda: _check(lib.t2v_conv1d_bwd(_p(w), _p(x), _p(dy), _p(dx), None, _p(wt), None, B, Cin, T, Cout, 5, _stream()), 'dx'))
    t_w = timeit(lambda: _check(lib.t2v_conv1d_bwd(_p(w), _p(x), _p(dy), None, _p(dw), None, _p(scr), B, Cin, T, Cout, 5, _stream()), 'dw'))
    print('%-22s %6.2f GFLOP | fwd %7.1f us %5.1f TF | dx %7.1f us %5.1f TF | dw %7.1f us %5.1f TF' %
          (name, fl / 1e9, t_f, fl / t_f / 1e6, t_x, fl / t_x / 1e6, t_w, fl / t_w / 1e6))
