"""Time the k = 5 Conv1d forward of the step's shapes on the x3 path and on the fp32-MFMA kernels: `python tools/dbg/conv_x3_time.py` (GPU)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tacotron2-vae_amd'))
import torch
import t2v_hip
lib = t2v_hip.load_library()
lib.t2v_conv1d_x3_set_mode(1)
g = torch.Generator().manual_seed(5)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for (B, Cin, Cout, T) in ((6, 512, 512, 400), (6, 80, 512, 400), (6, 512, 80, 400), (6, 512, 512, 84), (16, 512, 512, 400)):
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 5, generator=g) / (Cin * 5) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    y = torch.empty(B, Cout, T, device='cuda')
    line = 'conv %2d x %3d->%3d x %3d:' % (B, Cin, Cout, T)
    for mode in (1, 0):
        lib.t2v_gemm_f32_set_mode(mode)
        nblk = lib.t2v_conv1d_stat_blocks(B, T, Cin, Cout, 5)
        part = torch.empty(nblk, Cout, 2, device='cuda')
        for _ in range(3):
            lib.t2v_conv1d_fwd(p(w), p(x), p(b), p(y), p(part), B, Cin, T, Cout, 5, st)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        n = 20
        for _ in range(n):
            lib.t2v_conv1d_fwd(p(w), p(x), p(b), p(y), p(part), B, Cin, T, Cout, 5, st)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / n
        line += '  %s %7.1f us = %6.1f TFLOP/s' % ('x3' if mode else 'f32-mfma', us, 2.0 * B * T * Cin * Cout * 5 / us / 1e6)
    print(line, flush=True)
