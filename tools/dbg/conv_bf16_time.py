"""Time the bf16_run k = 5 Conv1d forward on the plane kernels (conv_x3.hip, one plane) and on k_conv5_fwd_bf16k32: `python tools/dbg/conv_bf16_time.py` (GPU)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tacotron2-vae_amd'))
import torch
import torch.nn.functional as F
import t2v_hip
lib = t2v_hip.load_library()
g = torch.Generator().manual_seed(5)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for (B, Cin, Cout, T) in ((16, 512, 512, 400), (16, 512, 512, 84), (6, 512, 512, 400), (16, 128, 512, 129)):
    x = torch.randn(B, Cin, T, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 5, generator=g) / (Cin * 5) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    ref = F.conv1d(x.bfloat16().double().cpu(), w.bfloat16().double().cpu(), b.double().cpu(), padding=2)
    wp = torch.empty(w.numel(), device='cuda', dtype=torch.bfloat16)
    line = 'bf16 conv %2d x %3d->%3d x %3d:' % (B, Cin, Cout, T)
    for mode in (1, 0):
        lib.t2v_conv1d_x3_set_mode(mode)
        nblk = lib.t2v_conv1d_stat_blocks_bf16(B, T, Cin, Cout, 5)
        part = torch.zeros(nblk, Cout, 2, device='cuda')
        y = torch.full((B, Cout, T), float('nan'), device='cuda')
        for _ in range(3):
            rc = lib.t2v_conv1d_fwd_bf16(p(w), p(x), p(b), p(y), p(part), p(wp), B, Cin, T, Cout, 5, st)
        torch.cuda.synchronize()
        err = (y.cpu().double() - ref).abs().max().item()
        s_err = (part.cpu().double().sum(0)[:, 0] - ref.sum((0, 2))).abs().max().item() / ref.abs().sum((0, 2)).max().item()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            lib.t2v_conv1d_fwd_bf16(p(w), p(x), p(b), p(y), p(part), p(wp), B, Cin, T, Cout, 5, st)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / 20
        line += '  %s %7.1f us = %6.1f TFLOP/s (rc %d, max err %.1e, BN sum err %.1e)' % ('planes' if mode else 'k32', us, 2.0 * B * T * Cin * Cout * 5 / us / 1e6, rc, err, s_err)
    print(line, flush=True)
