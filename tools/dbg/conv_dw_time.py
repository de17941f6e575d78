"""isolated timing of the k = 5 Conv1d weight gradient: fp32 kernel vs bf16 kernel (events around back-to-back launches)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd'))
import ctypes as C
import torch, t2v_hip as H
lib = H.load_library()
for B, Cin, Cout, T in ((16, 512, 512, 400), (6, 512, 512, 400), (16, 512, 80, 400), (16, 80, 512, 400), (16, 512, 512, 84)):
    x = torch.randn(B, Cin, T, device='cuda'); dy = torch.randn(B, Cout, T, device='cuda')
    dw = torch.empty(Cout, Cin, 5, device='cuda')
    nscr = lib.t2v_conv1d_dw_scratch_floats(B, Cin, T, Cout, 5)
    scr = torch.empty(max(nscr, 1), device='cuda')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = []
    for fn in (lib.t2v_conv1d_bwd, lib.t2v_conv1d_bwd_bf16):
        for _ in range(3):
            fn(None, C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), None, C.c_void_p(dw.data_ptr()), None, C.c_void_p(scr.data_ptr()), B, Cin, T, Cout, 5, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(None, C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), None, C.c_void_p(dw.data_ptr()), None, C.c_void_p(scr.data_ptr()), B, Cin, T, Cout, 5, st)
        e1.record(); torch.cuda.synchronize()
        res.append(100 * e0.elapsed_time(e1))
    fl = 2.0 * B * T * Cout * Cin * 5
    print("B=%d %d->%d T=%d: fp32 %.1f us (%.0f TF), bf16 %.1f us (%.0f TF), splits %s" % (B, Cin, Cout, T, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, nscr))
