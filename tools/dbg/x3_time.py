"""Time the large fp32 products of the step on the x3 kernel and on the fp32-MFMA kernel: `python tools/dbg/x3_time.py` (GPU)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tacotron2-vae_amd'))
import torch
import t2v_hip
g = torch.Generator().manual_seed(5)
shapes = [(4096, 2560, 2400, 'rr'), (4096, 1536, 2400, 'rr'), (4096, 1024, 2400, 'rr'), (4096, 512, 2400, 'rr'), (2400, 4096, 256, 'kk'),
          (2400, 256, 4096, 'kr'), (8192, 8192, 4096, 'rr'), (8192, 8192, 4096, 'kk')]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (M, N, K, form) in shapes:
    A = torch.randn(M, K, generator=g).cuda() if form[0] == 'k' else torch.randn(K, M, generator=g).cuda().t()
    B = torch.randn(N, K, generator=g).cuda() if form[1] == 'k' else torch.randn(K, N, generator=g).cuda().t()
    out = torch.empty(M, N, device='cuda')
    line = 'GEMM %5dx%5dx%5d %s:' % (M, N, K, form)
    for mode in (True, False):
        t2v_hip.set_f32_gemm_mode(mode)
        for _ in range(3):
            t2v_hip.gemm(A, B, out=out)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        n = 20
        for _ in range(n):
            t2v_hip.gemm(A, B, out=out)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / n
        line += '  %s %7.1f us = %6.1f TFLOP/s' % ('x3' if mode else 'f32-mfma', us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
