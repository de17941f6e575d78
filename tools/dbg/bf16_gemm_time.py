"""Time the large bf16_run products (LSTM weight gradients at B = 16: K = T*B = 6400) on the plane kernel vs k_gemm_bf16_big_rr:
`T2V_BF16_GEMM_PLANES=1|0 python tools/dbg/bf16_gemm_time.py` (GPU)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tacotron2-vae_amd'))
import torch
import t2v_hip
t2v_hip.set_bf16(True)
g = torch.Generator().manual_seed(5)
for (M, N, K) in ((4096, 2560, 6400), (4096, 1536, 6400), (4096, 1024, 6400), (4096, 512, 6400), (6400, 4096, 256)):
    A = torch.randn(K, M, generator=g).cuda().t()
    B = torch.randn(K, N, generator=g).cuda().t()
    out = torch.empty(M, N, device='cuda')
    for _ in range(3):
        t2v_hip.gemm(A, B, out=out)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        t2v_hip.gemm(A, B, out=out)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / 20
    print('bf16 GEMM %5dx%5dx%5d: %7.1f us = %6.1f TFLOP/s (planes=%s)' % (M, N, K, us, 2.0 * M * N * K / us / 1e6, os.environ.get('T2V_BF16_GEMM_PLANES', '1')), flush=True)
