"""Isolated timing of the time-batched GEMM shapes of a training step (`python tools/dbg/gemm_time.py [--bf16] [B]`):
the four deferred LSTM weight gradients, the Prenet data gradient, the hoisted attention_rnn input term."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tacotron2-vae_amd'))
import torch
import t2v_hip

bf16 = '--bf16' in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(args[0]) if args else (16 if bf16 else 6)
T = 400
TB = T * B
t2v_hip.set_bf16(bf16)
dev = 'cuda'
dg = torch.randn(TB, 4096, device=dev)
x = torch.randn(TB, 2560, device=dev)
w = torch.randn(4096, 768, device=dev)
pre = torch.randn(TB, 256, device=dev)
shapes = [
    ('dW_hh_att  4096x1024 K=TB', lambda: t2v_hip.gemm(dg.t(), x[:, :1024].t())),
    ('dW_ih_att  4096x512  K=TB', lambda: t2v_hip.gemm(dg.t(), x[:, 1024:1536].t())),
    ('dW_ih_dec  4096x1536 K=TB', lambda: t2v_hip.gemm(dg.t(), x[:, :1536].t())),
    ('dW_hh_dec  4096x1024 K=TB', lambda: t2v_hip.gemm(dg.t(), x[:, 1536:].t())),
    ('dW_pre     4096x256  K=TB', lambda: t2v_hip.gemm(dg.t(), pre.t())),
    ('d_pre      TBx256 K=4096 ', lambda: t2v_hip.gemm(dg, w[:, :256].t())),
    ('gpre       TBx4096 K=256 ', lambda: t2v_hip.gemm(pre, w[:, :256])),
]
flops = [2 * 4096 * 1024 * TB, 2 * 4096 * 512 * TB, 2 * 4096 * 1536 * TB, 2 * 4096 * 1024 * TB, 2 * 4096 * 256 * TB,
         2 * TB * 256 * 4096, 2 * TB * 4096 * 256]
for (name, fn), fl in zip(shapes, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%-28s %8.1f us  %6.1f TFLOP/s" % (name, us, fl / us / 1e6))
