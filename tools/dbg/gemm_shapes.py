"""Every t2v_hip.gemm call of one eager fp32 training step with its shape and its time alone (events around the call, synchronised):
which of the ~50 small products of a step are worth a different kernel.  `python tools/dbg/gemm_shapes.py [--bf16]` (GPU)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch
import hparams as HP, train as TR, t2v_hip
from bench import synthetic_batch
bf16 = '--bf16' in sys.argv
B = 16 if bf16 else 6
hp = HP.create_hparams("batch_size=%d,anneal_function=constant%s" % (B, ",bf16_run=True" if bf16 else ""))
torch.manual_seed(1234)
eng = TR.TrainEngine(hp, graph=False)
batch = tuple(t.pin_memory() for t in synthetic_batch(B, 84, 400, 1234))
for it in range(3):
    eng.step(batch, it)
torch.cuda.synchronize()
rows = []
orig = t2v_hip.gemm
def timed(A, Bm, bias=None, out=None, **kw):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(A, Bm, bias, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    rows.append((e0.elapsed_time(e1) * 1e3, A.shape[0], Bm.shape[0], A.shape[1], A.stride(1) == 1, Bm.stride(1) == 1, kw.get('accumulate', False)))
    return r
t2v_hip.gemm = timed
eng.step(batch, 3)
torch.cuda.synchronize()
t2v_hip.gemm = orig
if '--order' in sys.argv:
    for i, r in enumerate(rows):
        print('#%02d %7.1f us  M=%5d N=%5d K=%5d  A_kc=%d B_kc=%d acc=%d' % (i, r[0], r[1], r[2], r[3], r[4], r[5], r[6]))
tot = sum(r[0] for r in rows)
print('%d gemm calls, %.0f us alone in total' % (len(rows), tot))
for r in sorted(rows, key=lambda r: -r[0])[:40]:
    M, N, K = r[1], r[2], r[3]
    print('%7.1f us  M=%5d N=%5d K=%5d  A_kc=%d B_kc=%d acc=%d  %6.1f GFLOP/s-ish %5.1f TF' % (r[0], M, N, K, r[4], r[5], r[6], 2.0 * M * N * K / r[0] / 1e3, 2.0 * M * N * K / r[0] / 1e6))
