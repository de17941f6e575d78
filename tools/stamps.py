"""Phase time line of an UNDISTURBED training step (no tracer): T2V_STAMPS=1 makes the engine drop one-thread launches
that write the chip-wide 100 MHz clock at named points of the step (t2v_hip.stamp); this script runs the bench
configuration (B=6, T_in=84, T_out=400, graph replay unless --no-graph) and prints the stamps of the last step, averaged
over a few steps, relative to 'step_begin'.   usage: T2V_STAMPS=1 python tools/stamps.py [--no-graph] [--bf16]"""
import os, sys, time
os.environ['T2V_STAMPS'] = '0' if '--no-stamps' in sys.argv else '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch
import hparams as HP, train as TR, t2v_hip
from bench import synthetic_batch
graph = '--no-graph' not in sys.argv
bf16 = '--bf16' in sys.argv
B = 16 if bf16 else 6
hp = HP.create_hparams("batch_size=%d,anneal_function=constant%s" % (B, ",bf16_run=True" if bf16 else ""))
torch.manual_seed(1234)
eng = TR.TrainEngine(hp, graph=graph)
batch = tuple(t.pin_memory() for t in synthetic_batch(B, 84, 400, 1234))
acc, n = {}, 0
with eng.stream_context():
    for it in range(12):
        eng.step(batch, it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(20):
        eng.step(batch, 12 + it)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    for it in range(8):
        eng.step(batch, 40 + it)
        st = t2v_hip.read_stamps()
        b = st.get('step_begin', 0.0)
        for k, v in st.items():
            acc[k] = acc.get(k, 0.0) + (v - b)
        n += 1
print("%s step %.3f ms (20 back-to-back steps); stamps, us after step_begin (mean of %d synchronised steps):" % (
    'graph' if graph else 'eager', ms, n))
for k, v in sorted(acc.items(), key=lambda kv: kv[1]):
    print("  %-18s %9.1f" % (k, v / n))
