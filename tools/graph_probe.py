"""Where does a graph-mode step lose time?  Times (a) bare replays, (b) engine.step incl. its copies, (c) host time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, hparams as HP, train as TR
from bench import synthetic_batch
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(1234)
eng = TR.TrainEngine(hp, graph=True)
batch = tuple(t.pin_memory() for t in synthetic_batch(6, 84, 400, 1234))
for it in range(6): eng.step(batch, it)
torch.cuda.synchronize()
(graph, static_in, static_out), = eng._graphs.values()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3
cur = torch.cuda.current_stream()
srcs = [t.clone() for t in static_in]
def bare(i):
    with torch.cuda.stream(eng._stream): graph.replay()
def with_waits(i):
    eng._stream.wait_stream(cur)
    with torch.cuda.stream(eng._stream): graph.replay()
    cur.wait_stream(eng._stream)
def with_d2d(i):
    with torch.cuda.stream(eng._stream):
        for d, s_ in zip(static_in, srcs): d.copy_(s_, non_blocking=True)
        graph.replay()
def with_h2d(i):
    with torch.cuda.stream(eng._stream):
        xx, yy = eng.model.parse_batch(batch)
        graph.replay()
def step_in_stream(i):
    with torch.cuda.stream(eng._stream): eng.step(batch, 10 + i)
def step_default(i):
    eng.step(batch, 10 + i)
eng2 = TR.TrainEngine(hp, graph=False)
for it in range(3): eng2.step(batch, it)
def eager(i):
    eng2.step(batch, 10 + i)
for rnd in range(3):
    for name, fn in (('bare replay', bare), ('replay + stream waits', with_waits), ('replay + D2D', with_d2d),
                     ('replay + H2D', with_h2d), ('engine.step in stream', step_in_stream), ('engine.step default stream', step_default),
                     ('eager engine.step', eager)):
        print('%d %-28s host %.3f ms, wall %.3f ms/step' % ((rnd, name) + timeit(fn)))
