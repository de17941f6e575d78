"""Where inside a slow eager step does the GPU time go?  Events around forward / loss / backward / optimiser."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
eng = TR.TrainEngine(hp, world_size=1, graph=False)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
E = lambda: torch.cuda.Event(enable_timing=True)
rows = []
m, crit, opt = eng.model, eng.criterion, eng.optimizer
dec = m.decoder
with eng.stream_context():
    for it in range(16):
        eng._publish(it)
        x, y = m.parse_batch(batch)
        e = [E() for _ in range(6)]
        e[0].record(); opt.zero_grad()
        yp = m(x); e[1].record()
        loss, recon, kl, w = crit(yp, y, it); e[2].record()
        loss.backward(); e[3].record()
        opt.gather_grads(); gn = opt.step(); e[4].record()
        float(loss.item())
        rows.append([e[i].elapsed_time(e[i + 1]) for i in range(4)])
for it, r in enumerate(rows):
    if it >= 4:
        print('step %2d: forward %.1f  loss %.1f  backward %.1f  optimiser %.1f ms' % (it, *r))
