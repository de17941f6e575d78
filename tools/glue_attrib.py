"""Attribute the non-HIP-library launches of one eager training step (ATen element-wise / reduce / copy / fill kernels)
to the Python source line that issued them, using the torch profiler's stacks.  Prints launches per step and device us
per step, grouped by (op, first frame inside this repo)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench, hparams as HP, train as TR

hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(hp.seed)
eng = TR.TrainEngine(hp, world_size=1, graph=False)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
NS = 2
with eng.stream_context():
    for it in range(4):
        eng.step(batch, it)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for it in range(4, 4 + NS):
            eng.step(batch, it)
        torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
tot = [0, 0.0]
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if e.cpu_children:          # only the leaf op that actually launched
        leaf = all(not c.kernels for c in e.cpu_children)
        if not leaf:
            continue
    frame = '?'
    p_ = e
    while p_ is not None and frame == '?':
        for fr in (p_.stack or []):
            if 'tacotron2-vae_amd/' in fr or 'bench.py' in fr:
                frame = fr.split('tacotron2-vae_amd/')[-1]
                break
        if frame == '?' and not (p_.stack or []) and p_.cpu_parent is None:
            frame = '? under ' + p_.name[:60]
        p_ = p_.cpu_parent
    for k in e.kernels:
        if k.name.startswith('k_') or k.name.startswith('void k_'):
            continue
        key = (e.name, frame[:90])
        agg[key][0] += 1
        agg[key][1] += k.duration
        tot[0] += 1
        tot[1] += k.duration
print("non-library launches: %.1f per step, %.1f us per step" % (tot[0] / NS, tot[1] / NS))
for (op, fr), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%5.1f/step %7.1f us  %-28s %s" % (c / NS, t / NS, op[:28], fr))
