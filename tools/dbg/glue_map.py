"""Which Python line issues which PyTorch (aten) device op in one eager training step: a TorchDispatchMode logs every aten
call on device tensors with the innermost frames of this repo (ops issued by autograd's own nodes show no frame)."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten
import hparams as HP, train as TR
from bench import synthetic_batch

SKIP = ('aten.view', 'aten.transpose', 'aten.t.', 'aten.slice', 'aten.select', 'aten.unsqueeze', 'aten.squeeze', 'aten.expand',
        'aten.permute', 'aten.detach', 'aten.alias', 'aten._unsafe_view', 'aten.as_strided', 'aten.empty', 'aten.split',
        'aten.unbind', 'aten.reshape', 'aten.narrow', 'aten.chunk', 'aten.is_pinned', 'aten._local_scalar', 'aten.record_stream',
        'aten.new_empty', 'aten.zeros_like.default_', 'aten.lift_fresh', 'aten.result_type', 'aten.is_same_size', 'aten.sym_',
        'aten.unflatten', 'aten.flatten', 'aten.movedim', 'aten.contiguous')
log = []


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            flat = [a for a in tree_flatten((args, kwargs or {}))[0] if torch.is_tensor(a)]
            if any(a.is_cuda for a in flat):
                fr = [f for f in traceback.extract_stack() if ('tacotron2-vae_amd' in f.filename or f.filename.endswith('bench.py'))]
                where = ' <- '.join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[-3:]))
                shapes = ' '.join(str(tuple(a.shape)) for a in flat[:3])
                log.append("%-34s %-44s %s" % (name, shapes[:44], where))
        return out


hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(1234)
eng = TR.TrainEngine(hp, graph=False)
batch = tuple(t.pin_memory() for t in synthetic_batch(6, 84, 400, 1234))
with eng.stream_context():
    for it in range(4):
        eng.step(batch, it)
    torch.cuda.synchronize()
    with Log():
        eng.step(batch, 4)
        torch.cuda.synchronize()
print('\n'.join(log))
print(len(log), "aten device ops")
