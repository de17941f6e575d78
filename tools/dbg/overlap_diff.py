"""eager engine: per-parameter difference of the gradient arena between the multi-stream and the single-stream schedule"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import hparams as HP, model as M, train as TR
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_step.npz'))
M.drop_rate = 0.0
res = {}
graph = len(sys.argv) > 1 and sys.argv[1] == 'graph'
for mode in (True, False, 'again'):
    M.Tacotron2.overlap_branches = bool(mode)
    hp = HP.create_hparams("anneal_function=constant,p_attention_dropout=0.0,p_decoder_dropout=0.0")
    torch.manual_seed(hp.seed)
    eng = TR.TrainEngine(hp, graph=graph)
    eng.model.vae_gst.eps_override = torch.from_numpy(g['eps']).cuda()
    batch = (torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths']), torch.from_numpy(g['mel']),
             torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths']),
             torch.zeros(2, 1, dtype=torch.long), torch.from_numpy(g['emotions']))
    nsteps = int(os.environ.get('NSTEPS', '1'))
    for it in range(nsteps):
        out = eng.step(batch, it)
    torch.cuda.synchronize()
    res[mode] = (float(out[0]), eng.optimizer.grads.clone(), eng)
for ka, kb in ((True, False), (True, 'again')):
    a, b = res[ka], res[kb]
    print(ka, 'vs', kb, 'loss', a[0], b[0])
    named, offs = a[2].optimizer.arena_layout()
    for (n, p), o in zip(named, offs):
        d = (a[1][o:o + p.numel()] - b[1][o:o + p.numel()]).abs().max().item()
        if d != 0:
            print('   %-60s maxdiff %.3e  (scale %.3e)' % (n, d, b[1][o:o + p.numel()].abs().max().item()))
print('done')
