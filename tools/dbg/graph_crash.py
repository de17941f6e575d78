import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import hparams as HP, model as M, t2v_hip, train as TR
g = np.load(os.path.join(ROOT, 'tests/golden/train_step.npz'))
M.drop_rate = 0.0
modes = sys.argv[1].split(',')
pre_fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for mode in modes:
    hp = HP.create_hparams("anneal_function=constant,p_attention_dropout=0.0,p_decoder_dropout=0.0,bf16_run=%s" % (mode == 'bf16'))
    torch.manual_seed(hp.seed)
    eng = TR.TrainEngine(hp)
    eng.model.vae_gst.eps_override = torch.from_numpy(g['eps']).cuda()
    batch = (torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths']), torch.from_numpy(g['mel']),
             torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths']),
             torch.zeros(2, 1, dtype=torch.long), torch.from_numpy(g['emotions']))
    if pre_fwd:
        x, y = eng.model.parse_batch(batch)
        y_pred = eng.model(x)
    for i in range(5):
        print(mode, 'step', i, flush=True)
        l = eng.step(batch, i)
        torch.cuda.synchronize()
        print('  loss', float(l[0]), flush=True)
print('OK')
