import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import hparams as HP, model as M, t2v_hip, train as TR
g = np.load(os.path.join(ROOT, 'tests/golden/train_step.npz'))
M.drop_rate = 0.0
variant = sys.argv[1]
hp = HP.create_hparams("anneal_function=constant,p_attention_dropout=0.0,p_decoder_dropout=0.0")
torch.manual_seed(hp.seed)
eng = TR.TrainEngine(hp)
eng.model.vae_gst.eps_override = torch.from_numpy(g['eps']).cuda()
batch = (torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths']), torch.from_numpy(g['mel']),
         torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths']),
         torch.zeros(2, 1, dtype=torch.long), torch.from_numpy(g['emotions']))
if variant == 'del':
    x, y = eng.model.parse_batch(batch); y_pred = eng.model(x); del x, y, y_pred
elif variant == 'nograd':
    with torch.no_grad():
        x, y = eng.model.parse_batch(batch); y_pred = eng.model(x)
elif variant == 'onstream':
    with eng.stream_context():
        x, y = eng.model.parse_batch(batch); y_pred = eng.model(x)
elif variant == 'parse_only':
    x, y = eng.model.parse_batch(batch)
elif variant == 'keep':
    x, y = eng.model.parse_batch(batch); y_pred = eng.model(x)
elif variant == 'other_model':
    m2 = M.Tacotron2(hp).cuda().train()
    m2.vae_gst.eps_override = eng.model.vae_gst.eps_override
    x, y = m2.parse_batch(batch); y_pred = m2(x); del x, y, y_pred, m2
elif variant.startswith('part_'):
    m = eng.model
    x, y = m.parse_batch(batch)
    text, lens_in, mel, maxlen, lens_out, spk, emo = x
    if variant == 'part_embed':
        o = m.transcript_embedding(text)
    elif variant == 'part_encoder':
        with torch.no_grad():
            e = m.transcript_embedding(text).transpose(1, 2)
        o = m.encoder(e, lens_in)
    elif variant == 'part_vae':
        o = m.vae_gst(mel)
    elif variant == 'part_postnet':
        o = m.postnet(mel)
    elif variant == 'part_decoder':
        mem = torch.randn(2, text.shape[1], 512, device='cuda')
        o = m.decoder(mem, mel, lens_in)
    elif variant == 'part_prenet':
        o = m.decoder.prenet(torch.randn(5, 2, 80, device='cuda'))
    del o, x, y
os.environ['T2V_TRAIN_PERSISTENT'] = '0'
for i in range(4):
    l = eng.step(batch, i)
    torch.cuda.synchronize()
    print(variant, 'step', i, 'loss', float(l[0]), flush=True)
print(variant, 'OK')
