import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd'))
import torch, hparams as HP, model as M
hp = HP.create_hparams("max_decoder_steps=8")
torch.manual_seed(hp.seed); M.drop_rate = 0.0
m = M.Tacotron2(hp).cuda().eval()
dec = m.decoder
mem = torch.randn(1, 30, 512, device='cuda') * 0.5
with torch.no_grad():
    dec.initialize_decoder_states(mem, mask=None); torch.cuda.synchronize(); print('session ok', flush=True)
    s = dec._sess
    s.PRE[0].copy_(dec.prenet(dec.get_go_frame(mem))); torch.cuda.synchronize(); print('prenet ok', flush=True)
    for t in range(3):
        s.run(t, t + 1, 0.5, 0.0, False, 1); torch.cuda.synchronize(); print('frame', t, 'ok', s.MEL[t, 0, :3].tolist(), flush=True)
