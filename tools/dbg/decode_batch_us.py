"""us per frame of the persistent decode at B = 1..4 utterances of 150 symbols (800 frames, gate disabled)."""
import os, sys, time, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, hparams as HP, model as M, t2v_hip
steps, T_in = 800, int(sys.argv[1]) if len(sys.argv) > 1 else 150
hp = HP.create_hparams("max_decoder_steps=%d" % steps)
torch.manual_seed(hp.seed); M.drop_rate = 0.0
m = M.Tacotron2(hp).cuda().eval()
m.decoder.gate_threshold = 1.0
for B in (1, 2, 3, 4):
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(2, 80, (B, T_in), generator=g).cuda()
    z = torch.randn(B, 32, generator=torch.Generator().manual_seed(7)).cuda()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        mem = m.encoder.inference(m.transcript_embedding(ids).transpose(1, 2)) + m.vae_gst.fc3(z).unsqueeze(1)
        res = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m.decoder.inference(mem, persistent=True); torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e6)
    print('B=%d T_in=%d: %.2f us per frame (%.2f per utterance-frame), frames %d' % (B, T_in, sorted(res)[1], sorted(res)[1] / B, out[0].shape[2]), flush=True)
t2v_hip.check_async_errors()
