"""where the per-utterance overhead of Decoder.inference goes: session set-up vs the persistent launch vs the read-back"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch
import hparams as HP, model as M, t2v_hip
torch.manual_seed(1234)
m = M.Tacotron2(HP.create_hparams("max_decoder_steps=800")).cuda().eval()
dec = m.decoder
dec.gate_threshold = 1.0
ids = torch.randint(2, 80, (1, 200), generator=torch.Generator().manual_seed(1234)).cuda()
z = torch.randn(1, 32, generator=torch.Generator().manual_seed(7)).cuda()
import contextlib, io
with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
    mem = m.encoder.inference(m.transcript_embedding(ids).transpose(1, 2)) + m.vae_gst.fc3(z).unsqueeze(1)
    for _ in range(2):
        dec.inference(mem); torch.cuda.synchronize()
    rows = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = dec._session(mem, None, 800)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        s.PRE[0].copy_(dec.prenet(dec.get_go_frame(mem)))
        s.run_persistent(1.0, 0.5, 1)
        t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
        stop = int(s.stop.item()); bad = s.persistent_timed_out()
        mel = s.MEL[:800].permute(1, 2, 0).contiguous(); torch.cuda.synchronize(); t5 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
for r in rows:
    print("session host %.0f us (+%.0f us GPU drain) | launch host %.0f us | kernel wait %.0f us | read-back %.0f us" % tuple(1e6 * x for x in r))
