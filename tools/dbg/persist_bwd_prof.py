"""phase stamps of the persistent reverse pass (step T/2): L workgroup slots 0..6, T workgroup 0 slots 8..11"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import ctypes as C
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in sys.argv[1:4])
lib = H.load_library()
hp = HP.create_hparams(); torch.manual_seed(0)
M.drop_rate = 0.0
dec = M.Decoder(hp).cuda().train()
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5).requires_grad_(True)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
prof = torch.zeros(4096, dtype=torch.int64, device='cuda')
H.DecoderCore.keep_last = True
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(3):
    mel, gate, al = dec(mem, mels, lens)
    loss = mel.sum() + gate.sum()
    if it == 2:
        lib.t2v_set_phase_profile(C.c_void_p(prof.data_ptr()))
    loss.backward()
    torch.cuda.synchronize()
lib.t2v_set_phase_profile(None)
for it in range(3):
    ev[0].record(); H.replay_persistent_backward(); ev[1].record(); torch.cuda.synchronize()
    tb = ev[0].elapsed_time(ev[1]) * 1e3 / T
    ev[0].record(); H.replay_persistent_forward(); ev[1].record(); torch.cuda.synchronize()
    tf = ev[0].elapsed_time(ev[1]) * 1e3 / T
    print('replay: %.2f us per reverse step; persistent forward %.2f us per step (7.70 at nominal clocks) -> %.2f us clock-normalised'
          % (tb, tf, tb * 7.70 / tf))
raw = prof.cpu().tolist()
pv = [x / T for x in raw[:64]]
print('A role (cycles per step, mean over the pass): loop top %d | gather d(t+1) %d | ctx-column GEMV + sync %d | publish dctx %d | recurrent GEMV + sums %d | '
      'dq gather + sync %d | Wq^T dq + cell + publish %d | park factors + next cell pre-part %d || step %d'
      % (pv[0], pv[1], pv[2], pv[3], pv[5], pv[4], pv[6], pv[7], sum(pv[0:8])))
print('   inside: park %d, cell pre-part %d' % (pv[12], pv[13]))
print('T role (cycles per step): prefetch + wait dctx %d | softmax/tanh backward -> dq published %d | location backward + window partials %d | loop top %d || step %d'
      % (pv[9], pv[10], pv[11], pv[8], sum(pv[8:12])))
print('decoder_rnn role (free-running): %.2f us per step; attention_rnn role: %.2f us per step' % ((raw[41] - raw[40]) * 0.01 / T, (raw[43] - raw[42]) * 0.01 / T))
# per-workgroup time line of step T/2 (100 MHz chip-wide counter -> ns), relative to the first 'dq published' of that step
S = lib.t2v_attn_bwd_slices(T_in); NT = B * S
def col(w0, w1, slot):
    return [raw[64 + w * 8 + slot] * 10 for w in range(w0, w1) if raw[64 + w * 8 + slot]]
NL = 256 - NT; NA = max((3 * NL + 4) // 8, 79)
t0 = min(col(0, NT, 1))
def show(name, v):
    v = sorted(x - t0 for x in v)
    print('  %-44s first %6d  median %6d  last %6d ns' % (name, v[0], v[len(v) // 2], v[-1]))
show('T: dq(t) published', col(0, NT, 1))
show('A: dq gathered (P4 done)', col(NT, NT + NA, 4))
show('A: d(t) published', col(NT, NT + NA, 5))
print('  -- next step (t-1) happens one step period later; same slots measured at step T/2 only, so use differences:')
show('A: loop top', col(NT, NT + NA, 0))
show('A: d(t+1) gathered', col(NT, NT + NA, 1))
show('A: dctx(t) published', col(NT, NT + NA, 2))
show('A: recurrent GEMV done', col(NT, NT + NA, 3))
show('T: dctx(t) detected', col(0, NT, 0))
H.check_async_errors()
