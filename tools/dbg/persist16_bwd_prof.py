"""one turn of the attention chain of the bf16 persistent reverse pass (csrc/decoder_train_bwd_persist16.hip) through its roles,
on the chip-wide 100 MHz counter.  usage: python tools/dbg/persist16_bwd_prof.py B T_in T"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import ctypes as C
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in sys.argv[1:4])
lib = H.load_library()
H.set_bf16(True)
hp = HP.create_hparams("bf16_run=True"); torch.manual_seed(0)
dec = M.Decoder(hp).cuda().train()
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5).requires_grad_(True)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
prof = torch.zeros(64 + 256 * 8, dtype=torch.int64, device='cuda')
H.DecoderCore.persistent16 = 'force'
H.DecoderCore.keep_last = True


def run():
    mel, gate, al = dec(mem, mels, lens)
    (mel.sum() + 0.3 * gate.sum()).backward()


run()
torch.cuda.synchronize()
mel, gate, al = dec(mem, mels, lens)
lib.t2v_set_phase_profile(C.c_void_p(prof.data_ptr()))
(mel.sum() + 0.3 * gate.sum()).backward()
torch.cuda.synchronize()
lib.t2v_set_phase_profile(None)
assert H.DecoderCore.last_bwd_kernel == 'k_bwd_persist16', H.DecoderCore.last_bwd_kernel
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    H.replay_persistent_backward()
e1.record()
torch.cuda.synchronize()
print("kernel k_bwd_persist16: %.1f us per launch = %.2f us per reverse step" % (1000 * e0.elapsed_time(e1) / 3, 1000 * e0.elapsed_time(e1) / 3 / T))
pv = prof.cpu().tolist()
rows = [pv[64 + 8 * w: 64 + 8 * w + 8] for w in range(256)]
S = lib.t2v_decoder_bwd_persist16_slices(T_in)
NT = B * S
Tw, GA, GD, CA, CD = rows[:NT], rows[96:144], rows[144:224], rows[224:240], rows[240:256]
t0 = sorted(r[7] for r in CA)[len(CA) // 2]          # median: attention_rnn cells publish dga(T/2 + 1)


def stat(rs, i):
    v = sorted((r[i] - t0) * 10 for r in rs)
    return "%7d %7d %7d" % (v[0], v[len(v) // 2], v[-1])


print("ns relative to 'attention_rnn cells published dga(T/2+1)' (median); min / median / max over the workgroups of a role")
for name, rs, i in (("C_a  step T/2+1: dga published", CA, 7),
                    ("G_a  iter T/2+1: loop top", GA, 4), ("G_a  iter T/2+1: row quarter polled", GA, 5), ("G_a  iter T/2+1: partial sums published", GA, 6),
                    ("T    step T/2  : loop top", Tw, 0), ("T    step T/2  : d ctx complete", Tw, 1), ("T    step T/2  : dq published", Tw, 2), ("T    step T/2  : step end", Tw, 3),
                    ("C_a  step T/2  : loop top", CA, 0), ("C_a  step T/2  : E_h + ya_h partials in", CA, 1), ("C_a  step T/2  : dq in", CA, 2),
                    ("C_a  step T/2  : dga published", CA, 3)):
    print("  %-45s %s" % (name, stat(rs, i)))
td = sorted(r[7] for r in CD)[len(CD) // 2]
print("decoder_rnn chain (free-running), ns relative to 'decoder_rnn cells published dgd(T/2+1)'")
for name, rs, i in (("C_d  step T/2+1: dgd published", CD, 7), ("G_d  iter T/2+1: row quarter polled", GD, 5), ("G_d  iter T/2+1: partial sums published", GD, 6),
                    ("C_d  step T/2  : loop top", CD, 0), ("C_d  step T/2  : partials in", CD, 1), ("C_d  step T/2  : dgd published", CD, 3)):
    v = sorted((r[i] - td) * 10 for r in rs)
    print("  %-45s %7d %7d %7d" % (name, v[0], v[len(v) // 2], v[-1]))
print("decoder_rnn chain is %.1f us ahead of the attention chain at step T/2" % ((t0 - td) / 100.0))
H.check_async_errors()
