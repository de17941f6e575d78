"""per-workgroup time line of ONE step (t = T/2) of the bf16 persistent decoder forward (csrc/decoder_train_persist16.hip) on
the chip-wide 100 MHz counter.  usage: python tools/dbg/persist16_prof.py B T_in T"""
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
mem = (torch.randn(B, T_in, 512, device='cuda') * 0.5)
mels = torch.randn(B, 80, T, device='cuda')
lens = torch.full((B,), T_in, device='cuda')
prof = torch.zeros(64 + 256 * 8, dtype=torch.int64, device='cuda')
H.DecoderCore.persistent16 = 'force'
H.DecoderCore.keep_last = True
with torch.no_grad():
    dec(mem, mels, lens)
    lib.t2v_set_phase_profile(C.c_void_p(prof.data_ptr()))
    dec(mem, mels, lens)
    torch.cuda.synchronize()
    lib.t2v_set_phase_profile(None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        H.replay_persistent_forward()
    e1.record()
    torch.cuda.synchronize()
print("kernel %s: %.1f us per launch = %.2f us per time step" % (H.DecoderCore.last_kernel, 1000 * e0.elapsed_time(e1) / 3, 1000 * e0.elapsed_time(e1) / 3 / T))
pv = prof.cpu().tolist()
rows = [pv[64 + 8 * w: 64 + 8 * w + 8] for w in range(256)]
Lw = [rows[w] for w in range(128, 256)]
Tw = [rows[w] for w in range(8 * B)]
t0 = min(r[0] for r in Lw)


def stat(rs, i):
    v = sorted((r[i] - t0) * 10 for r in rs)
    return "%6d %6d %6d" % (v[0], v[len(v) // 2], v[-1])


print("ns after the first L workgroup entered the step (min / median / max over the workgroups of a role)")
for i, name in enumerate(("L: step top (ctx(t-1) in hand)", "L: barrier passed (partial tiles in LDS)", "L: cell done, h_att / h_dec published",
                          "L: h_att(t) polled + 16 MFMAs", "L: h_dec(t-1) polled + 8 MFMAs", "L: ctx(t) polled")):
    print("  %-45s %s" % (name, stat(Lw, i)))
print("  L ctx poll rounds (median) %d, nap %d" % (sorted(r[6] for r in Lw)[64], sorted(r[7] for r in Lw)[64]))
for i, name in enumerate(("T: step top", "T: h_att(t) seen", "T: partial energies stored", "T: 8 partials gathered", "T: softmax done", "T: ctx published")):
    print("  %-45s %s" % (name, stat(Tw, i)))
print("  T h_att poll rounds (median) %d, nap %d" % (sorted(r[6] for r in Tw)[len(Tw) // 2], sorted(r[7] for r in Tw)[len(Tw) // 2]))
L = pv[0:6]
print("L workgroup 128 wave 0 (cycles): finish MFMAs + LDS %d, barrier %d, cell + publish %d, poll h_att + MFMA %d, poll h_dec + MFMA %d"
      % (L[1] - L[0], L[2] - L[1], L[3] - L[2], L[4] - L[3], L[5] - L[4]))
print("  inside the cell (cycles after the barrier): partial tiles summed %d, gates + cell done %d, publish stores issued %d"
      % (pv[13] - L[2], pv[14] - L[2], pv[15] - L[2]))
H.check_async_errors()
