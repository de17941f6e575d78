import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd'))
import contextlib, io
import torch, hparams as HP, model as M, t2v_hip
steps, T_in = int(sys.argv[1]), int(sys.argv[2])
hp = HP.create_hparams("max_decoder_steps=%d" % steps)
torch.manual_seed(hp.seed); M.drop_rate = 0.0
m = M.Tacotron2(hp).cuda().eval()
m.decoder.gate_threshold = 1.0
g = torch.Generator().manual_seed(1234)
ids = torch.randint(2, 80, (1, T_in), generator=g).cuda()
z = torch.randn(1, 32, generator=torch.Generator().manual_seed(7)).cuda()
with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
    mem = m.encoder.inference(m.transcript_embedding(ids).transpose(1, 2)) + m.vae_gst.fc3(z).unsqueeze(1)
    a = m.decoder.inference(mem, chunk=steps, persistent=False)
    torch.cuda.synchronize()
    b = m.decoder.inference(mem, persistent=True)
    torch.cuda.synchronize()
    res = {}
    for name, flag in (('launch-per-stage', False), ('persistent', True)):
        m.decoder.inference(mem, chunk=steps, persistent=flag); torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.decoder.inference(mem, chunk=steps, persistent=flag); torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / steps * 1e6
print('frames', a[0].shape[2], b[0].shape[2])
print('mel max diff %.3e  gate %.3e  align %.3e  argmax path equal %s' % ((a[0] - b[0]).abs().max().item(), (a[1] - b[1]).abs().max().item(),
      (a[2] - b[2]).abs().max().item(), bool(torch.equal(a[2].argmax(-1), b[2].argmax(-1)))))
print('us/frame:', res)
t2v_hip.check_async_errors()
import ctypes as C
lib = t2v_hip.load_library()
buf = torch.zeros(64 + 256 * 16, dtype=torch.int64, device='cuda')
lib.t2v_set_phase_profile(C.c_void_p(buf.data_ptr()))
with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
    m.decoder.inference(mem, persistent=True); torch.cuda.synchronize()
lib.t2v_set_phase_profile(None)
v = buf.cpu().tolist()
if steps > 101:
    print('wg0  (attention role) frame 100 cycles: att_rnn %d | h_att hop %d | q+energies %d | exchange+softmax %d | context %d | ctx hop %d | dec_rnn %d | tail %d  = %d'
          % (v[1]-v[0], v[2]-v[1], v[3]-v[2], v[4]-v[3], v[5]-v[4], v[6]-v[5], v[7]-v[6], v[8]-v[7], v[8]-v[0]))
    print('wg64 (projection) frame 100: to stage 3 %d | stage 3 (ctx hop + dec_rnn) %d | h_dec hop + projection %d' % (v[11]-v[10], v[12]-v[11], v[13]-v[12]))
    print('wg128 (Prenet 1) frame 100: to stage 5 %d | pre0 hop + layer 1 %d' % (v[15]-v[14], v[16]-v[15]))
    if steps > 601 and v[23] > v[21]:
        print('500 frames = %d shader cycles = %.1f us on the 100 MHz wall clock  =>  %.2f GHz, %.2f us per frame' % (
            v[22] - v[20], (v[23] - v[21]) / 100.0, (v[22] - v[20]) / ((v[23] - v[21]) * 10.0), (v[23] - v[21]) / 100.0 / 500))

if steps > 101:
    # per-workgroup time line of frame 100 (ns after the first workgroup entered the frame)
    rt = lambda w, k: v[64 + w * 16 + k] * 10
    t0 = min(rt(w, 0) for w in range(256))
    names = ['frame entry done (pre1 + stop in)', 'h_att published', 'h_att gathered', 'energies published (attention wgs)', 'softmax done (attention wgs)',
             'ctx published / stage 3 entered', 'ctx (+ h_dec(t-1)) gathered', 'h_dec published', 'projection done (projection wgs) / stage 5 entered', 'frame end']
    groups = {'attention wgs 0..7': range(0, 8), 'projection wgs 64..106': range(64, 107), 'Prenet-1 wgs 128..159': range(128, 160), 'plain wgs 160..255': range(160, 256)}
    for k, nm in enumerate(names):
        line = '  %-52s' % nm
        for gname, ws in groups.items():
            xs = sorted(rt(w, k) - t0 for w in ws if v[64 + w * 16 + k])
            if xs:
                line += ' | %s: %5d..%5d' % (gname.split()[0][:5], xs[0], xs[-1])
        print(line)
