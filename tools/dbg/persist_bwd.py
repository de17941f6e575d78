"""Persistent backward pieces vs the launch-per-step backward.  python tools/dbg/persist_bwd.py B T_in T [p_drop] [r]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import ctypes as C
import torch, t2v_hip as H, hparams as HP, model as M
B, T_in, T = (int(x) for x in sys.argv[1:4])
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
ragged = len(sys.argv) > 5
lib = H.load_library()
hp = HP.create_hparams(); torch.manual_seed(0)
M.drop_rate = 0.0
dec = M.Decoder(hp).cuda().train()
dec.p_attention_dropout = dec.p_decoder_dropout = p
g = torch.Generator().manual_seed(1)
mem = (torch.randn(B, T_in, 512, generator=g) * 0.5).cuda().requires_grad_(True)
mels = torch.randn(B, 80, T, generator=g).cuda()
lens = torch.tensor([max(1, T_in - 7 * i) for i in range(B)] if ragged else [T_in] * B).cuda()
H.DecoderCore.keep_last = True
os.environ['T2V_BWD_PERSISTENT'] = '0'
mel, gate, al = dec(mem, mels, lens)
(mel.sum() + 0.3 * gate.sum() + 0.01 * (mel * mel).sum()).backward()
torch.cuda.synchronize(); H.check_async_errors()
W, Sb, Gb, dims, keep = H.DecoderCore.last_bwd
names = ('gpre', 'memory', 'pm', 'lengths', 'XS', 'CA', 'CD', 'GA', 'GD', 'QP', 'AL', 'ACUM', 'S', 'dHC', 'DGA', 'DGD', 'DQ', 'DCTX')
k = dict(zip(names, keep))
_, _, _, p_att, p_dec, seed = dims
f32 = dict(device='cuda', dtype=torch.float32)
DGD2 = torch.full_like(k['DGD'], float('nan'))
scr = torch.empty(lib.t2v_decoder_bwd_dchain_scratch_floats(B, T), **f32)
err = torch.zeros(1, dtype=torch.int32, device='cuda')
w_hh_dec = dec.decoder_rnn.weight_hh.detach().contiguous()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(3):
    ev[0].record()
    rc = lib.t2v_decoder_bwd_dchain(C.c_void_p(w_hh_dec.data_ptr()), C.c_void_p(k['dHC'].data_ptr()), C.c_void_p(k['GD'].data_ptr()),
                                    C.c_void_p(k['CD'].data_ptr()), C.c_void_p(DGD2.data_ptr()), C.c_void_p(scr.data_ptr()),
                                    C.c_void_p(err.data_ptr()), B, T, C.c_float(p_dec), C.c_uint64(seed), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ev[1].record()
    torch.cuda.synchronize()
    print('dchain rc', rc, 'err', int(err.item()), 'us/step %.2f' % (ev[0].elapsed_time(ev[1]) * 1e3 / T), flush=True)
d = (DGD2 - k['DGD']).abs()
print('DGD max diff %.3e (scale %.3e) nan %d' % (d[~torch.isnan(d)].max().item() if (~torch.isnan(d)).any() else -1, k['DGD'].abs().max().item(), int(torch.isnan(DGD2).sum())))
for t in (T - 1, T - 2, max(0, T // 2), 0):
    print('  t=%d diff %.3e' % (t, (DGD2[t] - k['DGD'][t]).abs().max().item()))
