import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tacotron2-vae_amd'))
import torch, torch.nn.functional as F
import t2v_hip
lib = t2v_hip.load_library()
lib.t2v_conv1d_x3_set_mode(int(os.environ.get('CXMODE', '2')))
g = torch.Generator().manual_seed(1)
B, Cin, Cout, T = 16, 512, 512, 400
x = torch.randn(B, Cin, T, generator=g); w = torch.randn(Cout, Cin, 5, generator=g) / 50; b = torch.randn(Cout, generator=g)
ref = F.conv1d(x.double(), w.double(), b.double(), padding=2)
gx, gw, gb = x.cuda(), w.cuda(), b.cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: C.c_void_p(t.data_ptr())
nblk = lib.t2v_conv1d_stat_blocks(B, T, Cin, Cout, 5)
for rep in range(3):
    y = torch.full((B, Cout, T), float('nan'), device='cuda'); part = torch.zeros(nblk, Cout, 2, device='cuda')
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    rc = lib.t2v_conv1d_fwd(p(gw), p(gx), p(gb), p(y), p(part), B, Cin, T, Cout, 5, st)
    e1.record(); torch.cuda.synchronize()
    print('rc', rc, 'nblk', nblk, 'us %.1f' % (e0.elapsed_time(e1) * 1e3), 'max err', (y.cpu().double() - ref).abs().max().item(), 'nan', torch.isnan(y).sum().item())
