"""fp32 MFMA GEMM (t2v_gemm_f32) and the LinearHIP autograd wrapper vs torch fp32 CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(2406, 81, 1536), (2406, 256, 80), (504, 128, 512), (7, 5, 3), (65, 129, 33),
                                   # large-tile kernel (M, N >= 128, 16-byte runs): ragged edges, K not a multiple of 16
                                   (2400, 4096, 256), (1340, 1536, 2404), (2000, 1028, 52), (4096, 640, 20),
                                   # split-K (few tiles, deep K): Prenet / projection / d_wq weight-gradient shapes
                                   (256, 80, 2406), (81, 1536, 2400), (128, 1024, 2400), (42, 256, 768), (256, 256, 2406)])
def test_gemm_forms(M, N, K):
    import t2v_hip
    g = torch.Generator().manual_seed(M + N + K)
    A, B, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ref = A @ B.t() + bias
    dA, dB = A.cuda(), B.cuda()
    out = t2v_hip.gemm(dA, dB, bias.cuda())
    tol = 2e-6 * K ** 0.5 * 10
    assert (out.cpu() - ref).abs().max() < tol * ref.abs().max()
    # strided operands: NN and TN forms
    out2 = t2v_hip.gemm(dA, dB.t().contiguous().t())                   # B given column-major
    assert (out2.cpu() - A @ B.t()).abs().max() < tol * ref.abs().max()
    At = dA.t().contiguous()                                           # (K,M): A passed as a transposed view
    out3 = t2v_hip.gemm(At.t(), dB, accumulate=False, relu=True)
    assert (out3.cpu() - torch.relu(A @ B.t())).abs().max() < tol * ref.abs().max()
    acc = out3.clone()
    t2v_hip.gemm(dA, dB, out=acc, accumulate=True)
    assert (acc.cpu() - (torch.relu(A @ B.t()) + A @ B.t())).abs().max() < 2 * tol * ref.abs().max()


def test_linear_autograd_with_relu_dropout():
    import t2v_hip
    g = torch.Generator().manual_seed(0)
    x = torch.randn(401, 6, 80, generator=g)
    w = torch.randn(256, 80, generator=g) * 0.1
    cx, cw = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    gx, gw = x.clone().cuda().requires_grad_(True), w.clone().cuda().requires_grad_(True)
    y = t2v_hip.LinearHIP.apply(gx, gw, None, True, 0.5, 123, 48, 7)
    keep = (y != 0).float().cpu()
    ref_pre = torch.relu(cx @ cw.t())
    live = ref_pre > 0
    assert abs(keep[live].mean().item() - 0.5) < 0.02                   # Bernoulli(0.5) on the active units
    ref = ref_pre * keep * 2.0                                          # same mask, 1/(1-p) scaling
    assert (y.cpu() - ref).abs().max() < 1e-4
    wo = torch.randn(401, 6, 256, generator=g)
    (y * wo.cuda()).sum().backward()
    (ref * wo).sum().backward()
    assert (gx.grad.cpu() - cx.grad).abs().max() < 2e-3 * cx.grad.abs().max()
    assert (gw.grad.cpu() - cw.grad).abs().max() < 2e-3 * cw.grad.abs().max()
    # identical (seed, stream, t) ⇒ identical mask; different t ⇒ different mask
    y2 = t2v_hip.LinearHIP.apply(gx, gw, None, True, 0.5, 123, 48, 7)
    y3 = t2v_hip.LinearHIP.apply(gx, gw, None, True, 0.5, 123, 48, 8)
    assert torch.equal(y, y2) and not torch.equal(y, y3)


def test_gemm_big_tile_throughput_shape():
    """the deferred decoder-LSTM weight gradient DGD^T·X at the bench shape: (4096 x 2400)·(2400 x 2560), TN form"""
    import t2v_hip
    g = torch.Generator().manual_seed(5)
    dg = (torch.randn(2400, 4096, generator=g) * 0.1).cuda()
    x = torch.randn(2400, 2560, generator=g).cuda()
    out = t2v_hip.gemm(dg.t(), x.t())
    ref = (dg.double().t() @ x.double()).float()
    assert out.shape == (4096, 2560)
    assert (out - ref).abs().max() < 1e-4 * ref.abs().max()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        t2v_hip.gemm(dg.t(), x.t(), out=out)
    ev1.record()
    torch.cuda.synchronize()
    tf = 10 * 2 * 4096 * 2400 * 2560 / (ev0.elapsed_time(ev1) * 1e-3) / 1e12
    print('k_gemm_f32_big TN 4096x2560x2400: %.1f TFLOP/s (fp32 MFMA peak 157)' % tf)
    lib0, lib1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib0.record()
    for _ in range(10):
        ref2 = dg.t() @ x
    lib1.record()
    torch.cuda.synchronize()
    print('library GEMM same shape: %.1f TFLOP/s' % (10 * 2 * 4096 * 2400 * 2560 / (lib0.elapsed_time(lib1) * 1e-3) / 1e12))
    assert tf > 40.0


@pytest.mark.parametrize("M,N", [(504, 1024), (2400, 4096), (400, 64512), (7, 5), (1, 300), (333, 1), (4099, 130)])
def test_colsum_matches_torch(M, N):
    """t2v_colsum (bias gradients, sum over decoder steps) vs a float64 column sum; strided rows too"""
    import t2v_hip
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, N, generator=g)
    ref = x.double().sum(0)
    out = t2v_hip.colsum(x.cuda())
    assert out.shape == (N,)
    assert (out.cpu().double() - ref).abs().max() < 2e-6 * M ** 0.5 * 10 + 1e-6
    assert torch.equal(out, t2v_hip.colsum(x.cuda()))              # fixed summation order: bit-reproducible
    wide = torch.randn(M, N + 12, generator=g).cuda()
    view = wide[:, 4:4 + N]                                          # row stride N + 12, base offset 16 bytes
    assert (t2v_hip.colsum(view).cpu().double() - view.cpu().double().sum(0)).abs().max() < 2e-6 * M ** 0.5 * 10 + 1e-6


def test_gemm_writes_column_block_of_wider_matrix():
    """out may be a column block (row stride ldc > N): the decoder's weight-gradient GEMMs write the halves of
    [weight_ih | weight_hh] products into their own tensors"""
    import t2v_hip
    g = torch.Generator().manual_seed(3)
    A, B = torch.randn(4096, 600, generator=g), torch.randn(512, 600, generator=g)
    big = torch.full((4096, 768), 7.0).cuda()
    t2v_hip.gemm(A.cuda(), B.cuda(), out=big[:, 256:])
    ref = A @ B.t()
    assert torch.equal(big[:, :256].cpu(), torch.full((4096, 256), 7.0))
    assert (big[:, 256:].cpu() - ref).abs().max() < 2e-4 * ref.abs().max()
    small = torch.zeros(40, 100).cuda()
    t2v_hip.gemm(A[:40, :64].cuda(), B[:30, :64].cuda(), out=small[:, 10:40])
    assert (small[:, 10:40].cpu() - A[:40, :64] @ B[:30, :64].t()).abs().max() < 1e-4
    assert float(small[:, :10].abs().max()) == 0.0 and float(small[:, 40:].abs().max()) == 0.0


def test_bf16_big_tile_gemm_for_the_lstm_weight_gradients():
    """bf16_run: DGA^T·X / DGD^T·X (both operands stored k-major) on the own 128x128x32 bf16 MFMA GEMM
    (k_gemm_bf16_big_rr) — result == fp32 product of the bf16-rounded operands, incl. accumulate, a column-block output
    (ldc > N) and a K that is not a multiple of the k-tile."""
    import t2v_hip
    g = torch.Generator().manual_seed(3)
    K, M, N = 2400 + 8, 512, 1536
    dg = torch.randn(K, M, generator=g).cuda()
    x = torch.randn(K, 2560, generator=g).cuda()
    want = (dg.bfloat16().float().t().double() @ x[:, 1024:1024 + N].bfloat16().float().double()).float()
    t2v_hip.set_bf16(True)
    try:
        wide = torch.zeros(M, N + 256, device='cuda')
        out = wide[:, 128:128 + N]
        t2v_hip.gemm(dg.t(), x[:, 1024:1024 + N].t(), out=out)
        assert (out - want).abs().max().item() < 2e-3 * want.abs().max().item()
        assert wide[:, :128].abs().max().item() == 0.0 and wide[:, 128 + N:].abs().max().item() == 0.0
        t2v_hip.gemm(dg.t(), x[:, 1024:1024 + N].t(), out=out, accumulate=True)
        assert (out - 2 * want).abs().max().item() < 4e-3 * want.abs().max().item()
    finally:
        t2v_hip.set_bf16(False)


@pytest.mark.parametrize("B,T_in,T", [(6, 84, 400), (3, 21, 37), (1, 200, 5)])
def test_batched_gemm_for_the_per_item_memory_gradient(B, T_in, T):
    """d_memory[b] = alignments_b^T (T_in x T) · d_ctx_b (T x 512) for all items in one launch (t2v_gemm_f32_batched), on the
    strided views the decoder's reverse pass hands over: AL (T+1, B, T_in) rows 1.., DCTX (T, B, 512)."""
    import ctypes as C
    import t2v_hip
    lib = t2v_hip.load_library()
    g = torch.Generator().manual_seed(B * 1000 + T)
    AL, DCTX = torch.rand(T + 1, B, T_in, generator=g), torch.randn(T, B, 512, generator=g)
    ref = torch.einsum('tbj,tbe->bje', AL[1:], DCTX)
    dAL, dD = AL.cuda(), DCTX.cuda()
    out = torch.full((B, T_in, 512), float('nan'), device='cuda')
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.t2v_gemm_f32_batched(p(dAL[1:]), T_in, 1, B * T_in, p(dD), 512, 1, B * 512, p(out), T_in * 512, 512, B, T_in, 512, T,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert (out.cpu() - ref).abs().max() < 2e-5 * T ** 0.5 * ref.abs().max()
    # argument checks: nothing is launched for a bad batch count / leading dimension
    assert lib.t2v_gemm_f32_batched(p(dAL), T_in, 1, B * T_in, p(dD), 512, 1, B * 512, p(out), T_in * 512, 100, B, T_in, 512, T, None) != 0
    assert lib.t2v_gemm_f32_batched(p(dAL), T_in, 1, B * T_in, p(dD), 512, 1, B * 512, p(out), T_in * 512, 512, 0, T_in, 512, T, None) != 0


@pytest.mark.parametrize("M,N,K,form", [(4096, 2560, 2400, 'rr'), (4096, 1536, 2400, 'rr'), (2400, 4096, 256, 'kk'), (1340, 1536, 2404, 'rr'),
                                        (2000, 1028, 52, 'rr'), (2400, 1024, 4096, 'kr'), (1024, 2048, 512, 'rk'), (1156, 1284, 40, 'kk')])
def test_x3_gemm_is_fp32_class(M, N, K, form):
    """Round 6: the large fp32 products on the bf16 matrix cores (k_gemm_f32x3_big: operands cut exactly into three bf16 values, six
    MFMAs per k-block, fp32 accumulation) against an fp64 product — held to the SAME bound as the fp32-MFMA kernel
    (v_mfma_f32_32x32x2_f32), and measured next to it: the x3 error must not exceed 1.5x the native kernel's (it is smaller in
    practice: the running sum is rounded 6 times per 16 k instead of once per k).  Operand forms: r = contiguous along its row index
    (weight-gradient operands stored (K, rows)), k = contiguous along k; incl. ragged edges, K not a multiple of the k-tile,
    accumulate, bias and the split-K path; bit-reproducible."""
    import t2v_hip
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    # wide dynamic range: magnitudes over 2^+-6 so that a dropped low-order term would show
    def rnd(r, c):
        return (torch.randn(r, c, generator=g) * torch.exp2(torch.randint(-6, 7, (r, c), generator=g).float())).cuda()
    A = rnd(M, K) if form[0] == 'k' else rnd(K, M).t()
    B = rnd(N, K) if form[1] == 'k' else rnd(K, N).t()
    bias = torch.randn(N, generator=g).cuda()
    ref = A.double() @ B.double().t() + bias.double()
    absref = A.double().abs() @ B.double().abs().t() + bias.double().abs()    # sum_k |a b| (+ |bias|): what fp32 round-off scales with
    errs = {}
    prev = t2v_hip.set_f32_gemm_mode(None)
    try:
        for mode in (True, False):
            t2v_hip.set_f32_gemm_mode(mode)
            out = t2v_hip.gemm(A, B, bias)
            out2 = t2v_hip.gemm(A, B, bias)
            assert torch.equal(out, out2), 'not reproducible'
            errs[mode] = ((out.double() - ref).abs() / absref).max().item()
            acc = out.clone()
            t2v_hip.gemm(A, B, None, out=acc, accumulate=True)
            e2 = ((acc.double() - (2 * ref - bias.double())).abs() / absref).max().item()
            assert e2 < 2.5 * max(errs[mode], 1e-7), (mode, e2)
    finally:
        t2v_hip.set_f32_gemm_mode(prev)
    print('GEMM %dx%dx%d %s: max |err| / sum|ab|  x3 %.2e  fp32-MFMA %.2e' % (M, N, K, form, errs[True], errs[False]))
    bound = 2e-7 * max(4.0, K ** 0.5)            # fp32 round-off through K terms (max over up to 1e7 outputs, heavy-tailed operands)
    assert errs[False] < bound and errs[True] < bound, errs
    assert errs[True] < 1.1 * errs[False] + 2e-8, errs     # measured: 0.3x .. 0.8x of the fp32-MFMA kernel's error


def test_x3_gemm_throughput_on_the_lstm_weight_gradient_shapes():
    """the four deferred LSTM weight gradients of the fp32 step (K = T*B = 2400, operands stored (K, rows)) on the x3 kernel next to
    the fp32-MFMA kernel it replaces (printed; the x3 kernel must not be slower)"""
    import t2v_hip
    g = torch.Generator().manual_seed(5)
    res = {}
    prev = t2v_hip.set_f32_gemm_mode(None)
    try:
        for (M, N) in ((4096, 2560), (4096, 1536), (4096, 1024), (4096, 512)):
            dg = (torch.randn(2400, M, generator=g) * 0.1).cuda()
            x = torch.randn(2400, N, generator=g).cuda()
            out = torch.empty(M, N, device='cuda')
            for mode in (True, False):
                t2v_hip.set_f32_gemm_mode(mode)
                for _ in range(3):
                    t2v_hip.gemm(dg.t(), x.t(), out=out)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(20):
                    t2v_hip.gemm(dg.t(), x.t(), out=out)
                ev1.record()
                torch.cuda.synchronize()
                res[(M, N, mode)] = 20 * 2 * M * N * 2400 / (ev0.elapsed_time(ev1) * 1e-3) / 1e12
            print('dW GEMM %dx%dx2400: x3 %.0f TFLOP/s (fp32-equivalent), fp32-MFMA %.0f TFLOP/s' % (M, N, res[(M, N, True)], res[(M, N, False)]))
        assert res[(4096, 2560, True)] > res[(4096, 2560, False)]
    finally:
        t2v_hip.set_f32_gemm_mode(prev)


@pytest.mark.parametrize("TB", [2400, 1237])
def test_grouped_gemm_for_the_lstm_weight_gradients(TB):
    """Round 6: the decoder's five LSTM weight-gradient products as ONE launch (t2v_gemm_f32_grouped: both gate-gradient operands split
    once, the column blocks [prenet | h_att | ctx] / [h_att + ctx | h_dec] of a product routed to their own tensors, incl. a column
    block of a wider matrix) against fp64 and against the products issued one by one; accumulate; x3 and fp32-MFMA mode;
    bit-reproducible."""
    import t2v_hip
    g = torch.Generator().manual_seed(TB)
    dga, dgd = (torch.randn(TB, 4096, generator=g) * 0.1).cuda(), (torch.randn(TB, 4096, generator=g) * 0.1).cuda()
    xs = torch.randn(TB + 1, 2560, generator=g).cuda()
    pre = torch.randn(TB, 256, generator=g).cuda()
    x_prev, x_cur = xs[:TB], xs[1:]

    def run(accumulate, seed_out):
        w_ih_att = torch.full((4096, 768), seed_out, device='cuda')
        w_hh_att = torch.full((4096, 1024), seed_out, device='cuda')
        w_ih_dec = torch.full((4096, 1536), seed_out, device='cuda')
        w_hh_dec = torch.full((4096, 1024), seed_out, device='cuda')
        t2v_hip.gemm_grouped([(dga.t(), [(pre.t(), w_ih_att[:, :256]), (x_prev[:, :1024].t(), w_hh_att), (x_prev[:, 1024:1536].t(), w_ih_att[:, 256:])]),
                              (dgd.t(), [(x_cur[:, :1536].t(), w_ih_dec), (x_cur[:, 1536:].t(), w_hh_dec)])], accumulate=accumulate)
        return w_ih_att, w_hh_att, w_ih_dec, w_hh_dec

    want = [torch.cat([dga.double().t() @ pre.double(), dga.double().t() @ x_prev[:, 1024:1536].double()], 1), dga.double().t() @ x_prev[:, :1024].double(),
            dgd.double().t() @ x_cur[:, :1536].double(), dgd.double().t() @ x_cur[:, 1536:].double()]
    prev = t2v_hip.set_f32_gemm_mode(None)
    try:
        for mode in (True, False):
            t2v_hip.set_f32_gemm_mode(mode)
            got = run(False, float('nan'))
            again = run(False, float('nan'))
            for a, b, w in zip(got, again, want):
                assert torch.equal(a, b)
                assert (a.double() - w).abs().max().item() < 2e-5 * w.abs().max().item()
            acc = run(True, 0.5)
            for a, w in zip(acc, want):
                assert (a.double() - (w + 0.5)).abs().max().item() < 2e-5 * w.abs().max().item()
            # the same products one by one
            one = torch.empty(4096, 1536, device='cuda')
            t2v_hip.gemm(dgd.t(), x_cur[:, :1536].t(), out=one)
            assert (one - got[2]).abs().max().item() < 2e-6 * want[2].abs().max().item()
    finally:
        t2v_hip.set_f32_gemm_mode(prev)
