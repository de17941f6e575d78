"""CPU oracle for the Tacotron2-VAE hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

A functional restatement (plain torch fp32 CPU ops + autograd) of the reference
algorithm for SURVEY.md §8(a) rows a-8 … a-20, written against a *state dict*
with the reference's 142 keys.  Every function cites the reference file:line it
follows.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this file; nothing under `tacotron2-vae_amd/`
does, and the product path raises when its HIP library is missing.

Parity pinning: `oracle/gen_golden.py` runs the real reference (imported in the
build container through `oracle/ref_shims`) on seeded inputs and writes
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file against
those vectors on CPU.  Stochastic pieces are made explicit inputs here:
`eps` (VAE reparameterisation noise, modules.py:19) and dropout keep-masks
(`drop` dict) — parity runs use dropout 0 and injected eps, as SURVEY §7 says.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- helpers
def get_mask_from_lengths(lengths, max_len=None):
    """utils.py:9-13 — True where position < length."""
    if max_len is None:
        max_len = int(lengths.max().item())
    return torch.arange(max_len, device=lengths.device)[None, :] < lengths[:, None]


def _bn(x, sd, prefix, training, stats_out=None):
    """BatchNorm1d/2d, SURVEY Appendix C: train = biased batch variance, eps 1e-5."""
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    if training:
        return F.batch_norm(x, None, None, w, b, True, 0.0, 1e-5)
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        w, b, False, 0.0, 1e-5)


def _apply_keep(x, keep, p):
    """F.dropout with an explicit keep mask (Appendix C: kept values × 1/(1-p))."""
    if keep is None or p == 0.0:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """nn.LSTMCell (Appendix C): gates stacked i,f,g,o on dim 0."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    H = h.shape[1]
    i, f, g, o = gates[:, :H], gates[:, H:2 * H], gates[:, 2 * H:3 * H], gates[:, 3 * H:]
    i, f, o = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o)
    g = torch.tanh(g)
    c2 = f * c + i * g
    return o * torch.tanh(c2), c2


# --------------------------------------------------------------------------- encoder (a-9, a-10)
def encoder_forward(sd, text, input_lengths, training=True, drop=None, p_conv=0.5):
    """model.py:528 + Encoder.forward model.py:175-192 (packed BiLSTM == per-sequence lengths)."""
    x = F.embedding(text, sd['transcript_embedding.weight']).transpose(1, 2)
    for i in range(3):
        pre = 'encoder.convolutions.%d' % i
        x = F.conv1d(x, sd[pre + '.0.conv.weight'], sd[pre + '.0.conv.bias'], padding=2)
        x = F.relu(_bn(x, sd, pre + '.1', training))
        if training:
            x = _apply_keep(x, None if drop is None else drop.get('enc%d' % i), p_conv)
    x = x.transpose(1, 2)  # (B, T, 512)
    B, T, _ = x.shape
    H = sd['encoder.lstm.weight_hh_l0'].shape[1]
    out = x.new_zeros(B, T, 2 * H)
    outs = []
    for suffix, reverse in (('', False), ('_reverse', True)):
        w_ih, w_hh = sd['encoder.lstm.weight_ih_l0' + suffix], sd['encoder.lstm.weight_hh_l0' + suffix]
        b_ih, b_hh = sd['encoder.lstm.bias_ih_l0' + suffix], sd['encoder.lstm.bias_hh_l0' + suffix]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        hs = [None] * T
        steps = range(T - 1, -1, -1) if reverse else range(T)
        for t in steps:
            valid = (t < input_lengths).to(x.dtype)[:, None]  # packed: only own length
            h2, c2 = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            h = valid * h2 + (1 - valid) * h
            c = valid * c2 + (1 - valid) * c
            hs[t] = valid * h2  # pad_packed_sequence → zeros
        outs.append(torch.stack(hs, 1))
    out = torch.cat(outs, -1)
    return out


# --------------------------------------------------------------------------- VAE / reference encoder (a-11)
def add_coords(x):
    """CoordConv.py:37-74 rank-2, with_r=True. x: (N,1,H,W); xx along H, yy along W."""
    N, _, Hh, Ww = x.shape
    xx = torch.arange(Hh, dtype=torch.int32).float() / (Hh - 1)
    yy = torch.arange(Ww, dtype=torch.int32).float() / (Ww - 1)
    xx = (xx * 2 - 1)[None, None, :, None].expand(N, 1, Hh, Ww)
    yy = (yy * 2 - 1)[None, None, None, :].expand(N, 1, Hh, Ww)
    rr = torch.sqrt((xx - 0.5) ** 2 + (yy - 0.5) ** 2)
    return torch.cat([x, xx, yy, rr], 1)


def gru_last(x, w_ih, w_hh, b_ih, b_hh):
    """nn.GRU batch_first, last hidden (Appendix C; modules.py:78-80)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    for t in range(T):
        gi = x[:, t] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
    return h


def vae_gst_forward(sd, mel, training=True, eps=None):
    """VAE_GST.forward modules.py:24-31; ReferenceEncoder.forward modules.py:65-80.
    NB modules.py:67: (B,80,T) memory is *reinterpreted* as (B,1,T,80) — no transpose."""
    N = mel.shape[0]
    out = mel.contiguous().view(N, 1, -1, 80)
    out = add_coords(out)
    for i in range(6):
        wk = 'vae_gst.ref_encoder.convs.%d.%s' % (i, 'conv.weight' if i == 0 else 'weight')
        bk = 'vae_gst.ref_encoder.convs.%d.%s' % (i, 'conv.bias' if i == 0 else 'bias')
        out = F.conv2d(out, sd[wk], sd[bk], stride=2, padding=1)
        out = F.relu(_bn(out, sd, 'vae_gst.ref_encoder.bns.%d' % i, training))
    out = out.transpose(1, 2)
    T = out.shape[1]
    out = out.contiguous().view(N, T, -1)
    h = gru_last(out, sd['vae_gst.ref_encoder.gru.weight_ih_l0'], sd['vae_gst.ref_encoder.gru.weight_hh_l0'],
                 sd['vae_gst.ref_encoder.gru.bias_ih_l0'], sd['vae_gst.ref_encoder.gru.bias_hh_l0'])
    mu = h @ sd['vae_gst.fc1.weight'].t() + sd['vae_gst.fc1.bias']
    logvar = h @ sd['vae_gst.fc2.weight'].t() + sd['vae_gst.fc2.bias']
    if training:
        if eps is None:
            eps = torch.randn_like(mu)
        z = eps * torch.exp(0.5 * logvar) + mu
    else:
        z = mu
    style = z @ sd['vae_gst.fc3.weight'].t() + sd['vae_gst.fc3.bias']
    return style, mu, logvar, z


# --------------------------------------------------------------------------- decoder (a-12 … a-16)
def prenet_forward(sd, x, drop=None, p=0.5):
    """Prenet.forward model.py:99-102 — dropout is on even at inference (training=True)."""
    for i in range(2):
        x = F.relu(x @ sd['decoder.prenet.layers.%d.linear_layer.weight' % i].t())
        x = _apply_keep(x, None if drop is None else drop.get('prenet%d' % i), p)
    return x


class DecoderState(object):
    pass


def decoder_init(sd, memory, mask):
    """Decoder.initialize_decoder_states model.py:260-291."""
    B, T_in, _ = memory.shape
    st = DecoderState()
    z = lambda n: memory.new_zeros(B, n)
    st.h_att, st.c_att, st.h_dec, st.c_dec = z(1024), z(1024), z(1024), z(1024)
    st.alpha, st.alpha_cum, st.ctx = z(T_in), z(T_in), z(512)
    st.memory = memory
    st.pm = memory @ sd['decoder.attention_layer.memory_layer.linear_layer.weight'].t()
    st.mask = mask
    return st


def decoder_step(sd, st, prenet_out, p_att=0.0, p_dec=0.0, keeps=None):
    """Decoder.decode model.py:346-389 + Attention.forward model.py:67-88.
    `keeps`: optional dict of keep-masks 'att_h','att_c','dec_h','dec_c' (B,1024)."""
    A = 'decoder.attention_layer.'
    x = torch.cat((prenet_out, st.ctx), -1)
    h, c = lstm_cell(x, st.h_att, st.c_att, sd['decoder.attention_rnn.weight_ih'],
                     sd['decoder.attention_rnn.weight_hh'], sd['decoder.attention_rnn.bias_ih'],
                     sd['decoder.attention_rnn.bias_hh'])
    k = keeps or {}
    st.h_att = _apply_keep(h, k.get('att_h'), p_att)
    st.c_att = _apply_keep(c, k.get('att_c'), p_att)
    # location-sensitive attention
    cat = torch.stack((st.alpha, st.alpha_cum), 1)                       # (B,2,T_in)
    q = st.h_att @ sd[A + 'query_layer.linear_layer.weight'].t()          # (B,128)
    loc = F.conv1d(cat, sd[A + 'location_layer.location_conv.conv.weight'], None, padding=15)
    loc = loc.transpose(1, 2) @ sd[A + 'location_layer.location_dense.linear_layer.weight'].t()
    e = torch.tanh(q[:, None, :] + loc + st.pm) @ sd[A + 'v.linear_layer.weight'].t()
    e = e.squeeze(-1)
    if st.mask is not None:
        e = e.masked_fill(st.mask, -float('inf'))
    st.alpha = F.softmax(e, dim=1)
    st.ctx = torch.bmm(st.alpha[:, None, :], st.memory).squeeze(1)
    st.alpha_cum = st.alpha_cum + st.alpha
    x = torch.cat((st.h_att, st.ctx), -1)
    h, c = lstm_cell(x, st.h_dec, st.c_dec, sd['decoder.decoder_rnn.weight_ih'],
                     sd['decoder.decoder_rnn.weight_hh'], sd['decoder.decoder_rnn.bias_ih'],
                     sd['decoder.decoder_rnn.bias_hh'])
    st.h_dec = _apply_keep(h, k.get('dec_h'), p_dec)
    st.c_dec = _apply_keep(c, k.get('dec_c'), p_dec)
    hc = torch.cat((st.h_dec, st.ctx), 1)
    mel = hc @ sd['decoder.linear_projection.linear_layer.weight'].t() + sd['decoder.linear_projection.linear_layer.bias']
    gate = hc @ sd['decoder.gate_layer.linear_layer.weight'].t() + sd['decoder.gate_layer.linear_layer.bias']
    return mel, gate, st.alpha


def decoder_forward(sd, memory, mels, memory_lengths, p_att=0.0, p_dec=0.0, drop=None,
                    p_prenet=0.5):
    """Decoder.forward model.py:391-426 (teacher forcing).  mels: (B,80,T_out)."""
    B = memory.shape[0]
    inp = torch.cat((mels.new_zeros(1, B, 80), mels.permute(2, 0, 1)), 0)   # go frame + frames
    pre = prenet_forward(sd, inp, drop, p_prenet)
    st = decoder_init(sd, memory, ~get_mask_from_lengths(memory_lengths, memory.shape[1]))
    outs, gates, aligns = [], [], []
    for t in range(inp.shape[0] - 1):
        keeps = None if drop is None or 'lstm' not in drop else drop['lstm'][t]
        m, g, a = decoder_step(sd, st, pre[t], p_att, p_dec, keeps)
        outs.append(m), gates.append(g.squeeze(1)), aligns.append(a)
    mel = torch.stack(outs).permute(1, 2, 0).contiguous()     # (B,80,T)
    gate = torch.stack(gates).transpose(0, 1).contiguous()    # (B,T)
    align = torch.stack(aligns).transpose(0, 1)               # (B,T,T_in)
    return mel, gate, align


def decoder_inference(sd, memory, max_steps=1000, gate_threshold=0.5, prenet_keep=None,
                      p_prenet=0.5, stop_on_gate=True):
    """Decoder.inference model.py:428-464 (B must be 1 for the stop rule, Appendix B-12).
    prenet_keep: optional callable t -> {'prenet0': mask, 'prenet1': mask} (None = no dropout)."""
    B = memory.shape[0]
    st = decoder_init(sd, memory, None)
    x = memory.new_zeros(B, 80)
    outs, gates, aligns = [], [], []
    while True:
        drop = prenet_keep(len(outs)) if prenet_keep is not None else None
        pre = prenet_forward(sd, x, drop, p_prenet if drop is not None else 0.0)
        m, g, a = decoder_step(sd, st, pre)
        outs.append(m), gates.append(g), aligns.append(a)
        if stop_on_gate and bool((torch.sigmoid(g) > gate_threshold).all()):
            break
        if len(outs) == max_steps:
            break
        x = m
    mel = torch.stack(outs).permute(1, 2, 0).contiguous()
    gate = torch.stack(gates).transpose(0, 1).contiguous()    # (B,T,1)
    align = torch.stack(aligns).transpose(0, 1)
    return mel, gate, align


# --------------------------------------------------------------------------- postnet (a-17)
def postnet_forward(sd, x, training=True, drop=None, p=0.5):
    """Postnet.forward model.py:143-148."""
    for i in range(5):
        pre = 'postnet.convolutions.%d' % i
        x = F.conv1d(x, sd[pre + '.0.conv.weight'], sd[pre + '.0.conv.bias'], padding=2)
        x = _bn(x, sd, pre + '.1', training)
        if i < 4:
            x = torch.tanh(x)
        if training:
            x = _apply_keep(x, None if drop is None else drop.get('post%d' % i), p)
    return x


# --------------------------------------------------------------------------- whole model (a-8, a-18)
def tacotron2_forward(sd, text, input_lengths, mels, output_lengths, training=True, eps=None,
                      p_att=0.0, p_dec=0.0, drop=None, p_conv=0.0, p_prenet=0.0,
                      quirk_inplace_mask=True):
    """Tacotron2.forward model.py:522-547 + parse_output 509-520.

    quirk_inplace_mask reproduces Appendix B-5: the reference zero-fills padded frames of
    the decoder mel *in place on .data* (model.py:515-517) after Postnet conv-0 saved that very
    tensor for backward, so conv-0's weight gradient is computed from the masked tensor.  We do
    literally the same (.data masked_fill_); False gives the clean out-of-place graph.
    """
    enc = encoder_forward(sd, text, input_lengths, training, drop, p_conv)
    style, mu, logvar, z = vae_gst_forward(sd, mels, training, eps)
    memory = enc + style[:, None, :]
    mel, gate, align = decoder_forward(sd, memory, mels, input_lengths, p_att, p_dec, drop, p_prenet)
    post = mel + postnet_forward(sd, mel, training, drop, p_conv)
    pad = ~get_mask_from_lengths(output_lengths, mels.shape[2])          # (B,T) True at padding
    if quirk_inplace_mask:
        mel.data.masked_fill_(pad[:, None, :], 0.0)
        post.data.masked_fill_(pad[:, None, :], 0.0)
        gate.data.masked_fill_(pad, 1e3)
    else:
        mel = mel.masked_fill(pad[:, None, :], 0.0)
        post = post.masked_fill(pad[:, None, :], 0.0)
        gate = gate.masked_fill(pad, 1e3)
    return [mel, post, gate, align, mu, logvar, z]


# --------------------------------------------------------------------------- loss (a-19)
def kl_weight(anneal_function, step, lag=50000, k=0.0025, x0=10000, upper=0.2):
    """loss_function.py:15-24."""
    if anneal_function == 'logistic':
        import numpy as np
        return float(upper / (upper + np.exp(-k * (step - x0))))
    if anneal_function == 'linear':
        return min(upper, step / x0) if step > lag else 0
    if anneal_function == 'constant':
        return 0.001
    return None


def loss_forward(outputs, mel_target, gate_target, step, anneal_function='constant', **kw):
    """Tacotron2Loss_VAE.forward loss_function.py:27-44."""
    mel, post, gate, _, mu, logvar = outputs[:6]
    mel_loss = F.mse_loss(mel, mel_target) + F.mse_loss(post, mel_target)
    gate_loss = F.binary_cross_entropy_with_logits(gate.reshape(-1, 1), gate_target.reshape(-1, 1))
    kl = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
    w = kl_weight(anneal_function, step, **kw)
    recon = mel_loss + gate_loss
    return recon + w * kl, recon, kl, w


# --------------------------------------------------------------------------- optimiser (a-20)
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics (Appendix C)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=1e-6):
    """torch.optim.Adam (L2-in-grad weight decay), Appendix C."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


# --------------------------------------------------------------------------- front end (a-1 … a-3)
def hann_periodic(n):
    return 0.5 - 0.5 * torch.cos(2 * math.pi * torch.arange(n, dtype=torch.float64) / n)


def slaney_mel_basis(sr=16000, n_fft=1024, n_mels=80, fmin=0.0, fmax=8000.0):
    """librosa 0.6.0 filters.mel(htk=False, norm=1) — layers.py:62-64 (float64 → float32)."""
    def hz2mel(f):
        f = torch.as_tensor(f, dtype=torch.float64)
        lin = f / (200.0 / 3)
        log = 15.0 + torch.log(torch.clamp(f, min=1e-10) / 1000.0) / (math.log(6.4) / 27.0)
        return torch.where(f >= 1000.0, log, lin)

    def mel2hz(m):
        lin = m * (200.0 / 3)
        log = 1000.0 * torch.exp((math.log(6.4) / 27.0) * (m - 15.0))
        return torch.where(m >= 15.0, log, lin)

    fft_f = torch.linspace(0, sr / 2.0, 1 + n_fft // 2, dtype=torch.float64)
    mel_f = mel2hz(torch.linspace(float(hz2mel(fmin)), float(hz2mel(fmax)), n_mels + 2, dtype=torch.float64))
    fdiff = mel_f[1:] - mel_f[:-1]
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = torch.clamp(torch.minimum(lower, upper), min=0)
    w = w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]
    return w.float()


def mel_spectrogram(y, n_fft=1024, hop=256, mel_basis=None):
    """TacotronSTFT.mel_spectrogram layers.py:75-92 via STFT.transform stft.py:77-105
    (reflect pad n_fft/2, periodic-Hann DFT basis, magnitude) + log(clamp(.,1e-5)).
    y: (B,N) in [-1,1].  Uses an FFT (≡ the dense DFT conv to fp32 roundoff)."""
    assert float(y.min()) >= -1 and float(y.max()) <= 1
    if mel_basis is None:
        mel_basis = slaney_mel_basis(n_fft=n_fft)
    yp = F.pad(y[:, None, :], (n_fft // 2, n_fft // 2), mode='reflect')[:, 0]
    frames = yp.unfold(1, n_fft, hop)                                    # (B,T,n_fft)
    spec = torch.fft.rfft(frames * hann_periodic(n_fft).float(), dim=-1)
    mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2).transpose(1, 2)    # (B,513,T)
    return torch.log(torch.clamp(mel_basis @ mag, min=1e-5))
