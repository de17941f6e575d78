"""Tacotron2-VAE boundary modules for MI355X.

Same attribute tree, `state_dict` keys (142) and call surface as the reference's model.py
(`Tacotron2` 467-547, `Encoder` 151-203, `Decoder` 206-464, `Attention` 31-88,
`LocationLayer` 12-28, `Prenet` 91-102, `Postnet` 105-148) so `load_model()`, checkpoints
and the synthesizer call sequence stay drop-in.  The modules only hold parameters (created by
the same stock containers in the same order, so seed 1234 reproduces the reference's step-0
weights) and dispatch the work to the HIP path: the decoder time loop — ≈95 % of the step —
runs in libt2vae_hip (t2v_hip.DecoderCore); there is no CPU fallback.
"""
from math import sqrt

import os

import torch
from torch import nn
from torch.nn import functional as F

import t2v_hip
from layers import ConvNorm, LinearNorm
from modules import VAE_GST
from utils import get_mask_from_lengths, to_gpu

drop_rate = 0.5   # module-level like reference model.py:11 (read at call time)
# Dropout masks of the Prenet / conv blocks are a pure function of (seed, stream, call index, element) — and of the
# device-side step counter (t2v_hip.StepParams.epoch = training iteration) when a training engine installed one, so
# they differ from iteration to iteration also under graph replay and after a checkpoint resume.  The seed derives
# from hparams.seed (set by Tacotron2.__init__).
_drop_seed = [0x5EED]


class LocationLayer(nn.Module):
    def __init__(self, attention_n_filters, attention_kernel_size, attention_dim):
        super().__init__()
        self.location_conv = ConvNorm(2, attention_n_filters, kernel_size=attention_kernel_size,
                                      padding=(attention_kernel_size - 1) // 2, bias=False, stride=1,
                                      dilation=1)
        self.location_dense = LinearNorm(attention_n_filters, attention_dim, bias=False, w_init_gain='tanh')


class Attention(nn.Module):
    """Parameter holder; the arithmetic of reference model.py:45-88 lives in k_attn_fwd/k_attn_bwd."""

    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size):
        super().__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(attention_location_n_filters, attention_location_kernel_size,
                                            attention_dim)
        self.score_mask_value = -float("inf")


class Prenet(nn.Module):
    def __init__(self, in_dim, sizes):
        super().__init__()
        self.layers = nn.ModuleList([LinearNorm(i, o, bias=False) for i, o in zip([in_dim] + sizes[:-1], sizes)])

    def forward(self, x, rng_t=None):
        # dropout stays on at inference too (reference model.py:101 hard-codes training=True);
        # Linear + ReLU + dropout is one MFMA GEMM with a fused epilogue.  rng_t: the mask index of this call (default: the
        # running block counter); the teacher-forced pass hands in a fixed one so that its masks do not depend on where
        # in the forward pass the Prenet is issued
        if rng_t is None:
            _block_calls[0] += 1
            rng_t = _block_calls[0]
        for i, linear in enumerate(self.layers):
            x = t2v_hip.LinearHIP.apply(x, linear.weight, None, True, drop_rate, _drop_seed[0], 48 + i, rng_t)
        return x


def _conv_bn(cin, cout, k, gain):
    return nn.Sequential(ConvNorm(cin, cout, kernel_size=k, stride=1, padding=(k - 1) // 2, dilation=1,
                                  w_init_gain=gain), nn.BatchNorm1d(cout))


_block_calls = [0]


def _conv_bn_act(block, x, act, training, rng_stream):
    """dropout(act(bn(conv(x)))) of one `_conv_bn` block on the HIP conv/BN kernels."""
    conv, bn = block[0].conv, block[1]
    _block_calls[0] += 1
    if training:
        t2v_hip.note_bn_counter(bn.num_batches_tracked)     # bumped in one launch at the end of the forward
    return t2v_hip.ConvBNAct1d.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     training, act, drop_rate, _drop_seed[0], rng_stream, _block_calls[0])


class Postnet(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        n, d, k = hparams.postnet_n_convolutions, hparams.postnet_embedding_dim, hparams.postnet_kernel_size
        blocks = [_conv_bn(hparams.n_mel_channels, d, k, 'tanh')]
        blocks += [_conv_bn(d, d, k, 'tanh') for _ in range(1, n - 1)]
        blocks.append(_conv_bn(d, hparams.n_mel_channels, k, 'linear'))
        self.convolutions = nn.ModuleList(blocks)

    def forward(self, x):
        last = len(self.convolutions) - 1
        for i, block in enumerate(self.convolutions):
            x = _conv_bn_act(block, x, t2v_hip.ACT_TANH if i < last else t2v_hip.ACT_NONE, self.training, 32 + i)
        t2v_hip.flush_bn_counters()
        return x


class Encoder(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        d, k = hparams.encoder_embedding_dim, hparams.encoder_kernel_size
        self.convolutions = nn.ModuleList([_conv_bn(d, d, k, 'relu') for _ in range(hparams.encoder_n_convolutions)])
        self.lstm = nn.LSTM(d, d // 2, 1, batch_first=True, bidirectional=True)

    def _convs(self, x):
        for i, block in enumerate(self.convolutions):
            x = _conv_bn_act(block, x, t2v_hip.ACT_RELU, self.training, 16 + i)
        t2v_hip.flush_bn_counters()
        return x.transpose(1, 2)

    def _bilstm(self, x, lengths):
        l = self.lstm
        return t2v_hip.BiLSTM.apply(x, lengths, l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0,
                                    l.weight_ih_l0_reverse, l.weight_hh_l0_reverse, l.bias_ih_l0_reverse,
                                    l.bias_hh_l0_reverse, torch.is_grad_enabled())

    def forward(self, x, input_lengths):
        """conv bank → BiLSTM over each sequence's own length (packed semantics), zero at padding."""
        x = self._convs(x)
        return self._bilstm(x, lengths_i32(input_lengths, x.device))

    def inference(self, x):
        x = self._convs(x)
        lengths = torch.full((x.size(0),), x.size(1), device=x.device, dtype=torch.int32)
        return self._bilstm(x, lengths)


class Decoder(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.n_mel_channels = hparams.n_mel_channels
        self.n_frames_per_step = hparams.n_frames_per_step
        self.encoder_embedding_dim = hparams.encoder_embedding_dim
        self.attention_rnn_dim = hparams.attention_rnn_dim
        self.decoder_rnn_dim = hparams.decoder_rnn_dim
        self.prenet_dim = hparams.prenet_dim
        self.max_decoder_steps = hparams.max_decoder_steps
        self.gate_threshold = hparams.gate_threshold
        self.p_attention_dropout = hparams.p_attention_dropout
        self.p_decoder_dropout = hparams.p_decoder_dropout
        if self.n_frames_per_step != 1:
            raise NotImplementedError("n_frames_per_step=1 only (as the reference, hparams.py:88)")

        self.prenet = Prenet(hparams.n_mel_channels * hparams.n_frames_per_step,
                             [hparams.prenet_dim, hparams.prenet_dim])
        self.attention_rnn = nn.LSTMCell(hparams.prenet_dim + self.encoder_embedding_dim, hparams.attention_rnn_dim)
        self.attention_layer = Attention(hparams.attention_rnn_dim, self.encoder_embedding_dim,
                                         hparams.attention_dim, hparams.attention_location_n_filters,
                                         hparams.attention_location_kernel_size)
        self.decoder_rnn = nn.LSTMCell(hparams.attention_rnn_dim + self.encoder_embedding_dim,
                                       hparams.decoder_rnn_dim, 1)
        self.linear_projection = LinearNorm(hparams.decoder_rnn_dim + self.encoder_embedding_dim,
                                            hparams.n_mel_channels * hparams.n_frames_per_step)
        self.gate_layer = LinearNorm(hparams.decoder_rnn_dim + self.encoder_embedding_dim, 1, bias=True,
                                     w_init_gain='sigmoid')
        self.dropout_seed = hparams.seed
        self._calls = 0

    # -- helpers kept for the reference's call sequence (synthesizer.py:135-156)
    def get_go_frame(self, memory):
        return memory.new_zeros(memory.size(0), self.n_mel_channels * self.n_frames_per_step)

    def parse_decoder_inputs(self, decoder_inputs):
        x = decoder_inputs.transpose(1, 2)
        x = x.reshape(x.size(0), x.size(1) // self.n_frames_per_step, -1)
        return x.transpose(0, 1)

    def parse_decoder_outputs(self, mel_outputs, gate_outputs, alignments):
        alignments = torch.stack(alignments).transpose(0, 1)
        gate_outputs = torch.stack(gate_outputs)
        if gate_outputs.dim() == 1:
            gate_outputs = gate_outputs.unsqueeze(1)
        gate_outputs = gate_outputs.transpose(0, 1).contiguous()
        mel_outputs = torch.stack(mel_outputs).transpose(0, 1).contiguous()
        mel_outputs = mel_outputs.view(mel_outputs.size(0), -1, self.n_mel_channels).transpose(1, 2)
        return mel_outputs, gate_outputs, alignments

    def _core_weights(self, b_dec=None):
        a, att, dec = self.attention_layer, self.attention_rnn, self.decoder_rnn
        return (att.weight_ih, att.weight_hh, dec.weight_ih, dec.weight_hh, dec.bias_ih + dec.bias_hh if b_dec is None else b_dec,
                a.query_layer.weight, a.location_layer.location_conv.conv.weight,
                a.location_layer.location_dense.weight, a.v.weight)

    # -- free-running decode ------------------------------------------------------------------
    def _session(self, memory, mask, max_steps):
        att, dec, al = self.attention_rnn, self.decoder_rnn, self.attention_layer
        lengths = None if mask is None else (~mask).sum(1).to(device=memory.device, dtype=torch.int32)
        pm = t2v_hip.LinearHIP.apply(memory, al.memory_layer.weight, None, False, 0.0, 0, 0, 0)
        return t2v_hip.InferenceSession(
            memory, pm, lengths, att.weight_ih, att.weight_hh, att.bias_ih + att.bias_hh, dec.weight_ih,
            dec.weight_hh, dec.bias_ih + dec.bias_hh, al.query_layer.weight,
            al.location_layer.location_conv.conv.weight, al.location_layer.location_dense.weight, al.v.weight,
            self.prenet.layers[0].weight, self.prenet.layers[1].weight, self.linear_projection.weight,
            self.linear_projection.bias, self.gate_layer.weight, self.gate_layer.bias, max_steps)

    def initialize_decoder_states(self, memory, mask):
        """reference model.py:260-291 — zero states; stores memory / processed memory / mask."""
        self._sess = self._session(memory, mask, self.max_decoder_steps)
        self.memory, self.processed_memory, self.mask = memory, self._sess.pm, mask
        B = memory.size(0)
        z = lambda n: memory.new_zeros(B, n)
        self.attention_hidden, self.attention_cell = z(self.attention_rnn_dim), z(self.attention_rnn_dim)
        self.decoder_hidden, self.decoder_cell = z(self.decoder_rnn_dim), z(self.decoder_rnn_dim)
        self.attention_weights, self.attention_weights_cum = z(memory.size(1)), z(memory.size(1))
        self.attention_context = z(self.encoder_embedding_dim)

    def decode(self, decoder_input):
        """One step from a prenet output (reference model.py:346-389): returns (mel (B,80), gate (B,1),
        attention weights (B,T_in)) and refreshes the state attributes the reference exposes."""
        s = self._sess
        t = s.t
        if t >= s.max_steps:
            raise RuntimeError("decode() called past max_decoder_steps")
        s.PRE[t].copy_(decoder_input)
        s.run(t, t + 1, self.gate_threshold, 0.0, True, 0)
        s.t = t + 1
        x1, x2 = s.XS[t + 1], s.XS[t + 2]
        self.attention_hidden, self.attention_context = x1[:, :1024], x1[:, 1024:1536]
        self.decoder_hidden = x2[:, 1536:]
        self.attention_cell, self.decoder_cell = s.CA[t + 1], s.CD[t + 1]
        self.attention_weights, self.attention_weights_cum = s.AL[t + 1], s.ACUM[t + 1]
        return s.MEL[t], s.GATE[t].unsqueeze(1), s.AL[t + 1]

    def inference(self, memory, chunk=32, persistent=None):
        """reference model.py:428-464: decode until sigmoid(gate) > gate_threshold or max_decoder_steps.
        The loop runs on the GPU in chunks of `chunk` frames between stop-flag reads.
        More than 8 utterances (the decode kernels take 8 per call; the reference's own loop only works for one) are decoded
        8 at a time, each group until all of ITS gates have fired, and padded to the longest group: mel 0, gate logit +1e3
        (fired), attention weights 0 (round 4; it used to raise)."""
        if memory.size(0) > 8:
            outs = [self.inference(memory[b0:b0 + 8], chunk, persistent) for b0 in range(0, memory.size(0), 8)]
            n = max(o[0].size(2) for o in outs)
            F_pad = torch.nn.functional.pad
            return (torch.cat([F_pad(o[0], (0, n - o[0].size(2))) for o in outs], 0),
                    torch.cat([F_pad(o[1], (0, 0, 0, n - o[1].size(1)), value=1e3) for o in outs], 0),
                    torch.cat([F_pad(o[2], (0, 0, 0, n - o[2].size(1))) for o in outs], 0))
        self.initialize_decoder_states(memory, mask=None)
        s = self._sess
        s.PRE[0].copy_(self.prenet(self.get_go_frame(memory)))
        self._calls += 1
        seed = (int(self.dropout_seed) * 1000003 + self._calls) & 0x7FFFFFFFFFFFFFFF
        n, t = None, 0
        if persistent is None:
            persistent = os.environ.get('T2V_DECODE_PERSISTENT', '1') != '0'
        if persistent and s.persistent_supported():
            # one persistent launch for the whole utterance: weights resident on chip, state handed between CUs as
            # tagged granules, the loop ends on the frame the gate fires
            s.run_persistent(self.gate_threshold, drop_rate, seed)
            stop = int(s.stop.item())
            if s.persistent_timed_out():
                # the 256 workgroups were not co-scheduled (a GPU shared with other work): the bounded spins gave up.
                # The launch-per-stage loop below restarts from the zero state — nothing of the failed run is kept
                print("Warning! persistent decode kernel could not be co-scheduled; using the launch-per-stage loop")
                s.reset_for_rerun()
            else:
                t2v_hip.check_async_errors()
                if stop < s.max_steps:
                    n = stop + 1
                t = s.max_steps
        while t < s.max_steps:
            t1 = min(s.max_steps, t + chunk)
            s.run(t, t1, self.gate_threshold, drop_rate, False, seed)
            stop = int(s.stop.item())
            if stop < t1:
                n = stop + 1
                break
            t = t1
        if n is None:
            print("Warning! Reached max decoder steps")
            n = s.max_steps
        s.t = n
        mel = s.MEL[:n].permute(1, 2, 0).contiguous()           # (B,80,T)
        gate = s.GATE[:n].transpose(0, 1).unsqueeze(-1).contiguous()   # (B,T,1)
        return mel, gate, s.AL[1:n + 1].transpose(0, 1)

    def _tf_rng_t(self):
        """mask index of the teacher-forced Prenet call: set per model forward by Tacotron2._forward (None = the running
        block counter, for a Decoder that is driven directly)"""
        return self.__dict__.pop('_tf_t', None)

    def prepare(self, decoder_inputs):
        """The part of the teacher-forced pass that depends on the target frames only (reference model.py:400-404: go
        frame + Prenet) and the Prenet term of attention_rnn's gates for all steps.  Tacotron2._forward issues it on the
        engine's deferred-work stream BEFORE the text encoder, so it runs next to the encoder instead of behind it;
        forward() picks the result up.  Returns nothing; a forward() without a prepare() computes the same inline."""
        frames = self.parse_decoder_inputs(decoder_inputs)                       # (T,B,80)
        T = frames.size(0)
        x = torch.cat((frames.new_zeros(1, frames.size(1), frames.size(2)), frames[:T - 1]), 0)     # go frame first
        pre = self.prenet(x, self._tf_rng_t())            # (T,B,256)
        att = self.attention_rnn
        gpre = t2v_hip.gemm(pre.detach().reshape(-1, pre.shape[-1]), att.weight_ih.detach()[:, :self.prenet_dim],
                            (att.bias_ih + att.bias_hh).detach())
        self._prepared = (decoder_inputs, pre, gpre, torch.cuda.current_stream(), self._param_operands())

    def _param_operands(self):
        """operands that depend on the parameters only: decoder_rnn's summed bias, linear_projection and gate_layer as one
        81-row matrix (reference model.py:385-388).  With a prepare() they are made on the deferred-work stream while the
        encoder runs (autograd runs their backward there, too) instead of as five launches in front of / behind the decoder."""
        dec = self.decoder_rnn
        return (dec.bias_ih + dec.bias_hh,
                torch.cat((self.linear_projection.weight, self.gate_layer.weight), 0),
                torch.cat((self.linear_projection.bias, self.gate_layer.bias), 0))

    def forward(self, memory, decoder_inputs, memory_lengths):
        """Teacher-forced pass (reference model.py:391-426).  Returns mel (B,80,T), gate (B,T),
        alignments (B,T,T_in)."""
        B, T_in = memory.size(0), memory.size(1)
        prep = self.__dict__.pop('_prepared', None)
        gpre = None
        operands = None
        if prep is not None and prep[0] is decoder_inputs:
            _, pre, gpre, st, operands = prep
            T = pre.size(0)
            if st != torch.cuda.current_stream():
                torch.cuda.current_stream().wait_stream(st)
                for t in operands:
                    t.record_stream(torch.cuda.current_stream())
        else:
            frames = self.parse_decoder_inputs(decoder_inputs)                       # (T,B,80)
            T = frames.size(0)
            x = torch.cat((self.get_go_frame(memory).unsqueeze(0), frames), 0)       # go frame first
            pre = self.prenet(x[:T], self._tf_rng_t())        # (T,B,256); the dropout mask is indexed per element, so dropping the unused last row first changes nothing
        att = self.attention_rnn
        lin = t2v_hip.LinearHIP.apply
        pm = lin(memory, self.attention_layer.memory_layer.weight, None, False, 0.0, 0, 0, 0)
        lengths = lengths_i32(memory_lengths, memory.device)
        training = self.training
        p_att = self.p_attention_dropout if training else 0.0
        p_dec = self.p_decoder_dropout if training else 0.0
        self._calls += 1
        seed = (int(self.dropout_seed) * 1000003 + self._calls) & 0x7FFFFFFFFFFFFFFF
        # the prenet term of attention_rnn's gates (pre · weight_ih[:, :256]^T + b_ih + b_hh for all steps) is computed
        # inside the node, so weight_ih has a single gradient producer
        t2v_hip.stamp('dec_fwd_begin')
        b_dec, w81, b81 = operands if operands is not None else self._param_operands()
        hc, alignments = t2v_hip.DecoderCore.apply(None, memory, pm, lengths, *self._core_weights(b_dec),
                                                   p_att, p_dec, seed, torch.is_grad_enabled(),
                                                   pre, att.bias_ih, att.bias_hh, gpre)
        t2v_hip.stamp('dec_fwd_end')
        # linear_projection and gate_layer as ONE 81-column MFMA tile (reference model.py:385-388)
        out = lin(hc, w81, b81, False, 0.0, 0, 0, 0)                              # (T,B,81)
        mel = out[..., :self.n_mel_channels].permute(1, 2, 0).contiguous()       # (B,80,T)
        gate = out[..., self.n_mel_channels].transpose(0, 1).contiguous()        # (B,T)
        return mel, gate, alignments


class SymbolEmbedding(nn.Embedding):
    """nn.Embedding with the same parameter (`weight`), init and call surface; on the GPU the lookup and its backward
    run on the HIP gather / per-symbol accumulation kernels (csrc/embed.hip)."""

    def forward(self, ids):
        return t2v_hip.SymbolEmbedding.apply(ids, self.weight)       # raises T2VHipError for CPU tensors: no CPU fallback


def lengths_i32(lengths, device):
    """int32 device copy of a length vector: the one BatchLayout uploaded next to it, else a conversion"""
    v = getattr(lengths, '_i32', None)
    if v is not None and v.device == device:
        return v
    return lengths.to(device=device, dtype=torch.int32)


class BatchLayout(object):
    """Byte layout of one collated batch (reference data_utils.py:84-128 TextMelCollate output) inside a single staging
    buffer: (text int64, input_lengths int64, mel f32, gate f32, output_lengths int64, speakers f32, emotions f32),
    every field 16-byte aligned.  `key` identifies the shapes (the training engine keys its captured graphs on it)."""
    _DT = (torch.int64, torch.int64, torch.float32, torch.float32, torch.int64, torch.float32, torch.float32)
    _ring = {}          # nbytes -> [pinned host buffers], [events], next slot
    RING = 4
    n_symbols = None    # set by Tacotron2.__init__: symbol ids are validated here, on the host, like nn.Embedding would

    @classmethod
    def check_ids(cls, text):
        """nn.Embedding (reference model.py:528) raises on an id outside [0, n_symbols); the HIP gather clamps instead, so
        the check happens where the ids are still host memory (a bad text front end must not train silently)"""
        if cls.n_symbols is not None and text.numel() and not text.is_cuda:
            lo, hi = int(text.min()), int(text.max())
            if lo < 0 or hi >= cls.n_symbols:
                raise IndexError("symbol id out of range: got [%d, %d], embedding has %d rows" % (lo, hi, cls.n_symbols))

    def __init__(self, batch):
        self.check_ids(batch[0])
        self.max_len = int(torch.max(batch[1]).item())
        self.fields = []
        off = 0
        for t, dt in zip(batch, self._DT):
            nb = t.numel() * (8 if dt == torch.int64 else 4)
            self.fields.append((off, nb, dt, tuple(t.shape)))
            off += (nb + 15) & ~15
        # (round 4) int32 copies of both length vectors ride along: the kernels take int32 lengths, and converting on the
        # device cost two launches per step on the critical path
        self.i32 = []
        for idx in (1, 4):
            nb = batch[idx].numel() * 4
            self.i32.append((idx, off, nb))
            off += (nb + 15) & ~15
        self.nbytes = max(off, 16)
        self.key = tuple(f[3] for f in self.fields) + (self.max_len,)

    def pack_into(self, host, batch):
        """write the seven tensors (+ the int32 copies of both length vectors) into a uint8 host buffer of `nbytes`"""
        for t, (off, nb, dt, shape) in zip(batch, self.fields):
            if nb:
                host[off:off + nb].view(dt).view(shape).copy_(t)      # dtype conversion happens here, on the host
        for idx, off, nb in self.i32:
            if nb:
                host[off:off + nb].view(torch.int32).copy_(batch[idx].reshape(-1))

    def upload(self, batch, into=None):
        ring = self._ring.get(self.nbytes)
        if ring is None:
            ring = self._ring[self.nbytes] = [[torch.empty(self.nbytes, dtype=torch.uint8).pin_memory()
                                               for _ in range(self.RING)], [None] * self.RING, 0]
        slot = ring[2]
        ring[2] = (slot + 1) % self.RING
        if ring[1][slot] is not None:
            ring[1][slot].synchronize()        # the copy that last read this pinned slot has finished
        host = ring[0][slot]
        self.pack_into(host, batch)
        dev = into if into is not None else torch.empty(self.nbytes, dtype=torch.uint8, device='cuda')
        assert dev.numel() == self.nbytes and dev.dtype == torch.uint8
        dev.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring[1][slot] = ev
        return dev

    def views(self, dev):
        """((text, input_lengths, mel, max_len, output_lengths, speakers, emotions), (mel, gate)) over `dev`"""
        v = [dev[off:off + nb].view(dt).view(shape) if nb else torch.empty(shape, dtype=dt, device=dev.device)
             for off, nb, dt, shape in self.fields]
        for idx, off, nb in self.i32:
            if nb:
                v[idx]._i32 = dev[off:off + nb].view(torch.int32)
        text, input_lengths, mel, gate, output_lengths, speakers, emotions = v
        return ((text, input_lengths, mel, self.max_len, output_lengths, speakers, emotions), (mel, gate))


class Tacotron2(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.mask_padding = hparams.mask_padding
        self.fp16_run = hparams.fp16_run
        _drop_seed[0] = (int(hparams.seed) * 2654435761 + 0x5EED) & 0x7FFFFFFFFFFFFFFF
        self.n_mel_channels = hparams.n_mel_channels
        self.n_frames_per_step = hparams.n_frames_per_step
        self.transcript_embedding = SymbolEmbedding(hparams.n_symbols, hparams.symbols_embedding_dim)
        BatchLayout.n_symbols = int(hparams.n_symbols)
        self.speaker_embedding = LinearNorm(hparams.n_speakers, hparams.speaker_embedding_dim, bias=True,
                                            w_init_gain='tanh')     # constructed, never used (B-7)
        self.emotion_embedding = LinearNorm(hparams.n_emotions, hparams.emotion_embedding_dim, bias=True,
                                            w_init_gain='tanh')     # constructed, never used (B-7)
        val = sqrt(3.0) * sqrt(2.0 / (hparams.n_symbols + hparams.symbols_embedding_dim))
        self.transcript_embedding.weight.data.uniform_(-val, val)
        self.encoder = Encoder(hparams)
        self.decoder = Decoder(hparams)
        self.postnet = Postnet(hparams)
        self.vae_gst = VAE_GST(hparams)

    def parse_batch(self, batch, into=None):
        """reference model.py:486-503.  With a GPU and a host batch the seven tensors travel as ONE pinned staging
        buffer and ONE async H2D copy (dtype conversions done on the host while packing); the returned tensors are
        views of that device buffer.  `into` (a device uint8 buffer of BatchLayout(batch).nbytes) receives the upload
        instead of a fresh allocation — the training engine passes the static input buffer of a captured graph."""
        text, input_lengths, mel, gate, output_lengths, speakers, emotions = batch
        # max_len is read from the HOST copy: `.item()` on the device tensor (reference model.py:496) would make the
        # host wait for everything queued so far — i.e. for the previous iteration's backward — every step
        if torch.cuda.is_available() and not any(t.is_cuda for t in batch):
            lay = BatchLayout(batch)
            dev = lay.upload(batch, into)
            return lay.views(dev)
        max_len = int(torch.max(input_lengths).item())
        BatchLayout.check_ids(text)
        text = to_gpu(text).long()
        speakers, emotions = to_gpu(speakers).float(), to_gpu(emotions).float()
        input_lengths = to_gpu(input_lengths).long()
        mel, gate = to_gpu(mel).float(), to_gpu(gate).float()
        output_lengths = to_gpu(output_lengths).long()
        return ((text, input_lengths, mel, max_len, output_lengths, speakers, emotions), (mel, gate))

    def parse_input(self, inputs):
        return inputs

    overlap_branches = True     # class-level switch (tests flip it to compare against the single-stream schedule)

    def parse_output(self, outputs, output_lengths=None):
        """In-place on .data exactly like reference model.py:509-520 (Appendix B-5: the Postnet's
        first conv therefore sees the zero-masked decoder mel in its weight gradient)."""
        if self.mask_padding and output_lengths is not None:
            mel, mel_post, gate = outputs[0].data, outputs[1].data, outputs[2].data
            if mel.is_cuda and mel.is_contiguous() and mel_post.is_contiguous() and gate.is_contiguous():
                # the three fills as ONE launch straight from the lengths (was arange, lt, not and three masked_fill_)
                t2v_hip.mask_outputs(mel, mel_post, gate, lengths_i32(output_lengths, mel.device))
            else:
                pad = ~get_mask_from_lengths(output_lengths, outputs[0].size(2))
                mel.masked_fill_(pad.unsqueeze(1), 0.0)
                mel_post.masked_fill_(pad.unsqueeze(1), 0.0)
                gate.masked_fill_(pad, 1e3)
        return outputs

    def forward(self, inputs):
        with t2v_hip.defer_bn_counters():
            return self._forward(inputs)

    def _forward(self, inputs):
        text, input_lengths, targets, _, output_lengths, speakers, emotions = self.parse_input(inputs)
        if t2v_hip.step_params(create=False) is not None:
            # a training engine drives the dropout epoch through the device-side step record: the host-side call
            # counters restart with every forward pass, so an iteration issues the same launches whether it runs
            # eagerly or as a replayed graph (and a resumed run continues with the masks of its iteration number)
            _block_calls[0] = 0
            self.decoder._calls = 0
            if not self.training:
                # several eval-mode forwards inside one training iteration (the validation loop; the Prenet drops out in
                # eval mode too, model.py:102-106 of the reference) must not reuse one set of masks: they are
                # never captured, so a host-side sequence number can offset their counters
                self._eval_seq = (getattr(self, '_eval_seq', 0) + 1) % 4096
                _block_calls[0] = self._eval_seq * 64
                self.decoder._calls = self._eval_seq * 64
        # The text encoder and the reference encoder (VAE) are independent until the add below, and both are chains
        # of small latency-bound kernels (persistent BiLSTM on 16 workgroups, GRU, stride-2 convs): run the VAE
        # branch on a side stream so the two chains share the chip.  Autograd replays each node on its forward
        # stream, so the backward passes of the two branches overlap the same way.
        # ... inside a captured graph only (the dependency becomes a graph edge); as eager launches the two cross-stream
        # waits of a step are resolved by the runtime's host threads, which buys nothing over one in-order stream there.
        # Round 4: the schedule belongs to the training engine (t2v_hip.Overlap, explicit side streams; forks and joins
        # become graph edges under capture).  Three chains start here and meet in front of the decoder: Prenet -> gpre
        # (+ the flipped conv weights of this step's backward) on the deferred-work stream, the reference encoder on its
        # own stream, embedding -> conv bank -> BiLSTM on the current one.  Without an engine everything runs inline.
        ov = t2v_hip.overlap()
        if ov is not None:
            ov.on = bool(self.overlap_branches) and targets.is_cuda and os.environ.get('T2V_OVERLAP', '1') != '0'
        self.decoder.__dict__['_tf_t'] = (1 << 20) + _block_calls[0]
        fork = t2v_hip.mark()
        enc_first = os.environ.get('T2V_FWD_ORDER', 'enc_first') == 'enc_first' and fork is not None
        t2v_hip.BiLSTM._prep.clear()        # (nothing of a step that was abandoned half-way may be picked up by this one)
        with t2v_hip.side('w', after=fork) as forked_p:
            if forked_p:    # parameter-only operands of the BiLSTM, ready long before the conv bank in front of it is done
                l = self.encoder.lstm
                t2v_hip.BiLSTM.prepare(l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0, l.weight_hh_l0_reverse, l.bias_ih_l0_reverse,
                                       l.bias_hh_l0_reverse)
        if enc_first:       # the host issues the longest chain first; the side chains still fork from `fork`
            embedded = self.transcript_embedding(text).transpose(1, 2)
            transcript = self.encoder(embedded, input_lengths)
        with t2v_hip.side('vae', keep=(targets,), after=fork) as forked:
            style, mu, logvar, z = self.vae_gst(targets)
            t2v_hip.stamp('vae_fwd_end')
        with t2v_hip.side('w', keep=(targets,), after=fork) as forked_w:
            if forked_w:
                if self.training and torch.is_grad_enabled():
                    t2v_hip.preflip_conv_weights([blk[0].conv.weight for blk in self.encoder.convolutions]
                                                 + [blk[0].conv.weight for blk in self.postnet.convolutions])
                self.decoder.prepare(targets)
                t2v_hip.stamp('prenet_gpre_end')
        if not enc_first:
            embedded = self.transcript_embedding(text).transpose(1, 2)
            transcript = self.encoder(embedded, input_lengths)
        t2v_hip.stamp('encoder_end')
        if forked:
            ov.wait('vae')
            main = torch.cuda.current_stream()
            for t in (style, mu, logvar, z):
                t.record_stream(main)
        memory = transcript + style.unsqueeze(1)
        mel, gate, alignments = self.decoder(memory, targets, memory_lengths=input_lengths)
        mel_post = mel + self.postnet(mel)
        t2v_hip.stamp('postnet_fwd_end')
        return self.parse_output([mel, mel_post, gate, alignments, mu, logvar, z, emotions], output_lengths)
