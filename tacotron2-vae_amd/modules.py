"""VAE / GST reference encoder (reference modules.py:8-85)."""
import torch
from torch import nn
from torch.nn import functional as F

from CoordConv import CoordConv2d
from layers import HipLinear


class ReferenceEncoder(nn.Module):
    """mel (N,80,T) → last GRU state (N, E/2).  modules.py:67 reinterprets the (80,T) block as
    (T,80) rows without a transpose; we keep that."""

    def __init__(self, hparams):
        super().__init__()
        filters = [1] + list(hparams.ref_enc_filters)
        K = len(hparams.ref_enc_filters)
        convs = [CoordConv2d(filters[0], filters[1], kernel_size=(3, 3), stride=(2, 2), padding=(1, 1),
                             with_r=True)]
        convs += [nn.Conv2d(filters[i], filters[i + 1], kernel_size=(3, 3), stride=(2, 2), padding=(1, 1))
                  for i in range(1, K)]
        self.convs = nn.ModuleList(convs)
        self.bns = nn.ModuleList([nn.BatchNorm2d(hparams.ref_enc_filters[i]) for i in range(K)])
        width = self.calculate_channels(hparams.n_mel_channels, 3, 2, 1, K)
        self.gru = nn.GRU(input_size=hparams.ref_enc_filters[-1] * width, hidden_size=hparams.E // 2,
                          batch_first=True)
        self.n_mels = hparams.n_mel_channels

    @staticmethod
    def calculate_channels(L, kernel_size, stride, pad, n_convs):
        for _ in range(n_convs):
            L = (L - kernel_size + 2 * pad) // stride + 1
        return L

    def forward(self, inputs):
        import t2v_hip
        n = inputs.size(0)
        out = inputs.contiguous().view(n, 1, -1, self.n_mels)      # raw reinterpretation, no transpose (B-1)
        for i, (conv, bn) in enumerate(zip(self.convs, self.bns)):
            c = conv.conv if i == 0 else conv                       # layer 0: the live CoordConv inner conv (B-2)
            if self.training:
                t2v_hip.note_bn_counter(bn.num_batches_tracked)
            out = t2v_hip.Conv2dBNReLU.apply(out, c.weight, c.bias, bn.weight, bn.bias, bn.running_mean,
                                             bn.running_var, self.training, i == 0)
        t2v_hip.flush_bn_counters()
        out = out.transpose(1, 2)
        out = out.contiguous().view(n, out.size(1), -1)
        g = self.gru
        return t2v_hip.GRULast.apply(out, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)


class VAE_GST(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.ref_encoder = ReferenceEncoder(hparams)
        # HipLinear: nn.Linear's parameters / init / state_dict keys; a direct `model.vae_gst.fc3(z)` (reference
        # synthesizer.py:131, README inference snippet) runs the own GEMM like the fused forward below
        self.fc1 = HipLinear(hparams.ref_enc_gru_size, hparams.z_latent_dim)
        self.fc2 = HipLinear(hparams.ref_enc_gru_size, hparams.z_latent_dim)
        self.fc3 = HipLinear(hparams.z_latent_dim, hparams.E)
        self.eps_override = None   # test hook: inject the reparameterisation noise

    def reparameterize(self, mu, logvar):
        if not self.training:
            return mu
        eps = self.eps_override if self.eps_override is not None else torch.randn_like(mu)
        import t2v_hip
        return t2v_hip.Reparam.apply(eps, mu, logvar)      # eps * exp(0.5 logvar) + mu, one launch (raises for CPU tensors)

    def forward(self, inputs):
        import t2v_hip
        lin = t2v_hip.LinearHIP.apply
        enc_out = self.ref_encoder(inputs)
        mu = lin(enc_out, self.fc1.weight, self.fc1.bias, False, 0.0, 0, 0, 0)
        logvar = lin(enc_out, self.fc2.weight, self.fc2.bias, False, 0.0, 0, 0, 0)
        z = self.reparameterize(mu, logvar)
        return lin(z, self.fc3.weight, self.fc3.bias, False, 0.0, 0, 0, 0), mu, logvar, z
