"""Tacotron2Loss_VAE (reference loss_function.py:6-44): 2×MSE + BCE-with-logits + w(step)·KL."""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


def kl_anneal_weight(kind, step, lag, k, x0, upper):
    if kind == 'logistic':
        return float(upper / (upper + np.exp(-k * (step - x0))))   # np.exp like the reference (1-ulp exact)
    if kind == 'linear':
        return min(upper, step / x0) if step > lag else 0
    if kind == 'constant':
        return 0.001
    return None   # reference falls through to None for unknown names


class Tacotron2Loss_VAE(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.anneal_function = hparams.anneal_function
        self.lag, self.k = hparams.anneal_lag, hparams.anneal_k
        self.x0, self.upper = hparams.anneal_x0, hparams.anneal_upper

    def kl_anneal_function(self, anneal_function, lag, step, k, x0, upper):
        return kl_anneal_weight(anneal_function, step, lag, k, x0, upper)

    def forward(self, model_output, targets, step):
        mel_out, mel_post, gate_out, _, mu, logvar = model_output[:6]
        w = kl_anneal_weight(self.anneal_function, step, self.lag, self.k, self.x0, self.upper)
        if mel_out.is_cuda:      # fused value+gradient kernel
            import t2v_hip
            total, out4 = t2v_hip.VAELoss.apply(mel_out, mel_post, gate_out, mu, logvar, targets[0].detach(),
                                                targets[1].detach(), w)
            return total, out4[1], out4[2], w
        mel_target, gate_target = targets[0].detach(), targets[1].detach().reshape(-1, 1)
        recon = (F.mse_loss(mel_out, mel_target) + F.mse_loss(mel_post, mel_target)
                 + F.binary_cross_entropy_with_logits(gate_out.reshape(-1, 1), gate_target))
        kl = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())   # SUM over batch and latent dims
        w = kl_anneal_weight(self.anneal_function, step, self.lag, self.k, self.x0, self.upper)
        return recon + w * kl, recon, kl, w
