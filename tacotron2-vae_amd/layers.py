"""Parameter holders with the reference's Xavier initialisation (reference layers.py:7-36)
and the mel front end class (reference layers.py:54-92) backed by the HIP STFT→mel kernel."""
import torch
from torch import nn
from torch.nn import functional as F


def _xavier(weight, gain_name):
    nn.init.xavier_uniform_(weight, gain=nn.init.calculate_gain(gain_name))


class LinearNorm(nn.Module):
    """state_dict keys `linear_layer.{weight,bias}` as in the reference."""

    def __init__(self, in_dim, out_dim, bias=True, w_init_gain='linear'):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        _xavier(self.linear_layer.weight, w_init_gain)

    @property
    def weight(self):
        return self.linear_layer.weight

    @property
    def bias(self):
        return self.linear_layer.bias

    def forward(self, x):
        return F.linear(x, self.linear_layer.weight, self.linear_layer.bias)


class ConvNorm(nn.Module):
    """state_dict keys `conv.{weight,bias}` as in the reference."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None,
                 dilation=1, bias=True, w_init_gain='linear'):
        super().__init__()
        if padding is None:
            if kernel_size % 2 != 1:
                raise ValueError("ConvNorm needs an odd kernel when padding is implicit")
            padding = dilation * (kernel_size - 1) // 2
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, dilation=dilation, bias=bias)
        _xavier(self.conv.weight, w_init_gain)

    def forward(self, signal):
        return self.conv(signal)
