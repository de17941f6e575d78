"""Parameter holders with the reference's Xavier initialisation (reference layers.py:7-36)
and the mel front end class (reference layers.py:54-92) backed by the HIP STFT→mel kernel."""
import torch
from torch import nn
from torch.nn import functional as F


def _xavier(weight, gain_name):
    nn.init.xavier_uniform_(weight, gain=nn.init.calculate_gain(gain_name))


def hip_linear(x, weight, bias):
    """x (..., in) @ weight^T + bias on the own fp32 MFMA GEMM (t2v_hip.LinearHIP, differentiable); CPU tensors are
    refused — there is no stock-library fallback on this path."""
    import t2v_hip
    if not x.is_cuda:
        raise t2v_hip.T2VHipError("linear layer called with a CPU tensor; this path has no CPU fallback")
    lead = x.shape[:-1]
    y = t2v_hip.LinearHIP.apply(x.reshape(-1, x.shape[-1]).float(), weight, bias, False, 0.0, 0, 0, 0)
    return y.view(*lead, weight.shape[0])


class HipLinear(nn.Linear):
    """nn.Linear parameters / state_dict keys, forward on the HIP GEMM (the VAE head: fc1 / fc2 / fc3)."""

    def forward(self, x):
        return hip_linear(x, self.weight, self.bias)


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
        return hip_linear(x, self.linear_layer.weight, self.linear_layer.bias)


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
        """ConvNorm is a parameter holder on this path: Encoder / Postnet feed `conv.weight` to the fused HIP
        Conv1d + BatchNorm + activation kernels, LocationLayer's filters are folded into the attention kernels.  A
        direct call would be a stock-library convolution standing in for the HIP path, so it is refused loudly."""
        import t2v_hip
        raise t2v_hip.T2VHipError("ConvNorm.forward is not a product path: call the owning module (Encoder / Postnet / "
                                  "Attention), which runs the fused HIP kernels on conv.weight")


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa 0.6.0 `filters.mel(htk=False, norm=1)` (the call at reference layers.py:62-63) in float64."""
    import numpy as np

    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), f / (200.0 / 3))

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))

    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]


class TacotronSTFT(nn.Module):
    """mel front end with the reference's constructor/`mel_spectrogram` surface (layers.py:54-92); the
    transform itself is the HIP kernel k_mel_frontend (hand-written 1024-point FFT per wavefront)."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80,
                 sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0):
        super().__init__()
        import numpy as np
        if (filter_length, hop_length, win_length, n_mel_channels) != (1024, 256, 1024, 80):
            raise NotImplementedError("HIP front end is built for n_fft=win=1024, hop=256, 80 mels")
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.filter_length, self.hop_length = filter_length, hop_length
        basis = slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        self.register_buffer('mel_basis', torch.from_numpy(basis).float())
        n = np.arange(filter_length)
        window = (0.5 - 0.5 * np.cos(2 * np.pi * n / filter_length)).astype(np.float32)   # scipy hann, fftbins=True
        k = np.arange(512)
        tw512 = np.stack((np.cos(2 * np.pi * k / 512), -np.sin(2 * np.pi * k / 512)), 1).astype(np.float32)
        k = np.arange(513)
        tw1024 = np.stack((np.cos(2 * np.pi * k / 1024), -np.sin(2 * np.pi * k / 1024)), 1).astype(np.float32)
        b32 = basis.astype(np.float32)
        nz = [np.nonzero(row)[0] for row in b32]
        start = np.array([int(i[0]) if len(i) else 0 for i in nz], dtype=np.int32)
        length = np.array([int(i[-1] - i[0] + 1) if len(i) else 0 for i in nz], dtype=np.int32)
        maxw = int(length.max())
        rows = np.zeros((n_mel_channels, maxw), dtype=np.float32)
        for m in range(n_mel_channels):
            rows[m, :length[m]] = b32[m, start[m]:start[m] + length[m]]
        self._host_tables = dict(window=window, tw512=tw512, tw1024=tw1024, mel_start=start, mel_len=length,
                                 mel_w=rows)
        self._maxw = maxw
        self._dev_tables = {}

    def _tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            t = {k: torch.from_numpy(v).to(device) for k, v in self._host_tables.items()}
            t['maxw'] = self._maxw
            self._dev_tables[key] = t
        return self._dev_tables[key]

    def mel_spectrogram(self, y, lengths=None, scale=1.0):
        """y: (B,N) in [-1,1] (float32) — or int16 PCM with scale=1/max_wav_value.  CPU input is
        copied to the GPU (the reference computes this on CPU in the DataLoader worker); the result
        lives on the device `y` came from.  lengths: optional per-utterance sample counts."""
        import t2v_hip
        if not torch.cuda.is_available():
            raise t2v_hip.T2VHipError("TacotronSTFT needs a GPU: the mel front end is HIP-only")
        src = y.device
        yd = y if y.is_cuda else y.cuda()
        if yd.dtype != torch.int16:
            yd = yd.float()
            assert torch.min(yd) >= -1 and torch.max(yd) <= 1     # reference layers.py:85-86
        n = torch.full((yd.size(0),), yd.size(1), dtype=torch.int64) if lengths is None else lengths
        mel = t2v_hip.mel_frontend(yd, n, self._tables(yd.device), scale)
        return mel if src.type == 'cuda' else mel.to(src)
