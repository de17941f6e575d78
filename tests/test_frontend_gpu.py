"""STFT→mel HIP kernel vs the reference's golden mels and vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def stft():
    import layers
    return layers.TacotronSTFT(1024, 256, 1024, 80, 16000, 0.0, 8000.0)


def test_matches_reference_golden(stft, golden_dir):
    g = np.load(os.path.join(golden_dir, 'mel_frontend.npz'))
    assert (stft.mel_basis - torch.from_numpy(g['mel_basis'])).abs().max() < 1e-7
    for name in ('speech', 'noise'):
        wav = torch.from_numpy(g[name + '_wav'])
        ref = torch.from_numpy(g[name + '_mel'])
        mel = stft.mel_spectrogram((wav.float() / 32768.0)[None])[0]           # float path (reference call)
        assert mel.shape == ref.shape
        d = (mel - ref).abs()
        assert d.mean() < 1e-5 and d.max() < 2e-3, (name, d.mean().item(), d.max().item())
        mel16 = stft.mel_spectrogram(wav[None].cuda(), scale=1.0 / 32768.0)[0].cpu()   # int16 on-device path
        assert (mel16 - mel).abs().max() < 1e-5


def test_batched_ragged_matches_oracle(stft):
    import t2v_oracle as O
    g = torch.Generator().manual_seed(0)
    lens = [102144, 48000, 32001, 1537, 70000, 1025]
    N = max(lens)
    wav = torch.zeros(len(lens), N)
    for i, n in enumerate(lens):
        wav[i, :n] = torch.clamp(0.1 * torch.randn(n, generator=g), -1, 1)
    mel = stft.mel_spectrogram(wav.cuda(), lengths=torch.tensor(lens)).cpu()
    assert mel.shape == (len(lens), 80, N // 256 + 1)
    for i, n in enumerate(lens):
        T = n // 256 + 1
        ref = O.mel_spectrogram(wav[i:i + 1, :n])[0]
        assert ref.shape[1] == T
        d = (mel[i, :, :T] - ref).abs()
        assert d.mean() < 1e-5 and d.max() < 2e-3, (i, d.mean().item(), d.max().item())
        assert float(mel[i, :, T:].abs().max()) == 0.0 if T < mel.shape[2] else True     # zero fill past T_i


def test_linearity_and_silence(stft):
    """Size-independent properties at full length: silence → log(1e-5) everywhere; scaling the
    waveform by c shifts the un-clamped log-mel by log(c)."""
    n = 102144
    z = stft.mel_spectrogram(torch.zeros(1, n).cuda())
    assert torch.allclose(z, torch.full_like(z, float(np.log(1e-5))))
    g = torch.Generator().manual_seed(1)
    x = torch.clamp(0.2 * torch.randn(1, n, generator=g), -1, 1).cuda()
    a, b = stft.mel_spectrogram(x), stft.mel_spectrogram(0.5 * x)
    assert (a - b - float(np.log(2.0))).abs().max() < 1e-4
