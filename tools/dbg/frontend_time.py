"""k_mel_frontend alone: us per launch at B = 6 (one training batch) and B = 128 (chip-filling), events around 20 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, layers, t2v_hip
stft = layers.TacotronSTFT(1024, 256, 1024, 80, 16000, 0.0, 8000.0)
for B in (6, 128):
    n_samples = 102144
    g = torch.Generator().manual_seed(0)
    wav = (torch.clamp(0.1 * torch.randn(B, n_samples, generator=g), -1, 1) * 32767).to(torch.int16).cuda()
    n = torch.full((B,), n_samples, dtype=torch.int64)
    tables = stft._tables(wav.device)
    for _ in range(3):
        mel = t2v_hip.mel_frontend(wav, n, tables, scale=1.0 / 32768.0, t_stride=n_samples // 256 + 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mel = t2v_hip.mel_frontend(wav, n, tables, scale=1.0 / 32768.0, t_stride=n_samples // 256 + 1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    frames = B * (n_samples // 256 + 1)
    print('B=%3d: %.1f us per launch, %.2f M frames/s, checksum %.6f' % (B, us, frames / us, float(mel.double().sum())))
