"""k_mel_frontend on a chip-filling batch (128 utterances x 102 144 samples = 51 328 frames), ten launches: the workload of the
counter passes behind "bound by its butterflies, not by HBM" (tools/r06_pmc_py.sh with PMC_FILTER=k_mel_frontend)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, layers, t2v_hip
B, n_samples = 128, 102144
stft = layers.TacotronSTFT(1024, 256, 1024, 80, 16000, 0.0, 8000.0)
g = torch.Generator().manual_seed(0)
wav = (torch.clamp(0.1 * torch.randn(B, n_samples, generator=g), -1, 1) * 32767).to(torch.int16).cuda()
n = torch.full((B,), n_samples, dtype=torch.int64)
tables = stft._tables(wav.device)
for _ in range(10):
    mel = t2v_hip.mel_frontend(wav, n, tables, scale=1.0 / 32768.0, t_stride=n_samples // 256 + 1)
torch.cuda.synchronize()
print('ok', tuple(mel.shape))
