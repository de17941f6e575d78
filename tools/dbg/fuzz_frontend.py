"""Random ragged waveform batches through the on-device STFT -> mel front end against the CPU oracle (the body of
tests/test_frontend_gpu.py::test_batched_ragged_matches_oracle with random lengths incl. the shortest legal ones):
`python tools/dbg/fuzz_frontend.py [seed]` from the repo root."""
import os, sys, random
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
sys.path.insert(0, os.path.join(os.getcwd(), 'oracle'))
import torch
import layers
import t2v_oracle as O
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
stft = layers.TacotronSTFT(1024, 256, 1024, 80, 16000, 0.0, 8000.0)
bad = 0
for it in range(8):
    nb = rng.choice([1, 2, 5, 9])
    lens = [rng.choice([513, 600, 1024, 1025, 1537, 4096, 20000, 48000, 70001]) for _ in range(nb)]
    g = torch.Generator().manual_seed(it)
    N = max(lens)
    wav = torch.zeros(nb, N)
    for i, n in enumerate(lens):
        wav[i, :n] = torch.clamp(0.1 * torch.randn(n, generator=g), -1, 1)
    try:
        mel = stft.mel_spectrogram(wav.cuda(), lengths=torch.tensor(lens)).cpu()
        assert mel.shape == (nb, 80, N // 256 + 1)
        for i, n in enumerate(lens):
            T = n // 256 + 1
            ref = O.mel_spectrogram(wav[i:i + 1, :n])[0]
            d = (mel[i, :, :T] - ref).abs()
            assert ref.shape[1] == T and d.mean() < 1e-5 and d.max() < 2e-3, (i, n, d.mean().item(), d.max().item())
            assert T >= mel.shape[2] or float(mel[i, :, T:].abs().max()) == 0.0
        print("ok  ", lens, flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", lens, repr(e)[:200], flush=True)
print("frontend fuzz failures:", bad)
