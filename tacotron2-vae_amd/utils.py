"""Small helpers on the path (reference utils.py:9-32)."""
import numpy as np
import torch

max_wav_value = 32768.0


def get_mask_from_lengths(lengths, max_len=None):
    """True where position < length (reference utils.py:9-13 returns a byte mask on
    `torch.cuda.LongTensor`; a bool mask on the lengths' device is the modern equivalent)."""
    if max_len is None:
        max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device, dtype=lengths.dtype)
    return ids.unsqueeze(0) < lengths.unsqueeze(1)


def load_wav_to_torch(full_path):
    """int16 PCM wav -> (float32 tensor of raw sample values, sampling rate) (utils.py:16-18)."""
    from scipy.io.wavfile import read
    sampling_rate, data = read(full_path)
    return torch.from_numpy(data.astype(np.float32)), sampling_rate


def load_filepaths_and_text(filename, split="|"):
    with open(filename, encoding='utf-8') as f:
        return [line.strip().split(split) for line in f]


def to_gpu(x):
    """utils.py:27-32: contiguous + async H2D when a GPU exists."""
    x = x.contiguous()
    if torch.cuda.is_available():
        x = x.cuda(non_blocking=True)
    return x
