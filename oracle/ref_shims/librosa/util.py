import numpy as np


def pad_center(data, size, axis=-1, **kwargs):
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths, mode='constant')


def tiny(x):
    x = np.asarray(x)
    dt = x.dtype if np.issubdtype(x.dtype, np.floating) else np.float32
    return np.finfo(dt).tiny


def normalize(S, norm=None, **kw):
    assert norm is None
    return S
