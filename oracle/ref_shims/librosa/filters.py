"""librosa 0.6.0 `filters.mel` (Slaney scale, norm=1) restated in numpy."""
import numpy as np


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = f >= min_log_hz
    out = np.array(mels, dtype=np.float64, ndmin=1)
    ff = np.array(f, dtype=np.float64, ndmin=1)
    out[np.atleast_1d(big)] = min_log_mel + np.log(ff[np.atleast_1d(big)] / min_log_hz) / logstep
    return out if np.ndim(f) else out[0]


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    freqs = np.array(freqs, ndmin=1)
    mm = np.array(m, ndmin=1)
    freqs[big] = min_log_hz * np.exp(logstep * (mm[big] - min_log_mel))
    return freqs


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm=1):
    assert not htk
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)))
    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == 1:
        enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, np.newaxis]
    return weights
