"""float64 numpy restatement (TEST INFRASTRUCTURE — see oracle/__init__).

Shares no code with torch's FFT/BLAS: frames are cut by hand and transformed with
``numpy.fft``.  Used to cross-check ``torch_ref`` (expected agreement ~2e-7 of max)
and as a second opinion on the HIP kernels.  Citations: ``torchaudio_contrib/functional.py``.
"""
import math

import numpy as np


def hann_periodic(n):
    """torch.hann_window(n) (periodic=True), used at functional.py:93-97."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _pad_rows(x, pad, mode):
    if pad == 0:
        return x
    np_mode = {'reflect': 'reflect', 'constant': 'constant', 'replicate': 'edge',
               'circular': 'wrap'}[mode]
    return np.pad(x, ((0, 0), (pad, pad)), mode=np_mode)


def stft(x, n_fft, hop=None, win_length=None, window=None, center=True,
         pad_mode='reflect', normalized=False, onesided=True):
    """functional.py:48-113 → complex128 array (*, C, F, T)."""
    x = np.asarray(x, dtype=np.float64)
    lead = x.shape[:-1]
    rows = x.reshape(-1, x.shape[-1])
    hop = n_fft // 4 if hop is None else hop
    wl = n_fft if win_length is None else win_length
    w = hann_periodic(wl) if window is None else np.asarray(window, dtype=np.float64)
    full = np.zeros(n_fft)
    off = (n_fft - wl) // 2                      # torch.stft centres a short window
    full[off:off + wl] = w
    if center:
        rows = _pad_rows(rows, n_fft // 2, pad_mode)
    n_frames = 1 + (rows.shape[1] - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = rows[:, idx] * full                 # (rows, T, N)
    spec = np.fft.rfft(frames, axis=-1) if onesided else np.fft.fft(frames, axis=-1)
    if normalized:
        spec = spec / math.sqrt(n_fft)
    spec = np.swapaxes(spec, 1, 2)               # (rows, F, T)
    return spec.reshape(lead + spec.shape[1:])


def hz_to_mel(hz, htk):
    """functional.py:26-45."""
    hz = np.asarray(hz, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + hz / 700.0)
    f_sp = 200.0 / 3
    brk = 1000.0
    step = math.log(6.4) / 27.0
    return np.where(hz >= brk, brk / f_sp + np.log(np.maximum(hz, 1e-300) / brk) / step, hz / f_sp)


def mel_to_hz(mel, htk):
    """functional.py:5-23."""
    mel = np.asarray(mel, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    brk = 1000.0
    step = math.log(6.4) / 27.0
    return np.where(mel >= brk / f_sp, brk * np.exp(step * (mel - brk / f_sp)), f_sp * mel)


def create_mel_filter(num_freqs, num_mels, min_freq, max_freq, htk):
    """functional.py:131-169."""
    bins = np.linspace(min_freq, max_freq, num_freqs)
    edges = mel_to_hz(np.linspace(hz_to_mel(min_freq, htk), hz_to_mel(max_freq, htk),
                                  num_mels + 2), htk)
    width = np.diff(edges)
    d = edges[None, :] - bins[:, None]
    return np.maximum(np.minimum(-d[:, :-2] / width[:-1], d[:, 2:] / width[1:]), 0.0)


def melspectrogram_db(x, n_fft, hop, num_mels, sample_rate, ref=1.0, amin=1e-7,
                      min_freq=0.0, max_freq=None, htk=False, db=True):
    """layers.py:307-381 chain in float64 (power=2 → mel → 10·log10(max(mel², amin)/ref))."""
    p = np.abs(stft(x, n_fft, hop)) ** 2
    fb = create_mel_filter(n_fft // 2 + 1, num_mels, min_freq,
                           max_freq if max_freq else sample_rate // 2, htk)
    mel = np.einsum('...ft,fm->...mt', p, fb)
    if not db:
        return mel
    return 10.0 * (np.log10(np.maximum(mel ** 2, amin)) - math.log10(ref))
