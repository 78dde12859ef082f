"""float64 tensors on a gfx950 device: launchers of the ``tac_*_f64`` entry points (``csrc/chain_f64.hip``).

The reference keeps float64 in, float64 out through every function of the path (``functional.py:48-113`` stft,
``:116-128`` complex_norm, ``:172-184`` apply_filterbank, ``:187-201`` angle / magphase, ``:277-314`` the dB pair).  The
measured hot path is float32 (``_hip.py``); this module is its float64 counterpart with the same physical layouts
(frame-major buffers returned as the logical ``(…, freq, time[, 2])`` views) so the two are interchangeable behind
``_ops.py``.  Gradients of float64 calls are taken by differentiating the stock-torch evaluation (announced as such).
"""
import torch

from . import _hip as H
from . import _native

_F64 = torch.float64


#: longest fft_length handed to the O(N^2) direct float64 transform of chain_f64.hip (odd lengths, halves with a prime
#: factor above 5).  It costs 2 N^2 multiply-adds per frame in one lane's scalar loop — 0.5 M at 512, 33 M at 4094 — so above
#: this the announced stock-torch route (hipFFT) is the faster one and is taken instead.
DIRECT_MAX = 512


def _half_is_5_smooth(n_fft):
    if n_fft % 2:
        return False
    m = n_fft // 2
    for r in (2, 3, 5):
        while m % r == 0:
            m //= r
    return m == 1


def covers(n_fft):
    """Even lengths <= 8192 whose half is 5-smooth (LDS Stockham transform, radix 4 / 2 / 3 / 5 passes), or any other length
    <= ``DIRECT_MAX`` (direct transform) — ``plan_f64`` / ``geometry_f64`` in csrc/chain_f64.hip accept more (any length
    <= 4096 through the direct kernel), the Python layer does not route long non-smooth sizes there."""
    if n_fft < 1 or n_fft > 8192:
        return False
    return _half_is_5_smooth(n_fft) or n_fft <= DIRECT_MAX


def all_f64(*tensors):
    return all(t.dtype == _F64 for t in tensors)


def _desc(g):
    return g.desc if g.desc is not None else _native.StftDesc(
        rows=g.rows, length=g.length, row_stride=g.row_stride, n_fft=g.n_fft, hop=g.hop, win_length=g.win_length,
        center=1 if g.center else 0, pad_mode=_native.PAD_MODES[g.pad_mode], normalized=1 if g.normalized else 0,
        onesided=1 if g.onesided else 0, reserved=0)


def _launch(name, dev, *args):
    with _native.on_device(dev):
        rc = getattr(_native.lib(), name)(*args, _native.stream_ptr(dev))
    _native.check(rc, name)
    H._count(name)


def stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
    g = H.geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    src = H._rows_of(wave, g)
    window = window.contiguous()
    out = torch.empty(g.stft_shape, dtype=_F64, device=wave.device)
    _launch('tac_stft_f64', wave.device, _native.ptr(src), _native.ptr(window), _desc(g), _native.ptr(out))
    return out.transpose(-3, -2)


def _spectrogram_rows(wave, window, g, power, db, ref, amin):
    src = H._rows_of(wave, g)
    window = window.contiguous()
    out = torch.empty(g.spec_shape, dtype=_F64, device=wave.device)
    _launch('tac_spectrogram_f64', wave.device, _native.ptr(src), _native.ptr(window), _desc(g), float(power),
            1 if db else 0, float(ref), float(amin), _native.ptr(out))
    return out                                                            # physical (…, T, F)


def spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref, amin):
    g = H.geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    return _spectrogram_rows(wave, window, g, power, db, ref, amin).transpose(-2, -1)


def apply_filterbank(spec, fb, db=None):
    """``spec``: logical (…, F, T) in any strides the (row, freq, frame) addressing covers; the result is frame-major."""
    fb = fb.contiguous()
    n_freqs, n_frames = spec.shape[-2], spec.shape[-1]
    lead = tuple(spec.shape[:-2])
    n_mels = fb.shape[1]
    out = torch.empty(lead + (n_frames, n_mels), dtype=_F64, device=spec.device)
    if out.numel():
        rows = spec.reshape(-1, n_freqs, n_frames)                       # a view for both layouts the ops here produce
        step = 65535                                                      # grid.z
        for r0 in range(0, rows.shape[0], step):
            part = rows[r0:r0 + step]
            _launch('tac_apply_filterbank_f64', spec.device, _native.ptr(part), part.shape[0], n_freqs, n_frames,
                    part.stride(0), part.stride(1), part.stride(2), _native.ptr(fb), n_mels, 0 if db is None else 1,
                    1.0 if db is None else float(db[0]), 0.0 if db is None else float(db[1]),
                    _native.ptr(out.reshape(-1, n_frames, n_mels)[r0:r0 + step]))
    return out.transpose(-2, -1)


def melspectrogram(wave, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref,
                   amin):
    """Two launches: |X|^p rows (frame-major, never transposed), then the contraction with the dB epilogue."""
    g = H.geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    rows = _spectrogram_rows(wave, window, g, power, False, 1.0, 0.0)
    return apply_filterbank(rows.transpose(-2, -1), fb, (ref, amin) if db else None)


def _pair_call(z, power, want_mag, want_phase):
    z = H._pairs(z)
    shape, strides = z.shape[:-1], tuple(s // 2 for s in z.stride()[:-1])
    mag = torch.empty_strided(shape, strides, dtype=_F64, device=z.device) if want_mag else None
    phase = torch.empty_strided(shape, strides, dtype=_F64, device=z.device) if want_phase else None
    n = (mag if want_mag else phase).numel()
    if n:
        _launch('tac_magphase_f64', z.device, _native.ptr(z), n, float(power), None if mag is None else _native.ptr(mag),
                None if phase is None else _native.ptr(phase))
    return mag, phase


def complex_norm(z, power):
    return _pair_call(z, power, True, False)[0]


def angle(z):
    return _pair_call(z, 1.0, False, True)[1]


def magphase(z, power):
    return _pair_call(z, power, True, True)


def _unary(x, name, *params):
    x = x if H.is_dense(x) else x.contiguous()
    out = torch.empty_like(x)
    if x.numel():
        _launch(name, x.device, _native.ptr(x), x.numel(), *params, _native.ptr(out))
    return out


def amplitude_to_db(x, ref, amin):
    return _unary(x, 'tac_amplitude_to_db_f64', float(ref), float(amin))


def db_to_amplitude(x, ref):
    return _unary(x, 'tac_db_to_amplitude_f64', float(ref))
