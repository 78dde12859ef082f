"""Functional API — same names, argument order, defaults and error behaviour as the reference's
``torchaudio_contrib/functional.py``.  Every function resolves its defaults and calls ONE PyTorch custom op of
the ``tac_amd`` namespace (``_ops.py``); for float32 tensors on a HIP device that op is a hand-written gfx950
kernel reached through the C ABI of ``include/tac_amd.h``.

Where the work runs (decided by the dispatcher, like for any torch op):
  * HIP device, float32 (float16 / bfloat16 widened)  → the gfx950 kernels; a missing ``libtac_amd.so`` raises.
  * CPU tensors (the reference's own test-suite, BASELINE configs[0]) → torch's CPU operators in the reference's
    operator order (``_composite.py``), so reference call sites written for the CPU keep working unchanged.
  * HIP device, float64 (the reference keeps f64 → f64) → the float64 kernels of the STFT chain and the phase vocoder
    (``_hip64.py``); float64 mu-law and HPSS → torch's operators on the device, announced.

Outputs are fresh tensors; the STFT-family results are returned as the same strided views the reference
produces (physically frame-major, logically ``(*, channel, freq, time[, 2])``).
"""
import math

import torch

from . import _hip
from . import _ops
from ._lazy import realize as _realize

__all__ = ['stft', 'complex_norm', 'create_mel_filter', 'apply_filterbank', 'angle', 'magphase',
           'phase_vocoder', 'amplitude_to_db', 'db_to_amplitude', 'mu_law_encoding', 'mu_law_decoding', 'hpss']

_call = _ops.call


def _tensor(x, what):
    if not torch.is_tensor(x):
        raise TypeError('%s must be a torch.Tensor, got %s' % (what, type(x).__name__))
    return _realize(x)


_window_cache = {}


def default_window(n, like):
    """periodic Hann of ``win_length`` — the constructor the reference calls (functional.py:93-97,
    layers.py:76-80), cached per (length, device, dtype) for ordinary tensors (never while tracing)."""
    dtype = like.dtype if like.dtype in (torch.float32, torch.float64) else torch.float32
    if type(like) is not torch.Tensor or torch.compiler.is_compiling():
        return torch.hann_window(n, dtype=dtype, device=like.device)
    key = (n, str(like.device), dtype)
    w = _window_cache.get(key)
    if w is None:
        # built outside inference mode whatever the caller's mode: an inference tensor in the cache could not be saved for
        # a later call's backward ("Inference tensors cannot be saved for backward")
        with torch.inference_mode(False):
            w = torch.hann_window(n, dtype=dtype, device=like.device)
        if len(_window_cache) > 64:
            _window_cache.clear()
        _window_cache[key] = w
    return w


def resolve_stft_args(waveforms, fft_length, hop_length, win_length, window):
    """Defaults of reference functional.py:48-49 / :86-97 and the window checks ``torch.stft`` performs."""
    n_fft = int(fft_length)
    hop = n_fft // 4 if hop_length is None else int(hop_length)
    win_length = n_fft if win_length is None else int(win_length)
    if window is None:
        if not 0 < win_length <= max(n_fft, 1):
            raise RuntimeError('stft: expected 0 < win_length <= n_fft, got win_length=%d' % win_length)
        window = default_window(win_length, waveforms)
    else:
        if not torch.is_tensor(window) or window.dim() != 1 or window.shape[0] != win_length:
            raise RuntimeError('stft: expected a 1D window tensor of size equal to win_length=%d' % win_length)
        if window.device != waveforms.device:
            raise RuntimeError('stft: input and window must be on the same device, got %s and %s'
                               % (waveforms.device, window.device))
    return n_fft, hop, win_length, window


# ----------------------------------------------------------------------------- public API
def stft(waveforms, fft_length, hop_length=None, win_length=None, window=None,
         center=True, pad_mode='reflect', normalized=False, onesided=True):
    """Short-time Fourier transform of ``(*, channel, time)`` waveforms →
    ``(*, channel, num_freqs, time, complex=2)``  (reference: functional.py:48-113).

    ``window=None`` means a periodic Hann window of ``win_length or fft_length`` (unlike torch.stft).
    On a HIP device framing, padding (``center``/``pad_mode``), windowing and the R2C FFT run in one gfx950 kernel.
    """
    x = _tensor(waveforms, 'waveforms')
    n_fft, hop, win_length, window = resolve_stft_args(x, fft_length, hop_length, win_length, window)
    _hip.check_stft_args(x.shape, n_fft, hop, win_length, center, pad_mode)
    return _call('stft', x, window, n_fft, hop, win_length, bool(center), pad_mode, bool(normalized), bool(onesided))


def complex_norm(complex_tensor, power=1.0):
    """``|z|**power`` over a trailing ``complex=2`` dim (reference: functional.py:116-128)."""
    z = _tensor(complex_tensor, 'complex_tensor')
    _check_pairs(z, 'complex_norm')
    return _call('complex_norm', z, float(power))


def _hz_to_mel(hz, htk):
    hz = torch.as_tensor(hz).to(torch.get_default_dtype())
    if htk:
        return 2595. * torch.log10(torch.tensor(1., dtype=torch.get_default_dtype()) + hz / 700.)
    f_sp = 200.0 / 3
    knee_hz = 1000.0
    knee_mel = (knee_hz - 0.0) / f_sp
    step = math.log(6.4) / 27.0
    return torch.where(hz >= knee_hz, knee_mel + torch.log(hz / knee_hz) / step, (hz - 0.0) / f_sp)


def _mel_to_hz(mel, htk):
    mel = torch.as_tensor(mel).to(torch.get_default_dtype())
    if htk:
        return 700. * (10 ** (mel / 2595.) - 1.)
    f_sp = 200.0 / 3
    knee_hz = 1000.0
    knee_mel = (knee_hz - 0.0) / f_sp
    step = math.log(6.4) / 27.0
    return torch.where(mel >= knee_mel, knee_hz * torch.exp(step * (mel - knee_mel)), 0.0 + f_sp * mel)


# the reference's (private) names of the two conversions, functional.py:5-45: call sites that import them keep working
_hertz_to_mel = _hz_to_mel
_mel_to_hertz = _mel_to_hz


def create_mel_filter(num_freqs, num_mels, min_freq, max_freq, htk):
    """Dense ``(num_freqs, num_mels)`` triangular mel filterbank, Slaney (default) or HTK scale, no
    area normalisation, bin grid ``linspace(min_freq, max_freq, num_freqs)`` (reference:
    functional.py:131-169).  One-off init-time constant: evaluated with the same float32 host
    arithmetic as the reference so the matrix is bit-identical; move it with ``.cuda()``."""
    grid = torch.linspace(min_freq, max_freq, num_freqs)
    knots = _mel_to_hz(torch.linspace(_hz_to_mel(min_freq, htk), _hz_to_mel(max_freq, htk), num_mels + 2), htk)
    gap = knots[1:] - knots[:-1]
    delta = knots.unsqueeze(0) - grid.unsqueeze(1)
    lower = (-1. * delta[:, :-2]) / gap[:-1]
    upper = delta[:, 2:] / gap[1:]
    return torch.clamp(torch.min(lower, upper), min=0.)


def apply_filterbank(mag_specgrams, filterbank):
    """``(…, num_freqs, time) x (num_freqs, num_bands) → (…, num_bands, time)`` (reference:
    functional.py:172-184); on a HIP device a band-sparse streaming contraction for triangular banks, the fp32
    matrix cores for dense ones."""
    spec = _tensor(mag_specgrams, 'mag_specgrams')
    fb = _tensor(filterbank, 'filterbank')
    if fb.dim() != 2 or spec.dim() < 2 or spec.shape[-2] != fb.shape[0]:
        raise RuntimeError('apply_filterbank: size mismatch, spectrogram %s vs filterbank %s'
                           % (tuple(spec.shape), tuple(fb.shape)))
    if fb.device != spec.device:
        raise RuntimeError('apply_filterbank: spectrogram and filterbank must be on the same device')
    return _call('apply_filterbank', spec, fb)


def _check_pairs(z, what):
    if z.dim() < 1 or z.shape[-1] != 2:
        raise RuntimeError('%s: expected a trailing dimension of size 2, got shape %s' % (what, tuple(z.shape)))


def angle(complex_tensor):
    """Phase ``atan2(im, re)`` of a ``(*, 2)`` tensor (reference: functional.py:187-191)."""
    z = _tensor(complex_tensor, 'complex_tensor')
    _check_pairs(z, 'angle')
    return _call('angle', z)


def magphase(complex_tensor, power=1.):
    """``(|z|**power, atan2(im, re))`` (reference: functional.py:194-201), both outputs from one pass over z."""
    z = _tensor(complex_tensor, 'complex_tensor')
    _check_pairs(z, 'magphase')
    return _call('magphase', z, float(power))


def phase_vocoder(complex_specgrams, rate, phase_advance):
    """Time-stretch a complex spectrogram by ``rate`` without changing pitch (reference:
    functional.py:204-274): ``(*, F, T, 2) → (*, F, ceil(T / rate), 2)``.  On a HIP device one kernel; each lane
    owns one (row, frequency) series and walks the output frames (csrc/phase_vocoder.hip)."""
    spec = _tensor(complex_specgrams, 'complex_specgrams')
    if spec.dim() < 3 or spec.shape[-1] != 2:
        raise RuntimeError('phase_vocoder: expected (*, num_freqs, time, 2), got shape %s' % (tuple(spec.shape),))
    pa = _tensor(phase_advance, 'phase_advance')
    # the reference broadcasts phase_advance against (..., num_freqs, time'): it must be (num_freqs, 1) (or a leading-1
    # variant of it) there, and every route here — HIP kernel, torch operators, fake — sees it in that one shape
    if pa.numel() != spec.shape[-3] or pa.dim() < 2 or pa.shape[-1] != 1 or pa.shape[-2] != spec.shape[-3]:
        raise RuntimeError('phase_vocoder: phase_advance must have shape (num_freqs, 1) = (%d, 1), got %s'
                           % (spec.shape[-3], tuple(pa.shape)))
    pa = pa.reshape(spec.shape[-3], 1)
    if pa.device != spec.device:
        raise RuntimeError('phase_vocoder: spectrogram and phase_advance must be on the same device')
    if not rate > 0:                 # the reference's torch.arange(0, T, rate) raises RuntimeError for such a step
        raise RuntimeError('phase_vocoder: rate must be positive, got %r' % (rate,))
    return _call('phase_vocoder', spec, pa, float(rate))


def amplitude_to_db(x, ref=1.0, amin=1e-7):
    """``10·(log10(max(x², amin)) − log10(ref))`` — the reference squares its input (functional.py:277-296)."""
    return _call('amplitude_to_db', _tensor(x, 'x'), float(ref), float(amin))


def db_to_amplitude(x, ref=1.0):
    """``(10^(x/10 + log10 ref))^0.5`` (reference: functional.py:299-314)."""
    return _call('db_to_amplitude', _tensor(x, 'x'), float(ref))


def mu_law_encoding(x, n_quantize=256):
    """mu-law companding to int64 codes (reference: functional.py:317-335).

    On a HIP device the codes are bit-exact with the reference for every input and ``n_quantize``: for 256 levels
    and ``|x| <= 1`` the kernel compares against the 255 float32 thresholds extracted from the reference
    (``_mulaw_tables.py``), everything else evaluates the closed form with the exact float32 roundings of the
    reference's CPU path (``csrc/exact_math.hpp``)."""
    return _call('mu_law_encoding', _tensor(x, 'x'), int(n_quantize))


def mu_law_decoding(x_mu, n_quantize=256, dtype=torch.get_default_dtype()):
    """mu-law expansion (reference: functional.py:338-354).  On a HIP device codes that are integers in
    ``[0, 256)`` — int64 or float-typed — with ``n_quantize == 256`` are decoded through the reference's own
    256-entry table (bit-exact); everything else evaluates the closed form in fp32 (within 1 ulp of ``exp``)."""
    return _call('mu_law_decoding', _tensor(x_mu, 'x_mu'), int(n_quantize), dtype)


def hpss(mag_specgrams, kernel_size=31, power=2.0, hard=False, mask_only=False):
    """Harmonic / percussive source separation by median filtering (reference: beta_hpss.py:35-127):
    ``(harmonic spectrogram, percussive spectrogram, harmonic mask, percussive mask)`` of a magnitude spectrogram
    ``(*, freq, time)``; with ``mask_only`` the first two are None.  ``kernel_size``: odd int, or ``(width along
    frequency of the percussive filter, width along time of the harmonic filter)`` — the reference only runs with equal
    widths (its padding and slicing mix the two up otherwise); here unequal widths do what its docstring describes.
    Hard masks are bool tensors, as in the reference.  On a HIP device one kernel (csrc/hpss.hip) for widths <= 31."""
    x = _tensor(mag_specgrams, 'mag_specgrams')
    if not isinstance(kernel_size, (tuple, int)):
        raise TypeError('kernel_size is expected to be either tuple of input, but it is: %s' % type(kernel_size))
    kf, kt = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
    if x.dim() < 2:
        raise RuntimeError('hpss: expected (*, freq, time), got shape %s' % (tuple(x.shape),))
    if kf // 2 >= x.shape[-2] or kt // 2 >= x.shape[-1]:
        raise RuntimeError('hpss: reflect padding (%d, %d) must be smaller than the spectrogram size %s'
                           % (kf // 2, kt // 2, tuple(x.shape[-2:])))
    if mask_only:                                         # (the kernels skip the two masked-spectrogram stores)
        harm = perc = None
        mask_h, mask_p = _call('hpss_masks', x, int(kf), int(kt), float(power), bool(hard))
    else:
        harm, perc, mask_h, mask_p = _call('hpss', x, int(kf), int(kt), float(power), bool(hard))
    if hard:
        mask_h, mask_p = mask_h > 0.5, mask_p > 0.5
    return harm, perc, mask_h, mask_p
