"""Functional API — same names, argument order, defaults and error behaviour as the reference's
``torchaudio_contrib/functional.py``; every body launches hand-written gfx950 kernels through the
C ABI in ``include/tac_amd.h`` (no torch compute ops on the hot path, no CPU fallback).

Inputs must live on a HIP device (``tensor.is_cuda``) and be float32 (float16/bfloat16 are widened,
as torch.stft does for half input in the reference).  Outputs are fresh tensors; the STFT-family
results are returned as the same strided views the reference produces (physically frame-major,
logically ``(*, channel, freq, time[, 2])``).
"""
import ctypes
import math
import os
import threading

import torch

from . import _native
from ._lazy import realize as _realize

__all__ = ['stft', 'complex_norm', 'create_mel_filter', 'apply_filterbank', 'angle', 'magphase',
           'phase_vocoder', 'amplitude_to_db', 'db_to_amplitude', 'mu_law_encoding', 'mu_law_decoding']


# ----------------------------------------------------------------------------- helpers
def _device_f32(x, what):
    """Validate + normalise an input tensor for the HIP path (never falls back to CPU)."""
    if not torch.is_tensor(x):
        raise TypeError('%s must be a torch.Tensor, got %s' % (what, type(x).__name__))
    x = _realize(x)
    if not x.is_cuda:
        raise RuntimeError('%s is on %s: torchaudio_contrib_amd only runs on a HIP device (MI355X); '
                           'move the tensor with .cuda() — there is no CPU path' % (what, x.device))
    if x.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError('%s requires grad: the gfx950 kernels are forward-only' % what)
    if x.dtype in (torch.float16, torch.bfloat16):
        x = x.float()
    if x.dtype != torch.float32:
        raise NotImplementedError('%s has dtype %s: the gfx950 kernels compute in float32' % (what, x.dtype))
    return x


def _is_dense(x):
    """True when x's elements tile one gap-free block of memory (in any dim order)."""
    if x.is_contiguous():
        return True
    dims = sorted((st, n) for st, n in zip(x.stride(), x.shape) if n > 1)
    expect = 1
    for st, n in dims:
        if st != expect:
            return False
        expect *= n
    return True


def _dense(x):
    """Return x if it is non-overlapping & dense (any dim order), else a contiguous copy."""
    return x if _is_dense(x) else x.contiguous()


_window_cache = {}
_cache_lock = threading.Lock()


def _default_window(n, device):
    key = (n, str(device))
    w = _window_cache.get(key)
    if w is None:
        # periodic Hann, same constructor the reference calls (functional.py:93-97, layers.py:76-80)
        w = torch.hann_window(n, device=device)
        with _cache_lock:
            _window_cache[key] = w
    return w


class _StftPlan(object):
    """Validated geometry of one stft call (mirrors the checks torch.stft performs)."""
    __slots__ = ('wave', 'window', 'desc', 'lead', 'n_frames', 'n_bins', 'n_fft', 'onesided', 'fft_kernel',
                 'hop', 'center', 'pad_mode', 'normalized', 'win_length')

    def __init__(self, waveforms, fft_length, hop_length, win_length, window, center, pad_mode,
                 normalized, onesided):
        x = _device_f32(waveforms, 'waveforms')
        if x.dim() < 1 or x.numel() == 0:
            raise RuntimeError('stft: expected a non-empty tensor of shape (*, channel, time)')
        n_fft = int(fft_length)
        hop = n_fft // 4 if hop_length is None else int(hop_length)
        win_length = n_fft if win_length is None else int(win_length)
        length = x.shape[-1]
        if n_fft <= 0 or hop <= 0:
            raise RuntimeError('stft: expected 0 < n_fft and 0 < hop_length, got n_fft=%d hop_length=%d'
                               % (n_fft, hop))
        if not 0 < win_length <= n_fft:
            raise RuntimeError('stft: expected 0 < win_length <= n_fft, got win_length=%d' % win_length)
        if window is None:
            window = _default_window(win_length, x.device)
        else:
            if not torch.is_tensor(window) or window.dim() != 1 or window.shape[0] != win_length:
                raise RuntimeError('stft: expected a 1D window tensor of size equal to win_length=%d'
                                   % win_length)
            if window.device != x.device:
                raise RuntimeError('stft: input and window must be on the same device, got %s and %s'
                                   % (x.device, window.device))
            window = _device_f32(window, 'window').contiguous()
        if pad_mode not in _native.PAD_MODES:
            raise NotImplementedError('stft: unsupported pad_mode %r' % (pad_mode,))
        pad = n_fft // 2 if center else 0
        if center and pad_mode == 'reflect' and pad >= length:
            raise RuntimeError('stft: reflect padding (%d, %d) must be smaller than the signal length %d'
                               % (pad, pad, length))
        if center and pad_mode == 'circular' and pad > length:
            raise RuntimeError('stft: circular padding (%d, %d) must not exceed the signal length %d'
                               % (pad, pad, length))
        if length + 2 * pad < n_fft:
            raise RuntimeError('stft: expected n_fft <= padded signal length %d, got n_fft=%d'
                               % (length + 2 * pad, n_fft))
        # power-of-two sizes in [32, 4096] take the wave-level FFT kernels; every other size up to 8192 is
        # evaluated as a windowed-DFT matrix product on the fp32 matrix cores (see _run_stft_dft)
        self.fft_kernel = (n_fft & (n_fft - 1)) == 0 and 32 <= n_fft <= 4096
        if not self.fft_kernel and n_fft > 8192:
            raise NotImplementedError('stft: fft_length %d is outside the HIP path (power of two in [32, 4096], '
                                      'or any length <= 8192 through the DFT-matrix kernel)' % n_fft)
        self.lead = tuple(x.shape[:-1])
        flat = x.reshape(-1, length)
        if flat.stride(1) != 1 or (flat.shape[0] > 1 and flat.stride(0) < length):
            flat = flat.contiguous()
        self.wave = flat
        self.window = window
        self.n_fft = n_fft
        self.onesided = bool(onesided)
        self.hop, self.center, self.pad_mode = hop, bool(center), pad_mode
        self.normalized, self.win_length = bool(normalized), win_length
        self.n_frames = 1 + (length + 2 * pad - n_fft) // hop
        self.n_bins = n_fft // 2 + 1 if onesided else n_fft
        self.desc = None if not self.fft_kernel else _native.StftDesc(
            rows=flat.shape[0], length=length, row_stride=flat.stride(0) if flat.shape[0] > 1 else length,
            n_fft=n_fft, hop=hop, win_length=win_length, center=1 if center else 0,
            pad_mode=_native.PAD_MODES[pad_mode], normalized=1 if normalized else 0,
            onesided=1 if onesided else 0, reserved=0)

    # launches ------------------------------------------------------------------
    def _run_stft_dft(self):
        """Any fft_length (non power of two, or 4096 < N <= 8192): the framed signal is never materialised — the
        filterbank GEMM kernel reads frame t, sample n at ``padded[row, t*hop + n]`` (stride_f = 1, stride_t = hop)
        and multiplies by the (N, 2F) matrix w[n]·(cos, -sin)(2*pi*k*n/N) on v_mfma_f32_16x16x4_f32.  The padded
        copy is plain data movement done by torch; everything arithmetic is the HIP kernel."""
        x = self.wave
        if self.center:
            pad = self.n_fft // 2
            x = torch.nn.functional.pad(x.unsqueeze(1), (pad, pad), mode=self.pad_mode).squeeze(1)
        x = x.contiguous()
        mat = _dft_matrix(self.window, self.n_fft, self.win_length, self.onesided, self.normalized)
        out = torch.empty(self.lead + (self.n_frames, self.n_bins, 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _native.lib().tac_apply_filterbank_f32(
                _native.ptr(x), x.shape[0], self.n_fft, self.n_frames, x.stride(0), 1, self.hop,
                _native.ptr(mat), None, 2 * self.n_bins, _native.ptr(out), _native.stream_ptr(x.device))
        _native.check(rc, 'tac_apply_filterbank_f32 (DFT matrix)')
        return out.transpose(-3, -2)

    def run_stft(self):
        if not self.fft_kernel:
            return self._run_stft_dft()
        out = torch.empty(self.lead + (self.n_frames, self.n_bins, 2), dtype=torch.float32,
                          device=self.wave.device)
        with torch.cuda.device(self.wave.device):
            rc = _native.lib().tac_stft_f32(_native.ptr(self.wave), _native.ptr(self.window), self.desc,
                                            _native.ptr(out), _native.stream_ptr(self.wave.device))
        _native.check(rc, 'tac_stft_f32')
        return out.transpose(-3, -2)

    def run_spectrogram(self, power, db=None):
        if not self.fft_kernel:
            mag = complex_norm(self._run_stft_dft(), power)
            return mag if db is None else amplitude_to_db(mag, ref=db[0], amin=db[1])
        out = torch.empty(self.lead + (self.n_frames, self.n_bins), dtype=torch.float32,
                          device=self.wave.device)
        ref, amin = db if db is not None else (1.0, 1e-7)
        with torch.cuda.device(self.wave.device):
            rc = _native.lib().tac_spectrogram_f32(
                _native.ptr(self.wave), _native.ptr(self.window), self.desc, float(power),
                1 if db is not None else 0, float(ref), float(amin), _native.ptr(out),
                _native.stream_ptr(self.wave.device))
        _native.check(rc, 'tac_spectrogram_f32')
        return out.transpose(-2, -1)

    def can_fuse_mel(self, filterbank, power=2.0):
        """True when the single fused kernel covers this geometry and filterbank (sparse enough for the
        register-resident weights); otherwise the caller chains spectrogram + apply_filterbank kernels."""
        if not (self.fft_kernel and self.onesided and self.n_fft <= 2048 and filterbank.dim() == 2 and
                filterbank.shape[0] == self.n_bins and 0 < filterbank.shape[1] <= 512 and
                filterbank.is_cuda and filterbank.dtype == torch.float32 and
                filterbank.device == self.wave.device and filterbank.is_contiguous()):
            return False
        if power in (1.0, 2.0) and _MEL_PATH != 'mfma' and _melbank_pack(filterbank, self.n_fft) is not None:
            return True
        if _MEL_PATH == 'sparse':
            return False
        _, host = _filterbank_plan(filterbank)
        rc = _native.lib().tac_melspec_supported(self.desc, float(power), ctypes.cast(host, ctypes.c_void_p),
                                                 filterbank.shape[1])
        return rc == _native.TAC_OK

    def run_melspec(self, power, filterbank, db=None):
        fb = filterbank
        n_mels = fb.shape[1]
        out = torch.empty(self.lead + (self.n_frames, n_mels), dtype=torch.float32, device=self.wave.device)
        ref, amin = db if db is not None else (1.0, 1e-7)
        pack = _melbank_pack(fb, self.n_fft) if (_MEL_PATH != 'mfma' and power in (1.0, 2.0)) else None
        if pack is not None:          # band-sparse contraction (the faster form for triangular banks)
            wpack, desc, info = pack
            with torch.cuda.device(self.wave.device):
                rc = _native.lib().tac_melspec_sparse_f32(
                    _native.ptr(self.wave), _native.ptr(self.window), self.desc, float(power), _native.ptr(wpack),
                    _native.ptr(desc), ctypes.cast(info, ctypes.c_void_p), n_mels, 1 if db is not None else 0,
                    float(ref), float(amin), _native.ptr(out), _native.stream_ptr(self.wave.device))
            _native.check(rc, 'tac_melspec_sparse_f32')
            return out.transpose(-2, -1)
        _, plan_host = _filterbank_plan(fb)
        with torch.cuda.device(self.wave.device):
            rc = _native.lib().tac_melspec_f32(
                _native.ptr(self.wave), _native.ptr(self.window), self.desc, float(power), _native.ptr(fb),
                ctypes.cast(plan_host, ctypes.c_void_p), n_mels, 1 if db is not None else 0, float(ref), float(amin),
                _native.ptr(out), _native.stream_ptr(self.wave.device))
        _native.check(rc, 'tac_melspec_f32')
        return out.transpose(-2, -1)


def _dft_matrix(window, n_fft, win_length, onesided, normalized):
    """(N, 2F) float32 device matrix [w[n] cos(2 pi k n / N), -w[n] sin(2 pi k n / N)] for the DFT-matrix path;
    evaluated once per (window tensor version, geometry) in float64 on the host — a constant table like the FFT
    twiddles — and cached on the window tensor."""
    import numpy as np
    cache = getattr(window, '_tac_dft', None)
    key = (window._version, n_fft, win_length, bool(onesided), bool(normalized))
    if cache is not None and cache[0] == key:
        return cache[1]
    w = np.zeros(n_fft, dtype=np.float64)
    off = (n_fft - win_length) // 2
    w[off:off + win_length] = window.detach().double().cpu().numpy()
    if normalized:
        w = w / math.sqrt(n_fft)
    n_bins = n_fft // 2 + 1 if onesided else n_fft
    n = np.arange(n_fft, dtype=np.int64)[:, None]
    k = np.arange(n_bins, dtype=np.int64)[None, :]
    ang = (2.0 * math.pi / n_fft) * ((n * k) % n_fft).astype(np.float64)
    mat = np.empty((n_fft, n_bins, 2), dtype=np.float32)
    mat[..., 0] = np.cos(ang) * w[:, None]
    mat[..., 1] = -np.sin(ang) * w[:, None]
    dev = torch.from_numpy(mat.reshape(n_fft, 2 * n_bins)).to(window.device)
    try:
        window._tac_dft = (key, dev)
    except Exception:
        pass
    return dev


# A/B knob for the two fused Melspectrogram kernels: 'auto' (band-sparse when the bank allows it, else MFMA),
# 'sparse', 'mfma'
_MEL_PATH = os.environ.get('TAC_MEL_PATH', 'auto')


def _melbank_pack(fb, n_fft):
    """(wpack, desc, info) device/host buffers of the band-sparse contraction for this filterbank and fft size, or
    None when the bank is not band-sparse enough (then the MFMA kernels are used).  Built once per filterbank
    version (one host sync) and cached on the tensor object."""
    cache = getattr(fb, '_tac_pack', None)
    if cache is None or cache[0] != fb._version:
        cache = (fb._version, {})
        try:
            fb._tac_pack = cache
        except Exception:
            pass
    if n_fft in cache[1]:
        return cache[1][n_fft]
    n_freqs, n_mels = fb.shape
    wpack = torch.empty(3072, dtype=torch.float32, device=fb.device)
    desc = torch.empty(4096, dtype=torch.int32, device=fb.device)
    info = (ctypes.c_int32 * 4)()
    with torch.cuda.device(fb.device):
        rc = _native.lib().tac_melbank_pack(_native.ptr(fb), n_freqs, n_mels, n_fft, _native.ptr(wpack), 3072,
                                            _native.ptr(desc), 4096, ctypes.cast(info, ctypes.c_void_p),
                                            _native.stream_ptr(fb.device))
    if rc == _native.TAC_E_UNSUPPORTED:
        result = None
    else:
        _native.check(rc, 'tac_melbank_pack')
        result = (wpack, desc, info)
    cache[1][n_fft] = result
    return result


def _filterbank_plan(fb):
    """(device int32 plan, host ctypes copy): non-zero bin range per 16-band tile, computed by a device
    kernel.  The plan rides on the filterbank tensor object itself (a module's constant buffer is scanned
    once — the only host sync on the path — and rescanned when modified in place); keying a cache on
    ``data_ptr`` would go stale when the allocator reuses an address."""
    hit = getattr(fb, '_tac_plan', None)
    if hit is not None and hit[0] == fb._version and hit[1].device == fb.device:
        return hit[1], hit[2]
    n_freqs, n_mels = fb.shape
    n_ints = 2 * ((n_mels + 15) // 16)
    plan = torch.empty(n_ints, dtype=torch.int32, device=fb.device)
    host = (ctypes.c_int32 * n_ints)()
    with torch.cuda.device(fb.device):
        rc = _native.lib().tac_filterbank_plan(_native.ptr(fb), n_freqs, n_mels, _native.ptr(plan),
                                               ctypes.cast(host, ctypes.c_void_p),
                                               _native.stream_ptr(fb.device))
    _native.check(rc, 'tac_filterbank_plan')
    try:
        fb._tac_plan = (fb._version, plan, host)
    except Exception:       # exotic tensor subclasses without attribute storage: just recompute next time
        pass
    return plan, host


# ----------------------------------------------------------------------------- public API
def stft(waveforms, fft_length, hop_length=None, win_length=None, window=None,
         center=True, pad_mode='reflect', normalized=False, onesided=True):
    """Short-time Fourier transform of ``(*, channel, time)`` waveforms →
    ``(*, channel, num_freqs, time, complex=2)``  (reference: functional.py:48-113).

    ``window=None`` means a periodic Hann window of ``win_length or fft_length`` (unlike torch.stft).
    Framing, padding (``center``/``pad_mode``), windowing and the R2C FFT run in one gfx950 kernel.
    """
    return _StftPlan(waveforms, fft_length, hop_length, win_length, window, center, pad_mode,
                     normalized, onesided).run_stft()


def complex_norm(complex_tensor, power=1.0):
    """``|z|**power`` over a trailing ``complex=2`` dim (reference: functional.py:116-128)."""
    z = _complex_pairs(complex_tensor, 'complex_norm')
    out = _pair_output(z)
    n = out.numel()
    if n:
        with torch.cuda.device(z.device):
            rc = _native.lib().tac_complex_norm_f32(_native.ptr(z), n, float(power), _native.ptr(out),
                                                    _native.stream_ptr(z.device))
        _native.check(rc, 'tac_complex_norm_f32')
    return out


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
    functional.py:172-184) on the fp32 matrix cores, skipping the zero blocks of sparse banks."""
    spec = _device_f32(mag_specgrams, 'mag_specgrams')
    fb = _device_f32(filterbank, 'filterbank')
    if fb.dim() != 2 or spec.dim() < 2 or spec.shape[-2] != fb.shape[0]:
        raise RuntimeError('apply_filterbank: size mismatch, spectrogram %s vs filterbank %s'
                           % (tuple(spec.shape), tuple(fb.shape)))
    if fb.device != spec.device:
        raise RuntimeError('apply_filterbank: spectrogram and filterbank must be on the same device')
    fb = fb if fb.is_contiguous() else fb.contiguous()
    n_freqs, n_frames = spec.shape[-2], spec.shape[-1]
    lead = tuple(spec.shape[:-2])
    n_mels = fb.shape[1]
    out = torch.empty(lead + (n_frames, n_mels), dtype=torch.float32, device=spec.device)
    if out.numel():
        rows = spec.reshape(-1, n_freqs, n_frames)
        # frame-major spectrogram (what the kernels here produce) + band-sparse bank: stream it through the fused
        # kernel's contraction; anything else goes through the fp32 MFMA GEMM
        pack = _melbank_pack(fb, 0) if (rows.stride(1) == 1 and _MEL_PATH != 'mfma') else None
        if pack is not None:
            wpack, desc, info = pack
            with torch.cuda.device(spec.device):
                rc = _native.lib().tac_apply_filterbank_sparse_f32(
                    _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0) if rows.shape[0] > 1 else 0,
                    rows.stride(2), _native.ptr(wpack), _native.ptr(desc), ctypes.cast(info, ctypes.c_void_p), n_mels,
                    _native.ptr(out), _native.stream_ptr(spec.device))
            if rc != _native.TAC_E_UNSUPPORTED:
                _native.check(rc, 'tac_apply_filterbank_sparse_f32')
                return out.transpose(-2, -1)
        plan, _ = _filterbank_plan(fb)
        with torch.cuda.device(spec.device):
            rc = _native.lib().tac_apply_filterbank_f32(
                _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0), rows.stride(1),
                rows.stride(2), _native.ptr(fb), _native.ptr(plan), n_mels, _native.ptr(out),
                _native.stream_ptr(spec.device))
        _native.check(rc, 'tac_apply_filterbank_f32')
    return out.transpose(-2, -1)


def _complex_pairs(complex_tensor, what):
    """Dense view of a ``(*, 2)`` tensor whose storage order the elementwise kernels can walk pair by pair."""
    z = _device_f32(complex_tensor, what)
    if z.dim() < 1 or z.shape[-1] != 2:
        raise RuntimeError('%s: expected a trailing dimension of size 2, got shape %s' % (what, tuple(z.shape)))
    if z.stride(-1) != 1 or not _is_dense(z) or \
            any(s % 2 for s, n in zip(z.stride()[:-1], z.shape[:-1]) if n > 1):
        z = z.contiguous()
    return z


def _pair_output(z):
    return torch.empty_strided(z.shape[:-1], tuple(s // 2 for s in z.stride()[:-1]), dtype=torch.float32,
                               device=z.device)


def angle(complex_tensor):
    """Phase ``atan2(im, re)`` of a ``(*, 2)`` tensor (reference: functional.py:187-191); one streaming kernel
    (SURVEY §8f rank 1)."""
    z = _complex_pairs(complex_tensor, 'complex_tensor')
    phase = _pair_output(z)
    if phase.numel():
        with torch.cuda.device(z.device):
            rc = _native.lib().tac_magphase_f32(_native.ptr(z), phase.numel(), 1.0, None, _native.ptr(phase),
                                                _native.stream_ptr(z.device))
        _native.check(rc, 'tac_magphase_f32')
    return phase


def magphase(complex_tensor, power=1.):
    """``(|z|**power, atan2(im, re))`` (reference: functional.py:194-201), both outputs from one pass over z."""
    z = _complex_pairs(complex_tensor, 'complex_tensor')
    mag, phase = _pair_output(z), _pair_output(z)
    if phase.numel():
        with torch.cuda.device(z.device):
            rc = _native.lib().tac_magphase_f32(_native.ptr(z), phase.numel(), float(power), _native.ptr(mag),
                                                _native.ptr(phase), _native.stream_ptr(z.device))
        _native.check(rc, 'tac_magphase_f32')
    return mag, phase


_PV_GRID_CACHE = {}


def _phase_vocoder_grid(n_frames, rate, device):
    """Source-frame indices and interpolation weights of every output frame, evaluated exactly as the reference's CPU
    path does (functional.py:233-247: float32 ``torch.arange(0, T, rate)``, ``% 1``, ``.long()``); which frames get
    paired depends on that rounding, so it is computed with the same host ops and cached per (T, rate, device)."""
    key = (int(n_frames), float(rate), str(device))
    hit = _PV_GRID_CACHE.get(key)
    if hit is None:
        steps = torch.arange(0, n_frames, rate)
        if len(_PV_GRID_CACHE) > 64:
            _PV_GRID_CACHE.clear()
        hit = (steps.long().to(torch.int32).to(device), (steps + 1).long().to(torch.int32).to(device),
               torch.remainder(steps, torch.tensor(1.)).to(device))
        _PV_GRID_CACHE[key] = hit
    return hit


def phase_vocoder(complex_specgrams, rate, phase_advance):
    """Time-stretch a complex spectrogram by ``rate`` without changing pitch (reference:
    functional.py:204-274): ``(*, F, T, 2) → (*, F, ceil(T / rate), 2)``.  One kernel; each lane
    owns one (row, frequency) series and walks the output frames (csrc/phase_vocoder.hip)."""
    spec = _device_f32(complex_specgrams, 'complex_specgrams')
    if spec.dim() < 3 or spec.shape[-1] != 2:
        raise RuntimeError('phase_vocoder: expected (*, num_freqs, time, 2), got shape %s' % (tuple(spec.shape),))
    n_freqs, n_frames = spec.shape[-3], spec.shape[-2]
    pa = _device_f32(phase_advance, 'phase_advance').reshape(-1).contiguous()
    if pa.numel() != n_freqs:
        raise RuntimeError('phase_vocoder: phase_advance has %d entries for %d frequency bins' % (pa.numel(), n_freqs))
    if pa.device != spec.device:
        raise RuntimeError('phase_vocoder: spectrogram and phase_advance must be on the same device')
    if not rate > 0:
        raise ValueError('phase_vocoder: rate must be positive, got %r' % (rate,))
    lead = tuple(spec.shape[:-3])
    idx0, idx1, alpha = _phase_vocoder_grid(n_frames, rate, spec.device)
    n_out = idx0.numel()
    if spec.stride(-1) != 1:
        spec = spec.contiguous()
    rows = spec.reshape((-1,) + tuple(spec.shape[-3:]))          # a view whenever the leading dims collapse
    out = torch.empty(lead + (n_out, n_freqs, 2), dtype=torch.float32, device=spec.device)
    if out.numel() and n_frames:
        with torch.cuda.device(spec.device):
            rc = _native.lib().tac_phase_vocoder_f32(
                _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0) if rows.shape[0] > 1 else 0,
                rows.stride(1), rows.stride(2), _native.ptr(pa), _native.ptr(idx0), _native.ptr(idx1),
                _native.ptr(alpha), n_out, _native.ptr(out), _native.stream_ptr(spec.device))
        _native.check(rc, 'tac_phase_vocoder_f32')
    return out.transpose(-3, -2)


def _unary(x, what, launch):
    x = _dense(_device_f32(x, what))
    out = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(x.device):
            rc = launch(_native.lib(), _native.ptr(x), x.numel(), _native.ptr(out), _native.stream_ptr(x.device))
        _native.check(rc, what)
    return out


def amplitude_to_db(x, ref=1.0, amin=1e-7):
    """``10·(log10(max(x², amin)) − log10(ref))`` — the reference squares its input
    (functional.py:277-296)."""
    return _unary(x, 'amplitude_to_db',
                  lambda h, p, n, o, s: h.tac_amplitude_to_db_f32(p, n, float(ref), float(amin), o, s))


def db_to_amplitude(x, ref=1.0):
    """``(10^(x/10 + log10 ref))^0.5`` (reference: functional.py:299-314)."""
    return _unary(x, 'db_to_amplitude',
                  lambda h, p, n, o, s: h.tac_db_to_amplitude_f32(p, n, float(ref), o, s))


_mulaw_consts = {}


def _mulaw_tables(device):
    key = str(device)
    hit = _mulaw_consts.get(key)
    if hit is None:
        from . import _mulaw_tables as tab
        thr = torch.tensor(list(tab.THR256_POS) + list(tab.THR256_NEG), dtype=torch.int32, device=device)
        lut = torch.tensor(list(tab.LUT256_BITS), dtype=torch.int64).to(torch.int32).view(torch.float32)
        hit = (thr, len(tab.THR256_POS), len(tab.THR256_NEG), tab.ZERO_CODE_256, lut.to(device))
        with _cache_lock:
            _mulaw_consts[key] = hit
    return hit


def mu_law_encoding(x, n_quantize=256):
    """mu-law companding to int64 codes (reference: functional.py:317-335).

    For ``n_quantize == 256`` and ``|x| <= 1`` the codes are bit-exact with the reference: the kernel
    compares against the 255 float32 thresholds extracted from it (``_mulaw_tables.py``).  Other
    ``n_quantize`` and out-of-range samples evaluate the closed form with the exact float32 roundings
    of the reference's CPU path (``csrc/exact_math.hpp``) and are bit-exact as well."""
    if torch.is_tensor(x) and not x.dtype.is_floating_point:
        x = _realize(x).to(torch.float)
    x = _device_f32(x, 'x')
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    if x.numel():
        n_quantize = int(n_quantize)
        if n_quantize == 256:
            thr, n_pos, n_neg, zero, _ = _mulaw_tables(x.device)
            thr_ptr = _native.ptr(thr)
        else:
            thr_ptr, n_pos, n_neg, zero = None, 0, 0, 0
        with torch.cuda.device(x.device):
            rc = _native.lib().tac_mulaw_encode_f32_i64(_native.ptr(x), x.numel(), n_quantize, thr_ptr,
                                                        n_pos, n_neg, zero, _native.ptr(out),
                                                        _native.stream_ptr(x.device))
        _native.check(rc, 'tac_mulaw_encode_f32_i64')
    return out


def mu_law_decoding(x_mu, n_quantize=256, dtype=torch.get_default_dtype()):
    """mu-law expansion (reference: functional.py:338-354).  Integer codes in ``[0, 256)`` with
    ``n_quantize == 256`` are decoded through the reference's own 256-entry table (bit-exact);
    everything else evaluates the closed form in fp32."""
    codes = _realize(x_mu)
    if not torch.is_tensor(codes):
        raise TypeError('x_mu must be a torch.Tensor')
    n_quantize = int(n_quantize)
    if not codes.dtype.is_floating_point:
        if dtype != torch.float32:
            raise NotImplementedError('mu_law_decoding: the gfx950 kernel decodes to float32, got %s' % dtype)
        if not codes.is_cuda:
            raise RuntimeError('x_mu is on %s: torchaudio_contrib_amd only runs on a HIP device' % codes.device)
        codes = codes.to(torch.int64).contiguous()
        out = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
        if codes.numel():
            lut_ptr = _native.ptr(_mulaw_tables(codes.device)[4]) if n_quantize == 256 else None
            with torch.cuda.device(codes.device):
                rc = _native.lib().tac_mulaw_decode_i64_f32(_native.ptr(codes), codes.numel(), n_quantize,
                                                            lut_ptr, _native.ptr(out),
                                                            _native.stream_ptr(codes.device))
            _native.check(rc, 'tac_mulaw_decode_i64_f32')
        return out
    return _unary(codes, 'mu_law_decoding',
                  lambda h, p, n, o, s: h.tac_mulaw_decode_f32_f32(p, n, n_quantize, o, s))
