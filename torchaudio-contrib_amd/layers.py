"""Layer API — the ``nn.Module`` surface of the reference's ``torchaudio_contrib/layers.py``
(same class names, constructor signatures, attributes, buffers, ``__repr__`` strings and
exceptions), backed by the gfx950 kernels in ``csrc/``.

Chains of these layers fuse automatically: ``STFT`` hands a deferred result to ``ComplexNorm`` →
``ApplyFilterbank`` → ``AmplitudeToDb`` (see ``_lazy.py``), so
``nn.Sequential(*Melspectrogram(...), AmplitudeToDb())`` — the reference's own idiom — is a single
kernel launch that reads the waveform once and writes only the mel-dB tensor.
"""
import math

import torch
import torch.nn as nn

from . import functional as F
from . import _hip
from . import _lazy
from ._lazy import DeferredSpectral, DeferredWave, can_defer, can_defer_codes, lazy_fusion_enabled, realize


class _ModuleNoStateBuffers(nn.Module):
    """Module whose buffers (window, filterbank, phase_advance) are derived constants: they follow
    ``.to()/.cuda()`` but never enter ``state_dict()`` and are ignored when loading one
    (contract of reference layers.py:11-32; ``state_dict()`` of a whole pipeline is empty)."""

    def state_dict(self, *args, **kwargs):
        prefix = kwargs.get('prefix', args[1] if len(args) > 1 else '')
        full = super(_ModuleNoStateBuffers, self).state_dict(*args, **kwargs)
        for name in self._buffers:
            full.pop(prefix + name, None)
        return full

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        kept, self._buffers = self._buffers, {}
        try:
            return super(_ModuleNoStateBuffers, self)._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        finally:
            self._buffers = kept


class STFT(_ModuleNoStateBuffers):
    """Short-time Fourier transform layer: ``(*, channel, time)`` → ``(*, channel, num_freqs, time, 2)``.

    Arguments and defaults are those of the reference (layers.py:35-109): ``hop_length`` defaults to
    ``fft_length // 4``, ``win_length`` to ``fft_length``, ``window`` to a periodic Hann window of
    ``win_length``; ``center``/``pad_mode``/``normalized``/``onesided`` as in ``torch.stft``.
    """

    def __init__(self, fft_length, hop_length=None, win_length=None,
                 window=None, center=True, pad_mode='reflect',
                 normalized=False, onesided=True):
        super(STFT, self).__init__()
        self.fft_length = fft_length
        self.hop_length = hop_length
        self.win_length = win_length
        self.center = center
        self.pad_mode = pad_mode
        self.normalized = normalized
        self.onesided = onesided
        if window is None:
            window = torch.hann_window(fft_length if win_length is None else win_length)
        self.register_buffer('window', window)

    def forward(self, waveforms):
        if torch.is_tensor(waveforms) and can_defer(waveforms, self.window):
            # validate now, so errors surface here; the launch itself waits for the rest of the chain (_lazy.py).  The
            # resolved arguments and the recipe's shapes depend on the input's shape only: remembered per shape, so that a
            # repeated call neither re-validates nor re-derives them (the attributes are read on every call: changing one
            # is seen)
            key = (waveforms.shape, self.fft_length, self.hop_length, self.win_length, self.center, self.pad_mode,
                   self.normalized, self.onesided, id(self.window))
            hit = self.__dict__.get('_recipes')
            if hit is None:
                hit = self.__dict__['_recipes'] = {}
            tmpl = hit.get(key)
            if tmpl is None or tmpl[0] is not self.window:
                n_fft, hop, win_length, window = F.resolve_stft_args(waveforms, self.fft_length, self.hop_length,
                                                                     self.win_length, self.window)
                _hip.check_stft_args(waveforms.shape, n_fft, hop, win_length, self.center, self.pad_mode)
                if window is self.window:
                    if len(hit) > 32:
                        hit.clear()
                    tmpl = hit[key] = (window, DeferredSpectral.template(
                        waveforms.shape, n_fft, hop, win_length, bool(self.center), self.pad_mode, bool(self.normalized),
                        bool(self.onesided)))
                else:       # (a window resolve_stft_args had to build or move: not remembered)
                    d = DeferredSpectral.from_stft(waveforms, window, n_fft, hop, win_length, bool(self.center),
                                                   self.pad_mode, bool(self.normalized), bool(self.onesided))
                    return d.realize() if _lazy.ends_chain(self) else d
            d = DeferredSpectral.from_template(waveforms, tmpl[0], tmpl[1])
            return d.realize() if _lazy.ends_chain(self) else d        # (the last layer of a user's nn.Sequential: an ordinary tensor)
        return F.stft(waveforms, self.fft_length, self.hop_length, self.win_length, self.window, self.center,
                      self.pad_mode, self.normalized, self.onesided)

    def __repr__(self):
        head = '(fft_length={}, hop_length={}, win_length={})'.format(
            self.fft_length, self.hop_length, self.win_length)
        tail = '(center={}, pad_mode={}, normalized={}, onesided={})'.format(
            self.center, self.pad_mode, self.normalized, self.onesided)
        return self.__class__.__name__ + head + tail


class ComplexNorm(nn.Module):
    """``|z| ** power`` over the trailing complex dim (reference layers.py:112-135)."""

    def __init__(self, power=1.0):
        super(ComplexNorm, self).__init__()
        self.power = power

    def forward(self, complex_tensor):
        if isinstance(complex_tensor, DeferredSpectral) and complex_tensor.pending() \
                and complex_tensor._stage == 'stft':
            d = complex_tensor.with_norm(self.power)
            return d.realize() if _lazy.ends_chain(self) else d
        return F.complex_norm(complex_tensor, self.power)

    def __repr__(self):
        return self.__class__.__name__ + '(power={})'.format(self.power)


class ApplyFilterbank(_ModuleNoStateBuffers):
    """Multiply the frequency axis by a ``(num_freqs, num_bands)`` matrix held as the non-persistent
    buffer ``filterbank`` (reference layers.py:138-155)."""

    def __init__(self, filterbank):
        super(ApplyFilterbank, self).__init__()
        self.register_buffer('filterbank', filterbank)

    def forward(self, mag_specgrams):
        x = mag_specgrams
        fb = self.filterbank
        # (the recipe's own record of bins / device: attribute access on the wrapper goes through __torch_function__)
        if isinstance(x, DeferredSpectral) and x.pending() and x._stage == 'spec' \
                and fb.dim() == 2 and fb.shape[0] == x._src.n_bins and fb.device == x._src.wave.device \
                and fb.dtype == torch.float32 and not (fb.requires_grad and torch.is_grad_enabled()):
            d = x.with_filterbank(fb)
            return d.realize() if _lazy.ends_chain(self) else d
        return F.apply_filterbank(x, fb)


class Filterbank(object):
    """Abstract provider of a filterbank matrix (reference layers.py:158-167)."""

    def __init__(self):
        super(Filterbank, self).__init__()

    def get_filterbank(self):
        raise NotImplementedError


class MelFilterbank(Filterbank):
    """Mel filterbank provider (reference layers.py:170-212): ``max_freq`` defaults to
    ``sample_rate // 2``; one of the two must be given."""

    def __init__(self, num_freqs=1025, num_mels=128,
                 min_freq=0.0, max_freq=None, sample_rate=None, htk=False):
        super(MelFilterbank, self).__init__()
        if sample_rate is None and max_freq is None:
            raise ValueError('Either max_freq or sample_rate should be specified.'
                             ', but both are None.')
        self.num_freqs = num_freqs
        self.num_mels = num_mels
        self.min_freq = min_freq
        self.max_freq = max_freq if max_freq else sample_rate // 2
        self.htk = htk

    def get_filterbank(self):
        return F.create_mel_filter(num_freqs=self.num_freqs, num_mels=self.num_mels,
                                   min_freq=self.min_freq, max_freq=self.max_freq, htk=self.htk)

    def __repr__(self):
        # string kept byte-for-byte (typo and bracket order included): it is visible API
        a = '(num_freqs={}, snum_mels={}'.format(self.num_freqs, self.num_mels)
        b = ', min_freq={}, max_freq={})'.format(self.min_freq, self.max_freq)
        c = ', htk={}'.format(self.htk)
        return self.__class__.__name__ + a + b + c


class TimeStretch(_ModuleNoStateBuffers):
    """Phase-vocoder time stretch of a complex spectrogram (reference layers.py:215-264).  Sits
    outside the Melspectrogram hot path; a deferred STFT handed to it is materialised first."""

    def __init__(self, hop_length, num_freqs, fixed_rate=None):
        super(TimeStretch, self).__init__()
        self.fixed_rate = fixed_rate
        self.register_buffer('phase_advance',
                             torch.linspace(0, math.pi * hop_length, num_freqs)[..., None])

    def forward(self, complex_specgrams, overriding_rate=None):
        rate = self.fixed_rate if overriding_rate is None else overriding_rate
        if rate is None:
            raise ValueError("If no fixed_rate is specified"
                             ", must pass a valid rate to the forward method.")
        if rate == 1.0:
            return complex_specgrams
        return F.phase_vocoder(complex_specgrams, rate, self.phase_advance)

    def __repr__(self):
        return self.__class__.__name__ + '(fixed_rate={})'.format(self.fixed_rate)


class _FusedSequential(nn.Sequential):
    """``nn.Sequential`` returned by the factories: children stay individually usable and ``*``-unpackable; a
    whole-chain call is ONE ``tac_amd::spectrogram`` / ``tac_amd::melspectrogram`` op (one kernel on a HIP device,
    differentiable, traceable by ``torch.compile``) and returns an ordinary tensor."""
    _tac_realizes = True            # (_lazy.ends_chain: this container launches what its children deferred by itself)

    def forward(self, input):
        kids = list(self._modules.values())
        if lazy_fusion_enabled() and torch.is_tensor(input) and 2 <= len(kids) <= 3 and type(kids[0]) is STFT and type(kids[1]) is ComplexNorm \
                and (len(kids) == 2 or type(kids[2]) is ApplyFilterbank):
            st = kids[0]
            x = realize(input)
            n_fft, hop, win_length, window = F.resolve_stft_args(x, st.fft_length, st.hop_length, st.win_length,
                                                                 st.window)
            _hip.check_stft_args(x.shape, n_fft, hop, win_length, st.center, st.pad_mode)
            args = (n_fft, hop, win_length, bool(st.center), st.pad_mode, bool(st.normalized), bool(st.onesided),
                    float(kids[1].power), False, 1.0, 1e-7)
            if len(kids) == 2:
                return F._call('spectrogram', x, window, *args)
            fb = kids[2].filterbank
            if fb.dim() == 2 and fb.shape[0] == (n_fft // 2 + 1 if st.onesided else n_fft) and fb.device == x.device:
                return F._call('melspectrogram', x, window, fb, *args)
        return realize(super(_FusedSequential, self).forward(input))


def Spectrogram(fft_length, hop_length=None, win_length=None,
                window=None, center=True, pad_mode='reflect',
                normalized=False, onesided=True, power=1.):
    """``Sequential(STFT(...), ComplexNorm(power))`` (reference layers.py:267-304); evaluated as one
    FFT kernel with the magnitude/power taken in its epilogue."""
    return _FusedSequential(
        STFT(fft_length, hop_length, win_length, window, center, pad_mode, normalized, onesided),
        ComplexNorm(power))


def Melspectrogram(num_mels=128, sample_rate=22050, min_freq=0.0, max_freq=None, num_freqs=None,
                   htk=False, mel_filterbank=None, **kwargs):
    """``Sequential(STFT, ComplexNorm(2.), ApplyFilterbank(mel))`` (reference layers.py:307-347).

    As in the reference, ``num_freqs`` is ignored and recomputed from ``kwargs['fft_length']``
    (1025 when absent), ``mel_filterbank`` may name a custom ``MelFilterbank``-like class, and the
    remaining ``kwargs`` go to ``Spectrogram`` (so omitting ``fft_length`` is a ``TypeError``)."""
    fft_length = kwargs.get('fft_length', None)
    num_freqs = fft_length // 2 + 1 if fft_length else 1025
    provider = MelFilterbank if mel_filterbank is None else mel_filterbank
    matrix = provider(num_mels=num_mels, sample_rate=sample_rate, min_freq=min_freq,
                      max_freq=max_freq, num_freqs=num_freqs, htk=htk).get_filterbank()
    return _FusedSequential(*Spectrogram(power=2., **kwargs), ApplyFilterbank(matrix))


class AmplitudeToDb(_ModuleNoStateBuffers):
    """``10·(log10(max(x², amin)) − log10(ref))`` (reference layers.py:350-381); fused into the
    producing kernel's epilogue when the input is a deferred spectrogram / mel-spectrogram."""

    def __init__(self, ref=1.0, amin=1e-7):
        super(AmplitudeToDb, self).__init__()
        self.ref = ref
        self.amin = amin
        assert ref > amin, "Reference value is expected to be bigger than amin, but I have" \
                           "ref:{} and amin:{}".format(ref, amin)

    def forward(self, x):
        if isinstance(x, DeferredSpectral) and x.pending() and x._stage in ('spec', 'mel'):
            # terminal stage: nothing can fuse behind the dB epilogue, so the fused kernel is launched now, on the
            # stream of the STFT call, and the caller gets an ordinary tensor
            return x.realize(db=(self.ref, self.amin))
        return F.amplitude_to_db(x, ref=self.ref, amin=self.amin)

    def __repr__(self):
        return self.__class__.__name__ + '(ref={}, amin={})'.format(self.ref, self.amin)


class DbToAmplitude(_ModuleNoStateBuffers):
    """Inverse of ``AmplitudeToDb`` (reference layers.py:384-412)."""

    def __init__(self, ref=1.0):
        super(DbToAmplitude, self).__init__()
        self.ref = ref

    def forward(self, x):
        return F.db_to_amplitude(x, ref=self.ref)

    def __repr__(self):
        return self.__class__.__name__ + '(ref={})'.format(self.ref)


class MuLawEncoding(_ModuleNoStateBuffers):
    """mu-law companding to int64 codes (reference layers.py:415-440)."""

    def __init__(self, n_quantize=256):
        super(MuLawEncoding, self).__init__()
        self.n_quantize = n_quantize

    def forward(self, x):
        return F.mu_law_encoding(x, self.n_quantize)

    def __repr__(self):
        return self.__class__.__name__ + '(n_quantize={})'.format(self.n_quantize)


class MuLawDecoding(_ModuleNoStateBuffers):
    """mu-law expansion; like the reference (layers.py:443-467) always decodes to the default dtype."""

    def __init__(self, n_quantize=256):
        super(MuLawDecoding, self).__init__()
        self.n_quantize = n_quantize

    def forward(self, x_mu):
        if torch.is_tensor(x_mu) and self.n_quantize == 256 and can_defer_codes(x_mu) \
                and torch.get_default_dtype() == torch.float32:
            # an STFT layer behind this one decodes the codes inside its frame load; anything else decodes now
            return DeferredWave(x_mu, self.n_quantize)
        return F.mu_law_decoding(x_mu, self.n_quantize)

    def __repr__(self):
        return self.__class__.__name__ + '(n_quantize={})'.format(self.n_quantize)


class HPSS(nn.Module):
    """Harmonic / percussive separation layer (reference beta_hpss.py:12-32): wraps ``functional.hpss``."""

    def __init__(self, kernel_size=31, power=2.0, hard=False, mask_only=False):
        super(HPSS, self).__init__()
        self.kernel_size = kernel_size
        self.power = power
        self.hard = hard
        self.mask_only = mask_only

    def forward(self, mag_specgrams):
        return F.hpss(mag_specgrams, self.kernel_size, self.power, self.hard, self.mask_only)

    def __repr__(self):
        return self.__class__.__name__ + '(kernel_size={}, power={}, hard={}, mask_only={})'.format(
            self.kernel_size, self.power, self.hard, self.mask_only)
