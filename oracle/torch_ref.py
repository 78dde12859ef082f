"""torch-CPU restatement of the reference hot path (TEST INFRASTRUCTURE — see oracle/__init__).

Every function states the reference lines it follows (paths relative to
``/root/reference/torchaudio_contrib``).  The arithmetic ORDER is the reference's, so
results are bit-identical to the shimmed reference on the same torch build; the code
itself is written fresh.
"""
import math

import torch

_LOG_6P4_OVER_27 = math.log(6.4) / 27.0
_SLANEY_HZ_PER_MEL = 200.0 / 3
_SLANEY_BREAK_HZ = 1000.0
_SLANEY_BREAK_MEL = _SLANEY_BREAK_HZ / _SLANEY_HZ_PER_MEL


# --------------------------------------------------------------------------- STFT
def stft(x, n_fft, hop=None, win_length=None, window=None, center=True,
         pad_mode='reflect', normalized=False, onesided=True):
    """functional.py:48-113 — flatten leading dims, periodic Hann default, torch.stft,
    return the legacy real view ``(*, C, F, T, 2)``."""
    lead = tuple(x.shape[:-1])
    flat = x.reshape(-1, x.shape[-1])
    if window is None:
        window = torch.hann_window(n_fft if win_length is None else win_length)
    spec = torch.stft(flat, n_fft=n_fft, hop_length=hop, win_length=win_length,
                      window=window, center=center, pad_mode=pad_mode,
                      normalized=normalized, onesided=onesided, return_complex=True)
    spec = torch.view_as_real(spec)
    return spec.reshape(lead + tuple(spec.shape[1:]))


def complex_norm(z, power=1.0):
    """functional.py:116-128 — L2 norm over the trailing (re, im) pair, then ``pow``."""
    mag = torch.norm(z, 2, -1)
    return mag if power == 1.0 else mag.pow(power)


def angle(z):
    """functional.py:187-191."""
    return torch.atan2(z[..., 1], z[..., 0])


def magphase(z, power=1.0):
    """functional.py:194-201."""
    return complex_norm(z, power), angle(z)


# --------------------------------------------------------------------------- mel
def hz_to_mel(hz, htk):
    """functional.py:26-45."""
    hz = torch.as_tensor(hz).to(torch.get_default_dtype())
    if htk:
        one = torch.tensor(1.0, dtype=torch.get_default_dtype())
        return 2595.0 * torch.log10(one + hz / 700.0)
    lin = (hz - 0.0) / _SLANEY_HZ_PER_MEL
    log = _SLANEY_BREAK_MEL + torch.log(hz / _SLANEY_BREAK_HZ) / _LOG_6P4_OVER_27
    return torch.where(hz >= _SLANEY_BREAK_HZ, log, lin)


def mel_to_hz(mel, htk):
    """functional.py:5-23."""
    mel = torch.as_tensor(mel).to(torch.get_default_dtype())
    if htk:
        return 700.0 * (10 ** (mel / 2595.0) - 1.0)
    lin = 0.0 + _SLANEY_HZ_PER_MEL * mel
    log = _SLANEY_BREAK_HZ * torch.exp(_LOG_6P4_OVER_27 * (mel - _SLANEY_BREAK_MEL))
    return torch.where(mel >= _SLANEY_BREAK_MEL, log, lin)


def create_mel_filter(num_freqs, num_mels, min_freq, max_freq, htk):
    """functional.py:131-169 — dense (num_freqs, num_mels) triangles, no area norm,
    bin grid = linspace(min_freq, max_freq, num_freqs)."""
    mel_lo, mel_hi = hz_to_mel(min_freq, htk), hz_to_mel(max_freq, htk)
    bins = torch.linspace(min_freq, max_freq, num_freqs)
    edges = mel_to_hz(torch.linspace(mel_lo, mel_hi, num_mels + 2), htk)
    widths = edges[1:] - edges[:-1]
    dist = edges.unsqueeze(0) - bins.unsqueeze(1)            # (F, M+2)
    falling = (-1.0 * dist[:, :-2]) / widths[:-1]
    rising = dist[:, 2:] / widths[1:]
    return torch.clamp(torch.min(falling, rising), min=0.0)


def apply_filterbank(spec, fb):
    """functional.py:172-184 — (…,F,T)·(F,M) → (…,M,T)."""
    return torch.matmul(spec.transpose(-2, -1), fb).transpose(-2, -1)


# --------------------------------------------------------------------------- dB
def amplitude_to_db(x, ref=1.0, amin=1e-7):
    """functional.py:277-296 — squares its input, clamps the square, 10·(log10 − log10 ref)."""
    sq = torch.clamp(x.pow(2.0), min=amin)
    ref_t = torch.tensor(ref, device=x.device, dtype=x.dtype)
    return 10.0 * (torch.log10(sq) - torch.log10(ref_t))


def db_to_amplitude(x, ref=1.0):
    """functional.py:299-314."""
    ref_t = torch.tensor(ref, device=x.device, dtype=x.dtype)
    return torch.pow(10.0, x / 10.0 + torch.log10(ref_t)).pow(0.5)


# --------------------------------------------------------------------------- mu-law
def mu_law_encoding(x, n_quantize=256):
    """functional.py:317-335 — int64 codes, ``.long()`` truncation toward zero."""
    if not x.dtype.is_floating_point:
        x = x.to(torch.float)
    mu = torch.tensor(n_quantize - 1, dtype=x.dtype)
    comp = x.sign() * torch.log1p(mu * x.abs()) / torch.log1p(mu)
    return ((comp + 1) / 2 * mu + 0.5).long()


def mu_law_decoding(codes, n_quantize=256, dtype=None):
    """functional.py:338-354."""
    if dtype is None:
        dtype = torch.get_default_dtype()
    if not codes.dtype.is_floating_point:
        codes = codes.to(dtype)
    mu = torch.tensor(n_quantize - 1, dtype=codes.dtype)
    y = (codes / mu) * 2 - 1.0
    return y.sign() * (torch.exp(y.abs() * torch.log1p(mu)) - 1.0) / mu


# --------------------------------------------------------------------------- phase vocoder
def phase_vocoder(spec, rate, phase_advance):
    """functional.py:204-274 — time-stretch of a (…, F, T, 2) complex spectrogram."""
    lead = [slice(None)] * (spec.dim() - 2)
    steps = torch.arange(0, spec.size(-2), rate, device=spec.device)
    frac = torch.remainder(steps, torch.tensor(1.0, device=spec.device))
    first_phase = angle(spec[tuple(lead + [slice(1)])])
    padded = torch.nn.functional.pad(spec, [0, 0, 0, 2])
    s0 = padded[tuple(lead + [steps.long()])]
    s1 = padded[tuple(lead + [(steps + 1).long()])]
    a0, a1 = angle(s0), angle(s1)
    n0, n1 = torch.norm(s0, dim=-1), torch.norm(s1, dim=-1)
    dphi = a1 - a0 - phase_advance
    dphi = dphi - 2 * math.pi * torch.round(dphi / (2 * math.pi))
    dphi = dphi + phase_advance
    dphi = torch.cat([first_phase, dphi[tuple(lead + [slice(-1)])]], dim=-1)
    acc = torch.cumsum(dphi, -1)
    mag = frac * n1 + (1 - frac) * n0
    return torch.stack([mag * torch.cos(acc), mag * torch.sin(acc)], dim=-1)


# --------------------------------------------------------------------------- hpss
def hpss(mag, kernel_size=31, power=2.0, hard=False):
    """beta_hpss.py:35-127 (int kernel_size) — column by column / row by row running medians of the reflect-padded
    spectrogram, as the reference loops; returns (harm spec, perc spec, harm mask, perc mask)."""
    k = int(kernel_size)
    half = k // 2
    padded = torch.nn.functional.pad(mag, (half, half, half, half), mode='reflect')
    n_freqs, n_frames = mag.shape[2], mag.shape[3]
    harm, perc = torch.empty_like(mag), torch.empty_like(mag)
    for f in range(n_freqs):
        perc[:, :, f, :] = padded[:, :, f:f + k, half:half + n_frames].median(dim=2)[0]
    for t in range(n_frames):
        harm[:, :, :, t] = padded[:, :, half:half + n_freqs, t:t + k].median(dim=3)[0]
    if power != 1.0:
        perc, harm = perc.pow(power), harm.pow(power)
    if hard:
        mh, mp = harm > perc, harm < perc
    else:
        mh, mp = (harm + 1e-6) / (harm + perc + 1e-6), (perc + 1e-6) / (harm + perc + 1e-6)
    return mag * mh, mag * mp, mh, mp


# --------------------------------------------------------------------------- pipelines
def spectrogram(x, n_fft, hop=None, win_length=None, window=None, center=True,
                pad_mode='reflect', normalized=False, onesided=True, power=1.0):
    """layers.py:267-304 — STFT → ComplexNorm(power)."""
    return complex_norm(stft(x, n_fft, hop, win_length, window, center, pad_mode,
                             normalized, onesided), power)


def melspectrogram(x, num_mels=128, sample_rate=22050, min_freq=0.0, max_freq=None,
                   htk=False, **stft_kw):
    """layers.py:307-347 — Spectrogram(power=2) → ApplyFilterbank(mel); ``num_freqs`` is
    always fft_length//2+1 (layers.py:330-331); max_freq defaults to sample_rate//2
    (layers.py:194)."""
    n_fft = stft_kw['n_fft']
    fb = create_mel_filter(n_fft // 2 + 1, num_mels, min_freq,
                           max_freq if max_freq else sample_rate // 2, htk)
    return apply_filterbank(spectrogram(x, power=2.0, **stft_kw), fb)


def melspectrogram_db(x, ref=1.0, amin=1e-7, **kw):
    """Sequential(*Melspectrogram(...), AmplitudeToDb(ref, amin)) — the benchmarked chain."""
    return amplitude_to_db(melspectrogram(x, **kw), ref, amin)
