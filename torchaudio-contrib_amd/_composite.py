"""Stock-torch backend of every op in ``_ops.py`` — the kernels registered for the dispatch keys the gfx950
library does not serve.

Who ends up here (``_ops.py`` decides, never silently for a float32 tensor that lives on the GPU):

  * CPU tensors.  The reference computes where its input lives (its whole test-suite is CPU-only,
    reference ``tests/test_layers.py:1-3``; BASELINE configs[0] is a CPU configuration), so a CPU tensor is
    evaluated on the CPU by torch's own operators, in the reference's operator order so the values are the
    reference's.
  * float64 tensors on either device (the reference keeps f64 -> f64, e.g. its phase-vocoder test,
    reference ``tests/test_functional.py:69-116``); the gfx950 kernels compute in float32.
  * the derivative of an op that has no hand-written backward kernel (``_ops.py`` re-evaluates the op here
    under ``torch.enable_grad`` and differentiates that).

Nothing in this file is the product's fast path and nothing in it is test infrastructure either: the tests'
checker package is never imported here; it checks this file like it checks the kernels
(``tests/test_cpu_dropin.py``).
Each function takes fully resolved arguments (the public wrappers in ``functional.py`` fill in defaults and
validate) and cites the reference lines whose operator sequence it keeps.
"""
import math

import torch
import torch.nn.functional as TF

TWO_PI = 2.0 * math.pi


def stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
    """reference functional.py:89-111: fold the leading dims into the FFT batch, ``torch.stft``, unfold; the
    complex result is handed back as the trailing-2 real view the reference's era of torch produced."""
    if wave.dtype == torch.int16:                                # PCM: sample * 2^-15 (the package's convention)
        wave = wave.to(torch.float32) * (1.0 / 32768.0)
    batch_shape = wave.shape[:-1]
    rows = wave.reshape(-1, wave.shape[-1])
    z = torch.stft(rows, n_fft, hop_length=hop, win_length=win_length, window=window, center=center,
                   pad_mode=pad_mode, normalized=normalized, onesided=onesided, return_complex=True)
    pairs = torch.view_as_real(z)
    return pairs.reshape(batch_shape + pairs.shape[1:])


def complex_norm(z, power):
    """reference functional.py:126-128: 2-norm of the (re, im) pair; the exponent is a second pass."""
    length = z.norm(p=2, dim=-1)
    if power == 1.0:
        return length
    return length.pow(power)


def angle(z):
    """reference functional.py:187-191."""
    re, im = z.unbind(-1)
    return torch.atan2(im, re)


def magphase(z, power):
    """reference functional.py:194-201."""
    return complex_norm(z, power), angle(z)


def apply_filterbank(spec, bank):
    """reference functional.py:183-184: frames to the row axis, one matmul, back."""
    frames_first = spec.transpose(-1, -2)
    return (frames_first @ bank).transpose(-1, -2)


def amplitude_to_db(x, ref, amin):
    """reference functional.py:291-296: the input is squared, the square clamped, then 10·(log10 − log10 ref)."""
    floor_applied = (x ** 2.0).clamp(min=amin)
    ref_level = torch.log10(torch.tensor(ref, dtype=x.dtype, device=x.device))
    return 10.0 * (floor_applied.log10() - ref_level)


def db_to_amplitude(x, ref):
    """reference functional.py:312-314."""
    ref_level = torch.log10(torch.tensor(ref, dtype=x.dtype, device=x.device))
    return torch.pow(10.0, x / 10.0 + ref_level) ** 0.5


def spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref, amin):
    """reference layers.py:267-304 (+ :350-381 when db)."""
    out = complex_norm(stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided), power)
    return amplitude_to_db(out, ref, amin) if db else out


def melspectrogram(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db,
                   ref, amin):
    """reference layers.py:307-347 (+ :350-381 when db)."""
    out = apply_filterbank(
        complex_norm(stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided), power), bank)
    return amplitude_to_db(out, ref, amin) if db else out


def mu_law_encoding(x, n_quantize):
    """reference functional.py:329-335."""
    if not x.is_floating_point():
        x = x.to(torch.float)
    mu = torch.tensor(n_quantize - 1, dtype=x.dtype)            # a 0-dim host tensor, as in the reference
    squashed = torch.sign(x) * torch.log1p(mu * torch.abs(x)) / torch.log1p(mu)
    return ((squashed + 1) / 2 * mu + 0.5).to(torch.int64)


def mu_law_decoding(codes, n_quantize, dtype):
    """reference functional.py:348-354."""
    if not codes.is_floating_point():
        codes = codes.to(dtype)
    mu = torch.tensor(n_quantize - 1, dtype=codes.dtype)
    unit = codes / mu * 2 - 1.
    return torch.sign(unit) * (torch.exp(torch.abs(unit) * torch.log1p(mu)) - 1.) / mu


def phase_vocoder(spec, rate, phase_advance):
    """reference functional.py:233-274: pick the two frames around every fractional time step, interpolate the
    magnitudes, accumulate the wrapped phase increments."""
    n_frames = spec.shape[-2]
    t = torch.arange(0, n_frames, rate, device=spec.device)      # default dtype, as the reference evaluates it
    frac = t % 1.0
    phase0 = angle(spec[..., :1, :])
    tail_padded = TF.pad(spec, [0, 0, 0, 2])
    left = tail_padded.index_select(-2, t.long())
    right = tail_padded.index_select(-2, (t + 1).long())
    ang_l, ang_r = angle(left), angle(right)
    len_l, len_r = left.norm(dim=-1), right.norm(dim=-1)
    step = ang_r - ang_l - phase_advance
    step = step - TWO_PI * torch.round(step / TWO_PI)
    step = step + phase_advance
    step = torch.cat([phase0, step[..., :-1]], dim=-1)
    running = step.cumsum(-1)
    length = frac * len_r + (1 - frac) * len_l
    return torch.stack([length * running.cos(), length * running.sin()], dim=-1)


def hpss(mag, kernel_f, kernel_t, power, hard):
    """reference beta_hpss.py:104-127 without its Python loops: reflect-pad both axes, running medians along frequency
    (percussive) and time (harmonic) as ``unfold(...).median``, ``pow``, soft / hard masks.  Masks come back as floats
    (the wrapper turns hard masks into bool like the reference)."""
    shape = mag.shape
    x = mag.reshape((-1, 1) + tuple(shape[-2:]))
    hf, ht = kernel_f // 2, kernel_t // 2
    padded = TF.pad(x, (ht, ht, hf, hf), mode='reflect')
    n_freqs, n_frames = shape[-2], shape[-1]
    # (an even width leaves n + 1 windows over the n + 2 (k // 2) padded positions; the reference's loops take the
    # first n, beta_hpss.py:84-91)
    perc = padded[..., ht:ht + n_frames].unfold(-2, kernel_f, 1)[..., :n_freqs, :, :].median(dim=-1).values
    harm = padded[..., hf:hf + n_freqs, :].unfold(-1, kernel_t, 1)[..., :n_frames, :].median(dim=-1).values
    if power != 1.0:
        perc, harm = perc.pow(power), harm.pow(power)
    if hard:
        mask_h, mask_p = (harm > perc).to(mag.dtype), (harm < perc).to(mag.dtype)
    else:
        eps = 1e-6
        mask_h = (harm + eps) / (harm + perc + eps)
        mask_p = (perc + eps) / (harm + perc + eps)
    mask_h, mask_p = mask_h.reshape(shape), mask_p.reshape(shape)
    return mag * mask_h, mag * mask_p, mask_h, mask_p


def melspectrogram_mulaw(codes, window, bank, n_quantize, n_fft, hop, win_length, center, pad_mode, normalized, onesided,
                         power, db, ref, amin):
    """reference functional.py:338-354 followed by layers.py:307-381: decode, then the Melspectrogram chain."""
    wave = mu_law_decoding(codes, n_quantize, torch.float32)
    return melspectrogram(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db,
                          ref, amin)
