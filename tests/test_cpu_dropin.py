"""not-gpu: the reference's own CPU-only call sites run unchanged through the PRODUCT API on CPU tensors.

The reference's test-suite is CPU-only (reference tests/test_layers.py:1-3) and BASELINE configs[0] is a CPU
configuration, so "call sites drop in unchanged" includes CPU tensors and float64.  Each test restates one of the
reference's librosa-free checks (cited) and pins the values to the golden vectors captured from the unmodified
reference (tests/golden/make_golden.py): on a CPU tensor the ``tac_amd::*`` ops dispatch to the package's stock-torch
kernels (``_composite.py``), which keep the reference's operator order — so the agreement is to float rounding.
The same call sites on a HIP device are the subject of tests/test_gpu_parity.py.
"""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import signals

T = torch.from_numpy


@pytest.fixture(scope='module')
def tac():
    import torchaudio_contrib_amd as t
    return t


def test_cfg1_spectrogram_on_cpu_matches_golden(tac, golden):
    """BASELINE configs[0] / reference tests/test_layers.py:55-83: batch 4, 1 ch, 16 kHz, 1 s, fft 512 / hop 256."""
    g = golden('g1_cfg1')
    x = T(signals.audio_like((4, 1, 16000), seed=1))
    window = torch.hann_window(512)
    z = tac.stft(x, 512, hop_length=256, window=window)
    assert z.device.type == 'cpu' and tuple(z.shape) == (4, 1, 257, 63, 2)
    assert rel_err(z.numpy(), g['stft']) <= 1e-6
    assert rel_err(tac.complex_norm(z).numpy(), g['mag']) <= 1e-6
    model = torch.nn.Sequential(*tac.Spectrogram(512, hop_length=256, window=window, pad_mode='reflect'),
                                tac.AmplitudeToDb(ref=1.0, amin=1e-7))
    db = model(x)
    assert type(db) is torch.Tensor
    assert np.abs(db.numpy() - g['spec_db']).max() <= 1e-4


def test_stft_shapes_and_short_input(tac):
    """reference tests/test_functional.py:26-58: shape bookkeeping; an input too short to reflect-pad raises
    RuntimeError (the strict xfail at :31)."""
    for shape in [(1, 100000), (1, 2, 100000)]:
        x = torch.randn(*shape)
        for fft_len, hop in [(512, 256)]:
            z = tac.stft(x, fft_len, hop_length=hop)
            mag = tac.complex_norm(z)
            frames = (x.size(-1) + 2 * (fft_len // 2) - fft_len + hop) // hop
            assert z.dim() == x.dim() + 2 and mag.dim() == z.dim() - 1
            assert z.size(-2) == mag.size(-1) == frames and z.size(-3) == fft_len // 2 + 1 and z.size(-1) == 2
    with pytest.raises(RuntimeError):
        tac.stft(torch.randn(1, 10), 512, hop_length=256)


@pytest.mark.parametrize('shape', [(1, 2, 1025, 400, 2), (1025, 400, 2)])
@pytest.mark.parametrize('power', [1, 2, 0.7])
def test_complex_norm(tac, shape, power):
    """reference tests/test_functional.py:119-128."""
    z = torch.randn(*shape)
    want = torch.pow(z.pow(2.).sum(-1), power / 2.)
    assert torch.allclose(tac.complex_norm(z, power), want, atol=1e-5)


@pytest.mark.parametrize('shape', [(1, 257, 391), (1, 2, 257, 391)])
def test_apply_filterbank(tac, shape):
    """reference tests/test_functional.py:131-141."""
    spec = torch.randn(*shape)
    fb = torch.randn(shape[-2], 120)
    out = tac.apply_filterbank(spec, fb)
    assert out.size(-2) == 120 and out.dim() == spec.dim()
    assert torch.allclose(out, torch.einsum('...ft,fm->...mt', spec, fb), atol=1e-3)


def test_amplitude_db_known_answers_and_round_trip(tac, golden):
    """reference tests/test_functional.py:144-158."""
    amp = torch.tensor([1e-6, 1e-4, 0.1, 1.0, 10.0, 1e6]).sqrt()
    db = tac.amplitude_to_db(amp, ref=1.0, amin=1e-7)
    assert torch.allclose(db, torch.tensor([-60., -40., -10., 0., 10., 60.]), atol=1e-5)
    assert np.array_equal(db.numpy(), golden('g5_mulaw')['db_known_amp'])
    back = tac.amplitude_to_db(tac.db_to_amplitude(db, ref=1.0), ref=1.0)
    assert torch.allclose(back, db, atol=1e-5)
    g = golden('g5_mulaw')
    xa = T(signals.audio_like((4, 5000), seed=10))
    assert np.abs(tac.amplitude_to_db(xa, ref=2.0, amin=1e-5).numpy() - g['a2db_ref2']).max() <= 1e-5
    assert rel_err(tac.db_to_amplitude(xa * 40, ref=2.0).numpy(), g['db2a_ref2']) <= 1e-6


def test_mu_law_bit_exact_on_cpu(tac, golden):
    """reference tests/test_functional.py:161-203 — encoding vs the manual formula, decoding of FLOAT-typed codes
    vs the manual formula with torch.eq, and the round trip; plus the golden codes of the reference."""
    g = golden('g5_mulaw')
    for shape in [(1, 100000), (1, 2, 100000)]:
        x = 2 * (torch.randn(*shape) - 0.5)
        mu = torch.tensor(255, dtype=x.dtype)
        want = ((x.sign() * torch.log1p(mu * x.abs()) / torch.log1p(mu) + 1) / 2 * mu + 0.5).long()
        assert torch.equal(tac.mu_law_encoding(x, 256), want)
        codes = torch.randint(low=0, high=255, size=(1, 1024))
        cf = codes.float()
        y = (cf / mu) * 2 - 1.
        want = y.sign() * (torch.exp(y.abs() * torch.log1p(mu)) - 1.) / mu
        assert torch.equal(tac.mu_law_decoding(cf, 256), want)
        assert torch.equal(codes, tac.mu_law_encoding(tac.mu_law_decoding(codes, 256), 256))
    x2 = T(signals.uniform((1000000,), seed=8, scale=1.0))
    if np.array_equal(tac.mu_law_decoding(torch.arange(256), 256).numpy().view(np.uint32),
                      g['lut256'].view(np.uint32)):        # this host's libm reproduces the capture host's bits
        assert np.array_equal(tac.mu_law_encoding(x2, 256).numpy(), g['enc256_unit'].astype(np.int64))
    assert tac.MuLawEncoding()(x2).dtype == torch.int64 and tac.MuLawDecoding()(torch.arange(256)).dtype == torch.float32


def test_phase_vocoder_float64(tac, golden):
    """reference tests/test_functional.py:69-116 runs the vocoder in float64 (float32 drifts in the running sum);
    f64 in -> f64 out through the product API, checked against an independent float64 restatement, and the float32
    golden of the reference."""
    hop, rate = 256, 1.3
    z = torch.randn(1, 2, 1025, 400, 2, dtype=torch.float64)
    adv = torch.linspace(0, math.pi * hop, 1025, dtype=torch.float64)[..., None]
    out = tac.phase_vocoder(z, rate, adv)
    assert out.dtype == torch.float64 and tuple(out.shape) == (1, 2, 1025, int(math.ceil(400 / rate)), 2)
    # independent restatement with complex arithmetic
    zc = torch.view_as_complex(z)
    t = torch.arange(0, 400, rate, dtype=torch.float32).double()
    i0, frac = t.long(), t - t.floor()
    zp = torch.nn.functional.pad(zc, [0, 2])
    a, b = zp[..., i0], zp[..., i0 + 1]
    dphi = b.angle() - a.angle() - adv
    dphi = dphi - 2 * math.pi * torch.round(dphi / (2 * math.pi)) + adv
    acc = torch.cumsum(torch.cat([zc[..., :1].angle(), dphi[..., :-1]], -1), -1)
    want = torch.polar(frac * b.abs() + (1 - frac) * a.abs(), acc)
    assert rel_err(torch.view_as_complex(out.contiguous()).numpy(), want.numpy()) <= 1e-9
    g = golden('g7_phase_vocoder')
    z32 = T(signals.audio_like((2, 1, 65, 40, 2), seed=41))
    adv32 = torch.linspace(0, math.pi * 32, 65)[..., None]
    for r in (1.3, 0.7, 2.0):
        assert rel_err(tac.phase_vocoder(z32, r, adv32).numpy(), g['pv_rate%g' % r]) <= 1e-5


def test_melspectrogram_stretch_pipeline_and_float64(tac):
    """reference tests/test_layers.py:86-106 (STFT -> TimeStretch -> ComplexNorm -> ApplyFilterbank on CPU) and the
    f64 -> f64 contract of the whole chain."""
    fft_length, hop, num_mels, rate = 512, 256, 128, 0.7
    num_freqs = fft_length // 2 + 1
    fb = tac.MelFilterbank(num_freqs=num_freqs, num_mels=num_mels, max_freq=1.0).get_filterbank()
    model = torch.nn.Sequential(tac.STFT(fft_length, hop_length=hop),
                                tac.TimeStretch(hop_length=hop, num_freqs=num_freqs, fixed_rate=rate),
                                tac.ComplexNorm(power=2.0), tac.ApplyFilterbank(fb))
    for x in (torch.randn(1, 2, 100000), torch.randn(4, 100000)):
        out = model(x)
        frames = (x.size(-1) + 2 * (fft_length // 2) - fft_length + hop) // hop
        assert out.size(-2) == num_mels and out.size(-1) == math.ceil(frames / rate)
    x64 = torch.randn(2, 1, 8000, dtype=torch.float64)
    mel = tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=512, hop_length=128).double()
    y = torch.nn.Sequential(*mel, tac.AmplitudeToDb())(x64)
    assert y.dtype == torch.float64 and tuple(y.shape) == (2, 1, 40, 63)
    y32 = torch.nn.Sequential(*mel.float(), tac.AmplitudeToDb())(x64.float())
    assert (y.float() - y32).abs().max() < 1e-3


def test_g2_melspectrogram_on_cpu_matches_golden(tac, golden):
    g = golden('g2_cfg2_slice')
    x = T(signals.audio_like((2, 1, 160000), seed=2))
    mel = tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512)
    assert rel_err(mel(x).numpy(), g['mel']) <= 1e-6
    full = torch.nn.Sequential(*mel, tac.AmplitudeToDb())
    assert np.abs(full(x).numpy() - g['mel_db']).max() <= 1e-4


def test_autograd_through_the_ops_on_cpu(tac):
    """The reference is differentiable end to end through stock torch (functional.py:99-107, 126-128, 183-184,
    291-296): gradients through the tac_amd ops equal those of the same chain written directly with torch ops."""
    x = torch.randn(2, 1, 4000, dtype=torch.float64, requires_grad=True)
    mel = tac.Melspectrogram(num_mels=20, sample_rate=8000, fft_length=256, hop_length=64).double()
    full = torch.nn.Sequential(*mel, tac.AmplitudeToDb(amin=1e-3))
    g_weight = torch.randn(2, 1, 20, 63, dtype=torch.float64)
    for chain in (full, mel):
        (gx,) = torch.autograd.grad((chain(x) * g_weight).sum(), x)
        z = torch.stft(x.reshape(-1, 4000), 256, 64, window=mel[0].window,
                       return_complex=True)
        ref = torch.matmul((z.abs() ** 2).transpose(-1, -2), mel[2].filterbank).transpose(-1, -2).reshape(2, 1, 20, 63)
        if chain is full:
            ref = 10 * torch.log10(torch.clamp(ref ** 2, min=1e-3))
        (gr,) = torch.autograd.grad((ref * g_weight).sum(), x)
        assert rel_err(gx.numpy(), gr.numpy()) <= 1e-9
    assert torch.autograd.gradcheck(lambda t: tac.amplitude_to_db(t, 2.0, 1e-5), (torch.rand(5, 3, dtype=torch.float64)
                                                                                 .add(0.1).requires_grad_(True),))
    # a learnable filterbank gets its gradient too
    fb = torch.rand(129, 7, dtype=torch.float64, requires_grad=True)
    spec = tac.Spectrogram(256, 64, power=2.).double()(x.detach())
    (gfb,) = torch.autograd.grad(tac.apply_filterbank(spec, fb).sum(), fb)
    assert rel_err(gfb.numpy(), spec.sum(-1).sum((0, 1)).unsqueeze(1).expand(129, 7).numpy()) <= 1e-9


def test_hpss_on_cpu_matches_golden(tac, golden):
    """beta_hpss.py:35-127 through the product API on CPU tensors: values, the None pair of mask_only, bool hard masks, the
    layer and its repr."""
    from test_oracle_golden import HPSS_CASES, hpss_input
    g = golden('g8_hpss')
    mag = T(hpss_input())
    for k, power, hard in HPSS_CASES:
        res = tac.hpss(mag, k, power, hard)
        tag = 'k%d_p%g_%s' % (k, power, 'hard' if hard else 'soft')
        for name, r in zip(('harm', 'perc', 'mask_harm', 'mask_perc'), res):
            want = g[tag + '_' + name]
            assert r.dtype == (torch.bool if hard and name.startswith('mask') else torch.float32)
            assert np.abs(r.numpy().astype(np.float32) - want.astype(np.float32)).max() <= 1e-6 * max(1.0, np.abs(want).max())
    a, b, mh, mp = tac.HPSS(kernel_size=7, power=1.0, mask_only=True)(mag)
    assert a is None and b is None and np.array_equal(mh.numpy(), g['k7_p1_soft_mask_harm'])
    assert repr(tac.HPSS()) == 'HPSS(kernel_size=31, power=2.0, hard=False, mask_only=False)'
    with pytest.raises(TypeError):
        tac.hpss(mag, 3.0)
    with pytest.raises(RuntimeError):
        tac.hpss(mag[..., :10], 31)                      # reflect padding wider than the spectrogram


def test_coded_waveforms_on_cpu(tac, golden):
    """The step before the path through the product API on CPU tensors: mu-law codes -> MuLawDecoding ->
    Melspectrogram -> AmplitudeToDb against the reference's golden outputs, and int16 PCM (value = sample * 2^-15)."""
    g = golden('g9_mulaw_mel')
    codes = T(g['codes'].astype(np.int64))
    for n_fft, hop, mels in ((2048, 512, 128), (512, 128, 40)):
        model = torch.nn.Sequential(tac.MuLawDecoding(256),
                                    *tac.Melspectrogram(num_mels=mels, sample_rate=16000, fft_length=n_fft, hop_length=hop),
                                    tac.AmplitudeToDb())
        assert np.abs(model(codes).numpy() - g['mel_db_n%d' % n_fft]).max() <= 1e-4
    pcm = T((signals.audio_like((2, 1, 9000), seed=62) * 20000).astype(np.int16))
    mel = tac.Melspectrogram(num_mels=32, sample_rate=16000, fft_length=512, hop_length=128)
    assert torch.equal(mel(pcm), mel(pcm.float() * (1.0 / 32768.0)))
    assert tac.stft(pcm, 256).dtype == torch.float32


def test_double_backward_matches_the_reference_chain(tac):
    """The reference is stock torch operators, hence twice differentiable (a gradient penalty on d out / d waveform):
    ``create_graph=True`` through the tac_amd ops must give a gradient that is itself differentiable, with the values
    of the same chain written directly with torch operators (oracle/torch_ref.py)."""
    from oracle import torch_ref
    x = torch.randn(2, 1, 3000, dtype=torch.float64)

    def penalty(chain, t):
        t = t.clone().requires_grad_(True)
        (g,) = torch.autograd.grad(chain(t).sum(), t, create_graph=True)
        assert g.requires_grad
        (g ** 2).sum().backward()
        return g.detach(), t.grad

    fb = tac.create_mel_filter(129, 12, 0.0, 4000.0, False).double()
    win = torch.hann_window(256, dtype=torch.float64)
    g1, h1 = penalty(lambda t: tac.amplitude_to_db(tac.apply_filterbank(tac.complex_norm(tac.stft(t, 256, 64, window=win), 2.0), fb), 1.0, 1e-4), x)
    g2, h2 = penalty(lambda t: torch_ref.amplitude_to_db(torch_ref.apply_filterbank(
        torch_ref.complex_norm(torch_ref.stft(t, 256, 64, window=win), 2.0), fb), 1.0, 1e-4), x)
    assert rel_err(g1.numpy(), g2.numpy()) <= 1e-9 and rel_err(h1.numpy(), h2.numpy()) <= 1e-9
    assert float(h1.abs().max()) > 0
    # through the fused op of the factory pipeline as well
    mel = tac.Melspectrogram(num_mels=12, sample_rate=8000, fft_length=256, hop_length=64).double()
    g3, h3 = penalty(mel, x)
    g4, h4 = penalty(lambda t: torch_ref.apply_filterbank(torch_ref.complex_norm(torch_ref.stft(t, 256, 64, window=mel[0].window), 2.0), fb), x)
    assert rel_err(g3.numpy(), g4.numpy()) <= 1e-9 and rel_err(h3.numpy(), h4.numpy()) <= 1e-9


def test_default_window_survives_inference_mode(tac):
    """A default Hann window first built inside ``torch.inference_mode()`` must not poison a later training call
    ('Inference tensors cannot be saved for backward')."""
    from torchaudio_contrib_amd import functional as TF
    TF._window_cache.clear()
    x = torch.randn(1, 1, 2000)
    with torch.inference_mode():
        tac.stft(x, 200)
    y = tac.complex_norm(tac.stft(x.clone().requires_grad_(True), 200), 2.0)
    y.sum().backward()


def test_hpss_even_kernel_takes_the_first_n_windows(tac):
    """beta_hpss.py:84-91 with an even width: n + 1 windows fit the n + 2 (k // 2) padded positions and the reference's
    loops use the first n of them."""
    from oracle import torch_ref
    mag = torch.rand(2, 1, 20, 17)
    for k in (4, 6):
        got = tac.hpss(mag, k, 2.0, False)
        want = torch_ref.hpss(mag, k, 2.0, False)
        for a, b in zip(got, want):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_phase_vocoder_wants_the_reference_shape_of_phase_advance(tac):
    """functional.py:204-274 broadcasts ``phase_advance`` against (..., num_freqs, time'): (num_freqs, 1) works, a flat
    (num_freqs,) does not — on every route; a non-positive rate raises RuntimeError (torch.arange's)."""
    z = torch.randn(2, 9, 12, 2)
    adv = torch.linspace(0, math.pi * 4, 9)
    assert tuple(tac.phase_vocoder(z, 1.5, adv[:, None]).shape) == (2, 9, 8, 2)
    assert tuple(tac.phase_vocoder(z, 1.5, adv[None, :, None]).shape) == (2, 9, 8, 2)
    with pytest.raises(RuntimeError):
        tac.phase_vocoder(z, 1.5, adv)
    with pytest.raises(RuntimeError):
        tac.phase_vocoder(z, 0.0, adv[:, None])
