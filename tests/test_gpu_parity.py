"""-m gpu: HIP path (through the C ABI) vs the committed golden vectors captured from the reference
and vs the live CPU oracle on the same seeded inputs.

Tolerances (north_star): <= 1e-4 relative fp32 for floating outputs — measured as max|a-b|/max|b|
per tensor for linear quantities and as absolute dB for dB quantities (the reference's own bar is
1e-2 dB, tests/test_layers.py:83) — and bit-exact for mu-law integer codes.

dB in the parameter sweeps: a dB value is compared only where the LINEAR reference value exceeds 1e-6 of its tensor's
maximum (the ``big`` masks below).  Every bin, masked or not, is first checked on the linear output (<= 2e-5 of the tensor's
maximum); below 1e-6 of the maximum that bound is wider than the value itself — the float32 FFT's own rounding (1e-7 of the
frame's largest bin) decides those digits, for the reference's CPU FFT as much as for these kernels — so their logarithm
is not comparable to 1e-3 dB.  The golden vectors g1 / g2 / g4 / g9 / g10 pin dB outputs UNMASKED on audio-like inputs.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import signals, torch_ref, numpy_ref

pytestmark = pytest.mark.gpu

REL = 1e-4        # north_star tolerance
TIGHT = 2e-6      # what fp32 kernels should really achieve on linear outputs
DB_ABS = 1e-3     # absolute dB


@pytest.fixture(scope='module')
def tac():
    import torchaudio_contrib_amd as t
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    t._native.lib()                       # fail loudly if the HIP library is missing
    t.set_strict(True)                    # nothing checked here may come from the stock-torch route
    t._hip.POISON_OUTPUTS = True          # gradient buffers start as NaN: a sample no kernel writes cannot pass by luck
    yield t
    t._hip.POISON_OUTPUTS = False
    t.set_strict(False)


def launches(tac_):
    """Snapshot of the per-entry-point launch counters of the C ABI (what really ran on the device)."""
    return dict(tac_._hip.launches)


def launched_since(tac_, before):
    now = tac_._hip.launches
    return {k: now[k] - before.get(k, 0) for k in now if now[k] != before.get(k, 0)}


@pytest.fixture(scope='module')
def mulaw_oracle_pinned(golden):
    """The reference's mu-law bits depend on the host's vectorised log1p/exp (Sleef via torch CPU).
    The live oracle is trusted on this host only if it reproduces the golden vectors captured from the
    reference; otherwise mu-law checks fall back to the golden vectors alone."""
    g = golden('g5_mulaw')
    x2 = torch.from_numpy(signals.uniform((1000000,), seed=8, scale=1.0))
    ok = np.array_equal(torch_ref.mu_law_encoding(x2, 256).numpy(), g['enc256_unit'].astype(np.int64))
    ok = ok and np.array_equal(torch_ref.mu_law_decoding(torch.arange(256), 256).numpy().view(np.uint32),
                               g['lut256'].view(np.uint32))
    return ok


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t_):
    return t_.detach().cpu().numpy()


# ------------------------------------------------------------------ golden: cfg-1 (tests/test_layers.py path)
def test_g1_stft_spectrogram_db(tac, golden):
    g = golden('g1_cfg1')
    x = dev(signals.audio_like((4, 1, 16000), seed=1))
    win = torch.hann_window(512).cuda()
    z = tac.stft(x, 512, hop_length=256, window=win)
    assert tuple(z.shape) == (4, 1, 257, 63, 2)
    assert rel_err(host(z), g['stft']) < TIGHT
    mag = tac.complex_norm(z, 1.0)
    assert rel_err(host(mag), g['mag']) < TIGHT
    seq = torch.nn.Sequential(*tac.Spectrogram(512, hop_length=256, window=win),
                              tac.AmplitudeToDb(ref=1.0, amin=1e-7)).cuda()
    before = launches(tac)
    out = seq(x)
    assert type(out) is torch.Tensor and launched_since(tac, before) == {'tac_spectrogram_f32': 1}
    assert np.abs(host(out) - g['spec_db']).max() < DB_ABS


# ------------------------------------------------------------------ golden: cfg-2 slice (the benchmarked chain)
def test_g2_melspectrogram_db(tac, golden):
    g = golden('g2_cfg2_slice')
    x = dev(signals.audio_like((2, 1, 160000), seed=2))
    mel = tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512).cuda()
    out = mel(x)
    assert tuple(out.shape) == (2, 1, 128, 313)
    assert rel_err(host(out), g['mel']) < 1e-5
    full = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
    before = launches(tac)
    y = full(x)
    # the user-owned Sequential of the unpacked chain is ONE kernel launch, issued by the terminal AmplitudeToDb,
    # and what comes back is an ordinary tensor (reference idiom, tests/test_layers.py:69)
    assert type(y) is torch.Tensor and launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
    assert np.abs(host(y) - g['mel_db']).max() < DB_ABS
    power = tac.Spectrogram(2048, hop_length=512, power=2.).cuda()(x)
    assert rel_err(host(power[..., [int(i) for i in g['frame_index']]]), g['power_frames']) < 1e-5
    # eager (unfused) evaluation must agree with the fused kernel
    tac.set_lazy_fusion(False)
    try:
        y2 = full(x)
        assert not isinstance(y2, tac.DeferredSpectral)
        assert np.abs(host(y2) - g['mel_db']).max() < DB_ABS
    finally:
        tac.set_lazy_fusion(True)


def test_g2_melspectrogram_on_the_matrix_pipe(tac, golden):
    """Round 6: the fused fft_length-2048 chain with the 1024-point transform as two chained 32 x 32 complex DFT products on
    v_mfma_f32_32x32x16_f16 (fp16 hi / lo operand pairs, csrc/melspec_mfma.hpp; tools/emulate_mfma_fft.py is its CPU emulation) —
    an opt-in route (slower on MI355X: tools/ablation/README.md, round 6), held to the same golden vectors as the default one:
    reference layers.py:307-381 through the unmodified reference, tests/golden/make_golden.py."""
    g = golden('g2_cfg2_slice')
    x = dev(signals.audio_like((2, 1, 160000), seed=2))
    mel = tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512).cuda()
    full = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
    valu = full(x)
    assert tac._native.lib().tac_last_route().decode().startswith('melspec_stream3_kernel')
    prev = tac.set_fft_pipe('mfma')
    try:
        out = tac.realize(mel(x))
        assert tac._native.lib().tac_last_route().decode().startswith('melspec_mfma_kernel')
        assert rel_err(host(out), g['mel']) < 1e-5
        y = full(x)
        assert np.abs(host(y) - g['mel_db']).max() < DB_ABS
        assert (y - valu).abs().max().item() < 1e-3
        # power 1 (the magnitude chain), another bank (40 bands: the general contraction loop), frames that touch the padding
        # on short rows, samples at PCM scale and tiny ones (the per-frame power-of-two scaling of the fp16 operands)
        fused = 0
        for scale in (1.0, 32768.0, 1e-12):
            xs = dev(signals.audio_like((3, 2, 9000), seed=5)) * scale
            for num_mels, power, htk in ((128, 1., False), (128, 2., False), (96, 2., False), (80, 2., True)):
                bank = tac.MelFilterbank(num_freqs=1025, num_mels=num_mels, sample_rate=16000, htk=htk).get_filterbank()
                m2 = torch.nn.Sequential(tac.STFT(2048, hop_length=512), tac.ComplexNorm(power), tac.ApplyFilterbank(bank)).cuda()
                before = launches(tac)
                got = tac.realize(m2(xs))
                # (a bank the one-band-per-lane layout rejects takes the three-phase kernel under either setting)
                streamed = launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
                took = tac._native.lib().tac_last_route().decode()
                tac.set_fft_pipe('valu')
                want = tac.realize(m2(xs))
                if streamed:
                    assert took.startswith('melspec_mfma_kernel'), took
                    assert tac._native.lib().tac_last_route().decode().startswith('melspec_stream3_kernel')
                    fused += 1
                tac.set_fft_pipe('mfma')
                assert rel_err(host(got), host(want)) < 2e-6, (scale, num_mels, power, htk)
        assert fused >= 6
    finally:
        tac.set_fft_pipe(prev)


STFT_CASES = {
    'n4096_h1024': dict(n=4096, kw=dict(hop_length=1024), full=True),
    'n512_h128_win400': dict(n=512, kw=dict(hop_length=128, win_length=400)),
    'n256_h64_normalized': dict(n=256, kw=dict(hop_length=64, normalized=True)),
    'n256_h100_twosided': dict(n=256, kw=dict(hop_length=100, onesided=False)),
    'n1024_h256_nocenter': dict(n=1024, kw=dict(hop_length=256, center=False)),
    'n512_h256_constant': dict(n=512, kw=dict(hop_length=256, pad_mode='constant')),
    'n512_h256_replicate': dict(n=512, kw=dict(hop_length=256, pad_mode='replicate')),
    'n512_h256_circular': dict(n=512, kw=dict(hop_length=256, pad_mode='circular')),
    'n1024_hdefault': dict(n=1024, kw=dict()),
    'n2048_h512': dict(n=2048, kw=dict(hop_length=512)),
    'n128_h32': dict(n=128, kw=dict(hop_length=32), length=2000),
    'n64_h16': dict(n=64, kw=dict(hop_length=16), length=1000),
}


@pytest.mark.parametrize('name', sorted(STFT_CASES))
def test_g4_stft_variants(tac, golden, name):
    g = golden('g4_variants')
    case = STFT_CASES[name]
    base = signals.audio_like((1, 2, 20000), seed=4)
    x = base if case.get('full') else base[..., :case.get('length', 6000)]
    z = tac.stft(dev(x), case['n'], **case['kw'])
    want = g[name]
    assert tuple(z.shape) == want.shape
    assert rel_err(host(z), want) < TIGHT


def test_g4_custom_window_leading_dims_power(tac, golden):
    g = golden('g4_variants')
    base = signals.audio_like((1, 2, 20000), seed=4)
    xs = dev(base[..., :6000])
    win = dev(np.abs(signals.uniform((512,), seed=44)) + np.float32(0.25))
    assert rel_err(host(tac.stft(xs, 512, hop_length=256, window=win)), g['n512_h256_customwin']) < TIGHT
    x4 = dev(signals.audio_like((2, 2, 2, 3000), seed=5))
    z = tac.stft(x4, 256, hop_length=64)
    assert tuple(z.shape) == g['lead3_n256_h64'].shape
    assert rel_err(host(z), g['lead3_n256_h64']) < TIGHT
    assert rel_err(host(tac.Spectrogram(512, hop_length=256, power=0.7).cuda()(xs)), g['spec_p07_n512']) < 1e-5
    assert rel_err(host(tac.Spectrogram(4096, hop_length=1024).cuda()(dev(base))), g['spec_p1_n4096']) < TIGHT


def test_g4_mel_variants(tac, golden):
    g = golden('g4_variants')
    xm = dev(signals.audio_like((3, 1, 30000), seed=6))
    mel = tac.Melspectrogram(num_mels=128, sample_rate=44100, fft_length=2048, hop_length=512).cuda()
    assert rel_err(host(mel(xm)), g['mel_sr44100']) < 1e-5
    melh = tac.Melspectrogram(num_mels=40, sample_rate=16000, min_freq=20.0, max_freq=7600.0, htk=True,
                              fft_length=512, hop_length=160, win_length=400).cuda()
    assert rel_err(host(melh(xm)), g['mel_htk40_n512']) < 1e-5
    chain = torch.nn.Sequential(*melh, tac.AmplitudeToDb(ref=2.0, amin=1e-5)).cuda()
    assert np.abs(host(chain(xm)) - g['mel_htk40_n512_db']).max() < DB_ABS


def test_fused_mel_parameter_sweep_vs_oracle(tac):
    """Seeded sweep over the fused Melspectrogram(+dB) chain — FFT size, hop, window length, band count, sample
    rate, HTK / Slaney scale, frequency range, batch shapes with ragged last tiles — against the oracle chain
    (layers.py:307-381).  Mel power within 2e-5 of the tensor maximum, dB within 1e-3 where the band is not at
    cancellation level."""
    rng = np.random.default_rng(7)
    for case in range(40):
        n = int(rng.choice([256, 400, 512, 1024, 2048]))
        hop = int(rng.choice([n // 4, n // 2, n // 8, int(rng.integers(1, n))]))
        win_length = n if case % 3 else int(rng.integers(n // 2, n + 1))
        num_mels = int(rng.choice([13, 40, 64, 80, 128]))
        sr = int(rng.choice([8000, 16000, 22050, 44100]))
        htk = bool(case % 2)
        min_freq = float(rng.choice([0.0, 20.0, 125.0]))
        max_freq = None if case % 4 else float(sr // 2 - int(rng.integers(0, sr // 8)))
        shape = (int(rng.integers(1, 4)), int(rng.integers(1, 3)), int(rng.integers(2 * n, 6 * n)) + case % 2)
        x = signals.uniform(shape, seed=300 + case)
        mel = tac.Melspectrogram(num_mels=num_mels, sample_rate=sr, min_freq=min_freq, max_freq=max_freq, htk=htk,
                                 fft_length=n, hop_length=hop, win_length=win_length).cuda()
        want = torch_ref.melspectrogram(torch.from_numpy(x), num_mels=num_mels, sample_rate=sr, min_freq=min_freq,
                                        max_freq=max_freq, htk=htk, n_fft=n, hop=hop, win_length=win_length)
        got = host(mel(dev(x)))
        tag = (case, n, hop, win_length, num_mels, sr, htk, min_freq, max_freq, shape)
        assert got.shape == tuple(want.shape), tag
        assert rel_err(got, want.numpy()) < 2e-5, tag
        chain = torch.nn.Sequential(*mel, tac.AmplitudeToDb(ref=1.0, amin=1e-7)).cuda()
        want_db = torch_ref.amplitude_to_db(want, ref=1.0, amin=1e-7).numpy()
        got_db = host(chain(dev(x)))
        big = want.numpy() > 1e-6 * want.numpy().max()
        assert np.abs(got_db - want_db)[big].max() < DB_ABS, tag


def test_non_power_of_two_and_large_n_fft(tac, golden):
    """fft_length outside the FFT kernels (powers of two, 400, even lengths with a 7-smooth half) runs as a windowed-DFT matrix
    product on the fp32 MFMA (reference: torch.stft accepts any n_fft, SURVEY §8 a-1 [probed] N=400)."""
    base = signals.audio_like((1, 2, 20000), seed=4)
    z = tac.stft(dev(base[..., :6000]), 400, hop_length=160)
    want = golden('g4_variants')['n400_h160']
    assert tuple(z.shape) == want.shape
    assert rel_err(host(z), want) < 5e-6
    x = signals.audio_like((2, 1, 30000), seed=31)
    for n, hop, kw in ((1001, 250, {}), (2018, 500, dict(win_length=1600)), (77, 30, dict(onesided=False)),
                       (5006, 1250, dict(center=False, normalized=True)), (16, 4, {})):   # odd, or a half with a prime factor above 7
        before = launches(tac)
        got = host(tac.stft(dev(x), n, hop_length=hop, **kw))
        assert launched_since(tac, before) == {'tac_apply_filterbank_f32': 1}, n
        ref = numpy_ref.stft(x, n, hop, **kw)
        assert got.shape[:-1] == ref.shape
        assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6
    spec = torch.nn.Sequential(*tac.Spectrogram(400, hop_length=160, power=2.), tac.AmplitudeToDb()).cuda()
    want_db = torch_ref.amplitude_to_db(torch_ref.spectrogram(torch.from_numpy(x), 400, 160, power=2.0)).numpy()
    assert np.abs(host(spec(dev(x))) - want_db).max() < DB_ABS
    mel = tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=400, hop_length=160).cuda()
    want_mel = torch_ref.melspectrogram(torch.from_numpy(x), num_mels=40, sample_rate=16000, n_fft=400, hop=160).numpy()
    assert rel_err(host(mel(dev(x))), want_mel) < 1e-5
    with pytest.raises(RuntimeError, match='strict mode'):
        tac.stft(dev(x), 10000, hop_length=2500)          # beyond every HIP path: loud under strict mode
    tac.set_strict(False)
    try:                                                  # otherwise torch's own GPU operators, with a warning
        tac._ops._warned.clear()
        with pytest.warns(tac.CompositeRouteWarning, match='fft_length 10000'):
            got = host(tac.stft(dev(x), 10000, hop_length=2500))
        ref = numpy_ref.stft(x, 10000, 2500)
        assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6
    finally:
        tac.set_strict(True)


def test_smooth_fft_lengths_generic_stockham_kernel(tac):
    """Even fft_lengths with a 7-smooth half that are not powers of two (480 / 960 / 1200 / 1920: 10 - 40 ms at 48 kHz, 882: 20 ms at
    44.1 kHz; csrc/stft_smooth.hip, round 5: radix 4 / 2 / 3 / 5 / 7 Stockham passes, `group` frames per workgroup) against the
    float64 restatement: complex rows one- and two-sided, pad modes, centre off, short windows, odd hops (scalar loads),
    normalisation, tiny and maximal sizes; |X|^p rows and dB; the Melspectrogram chain.  Strict mode is on."""
    x = signals.audio_like((3, 2, 20000), seed=78)
    cases = [(480, 120, {}), (882, 441, {}), (960, 240, dict(onesided=False)), (1200, 300, dict(center=False, normalized=True)),
             (1920, 480, dict(pad_mode='constant', win_length=1000)), (3000, 751, dict(pad_mode='replicate')),
             (6000, 1500, dict(win_length=4800)), (8100, 2025, dict(pad_mode='circular')), (12, 5, dict(onesided=False)),
             (14, 7, {}), (100, 30, dict(onesided=False)), (1000, 250, dict(center=False, normalized=True)), (7168, 1792, {}),
             (1536, 384, dict(win_length=3)), (2450, 613, {})]
    for n, hop, kw in cases:
        before = launches(tac)
        got = host(tac.stft(dev(x), n, hop_length=hop, **kw))
        assert launched_since(tac, before) == {'tac_stft_f32': 1}, (n, hop, kw)
        assert tac._native.lib().tac_last_route().decode() .startswith('stft_smooth_kernel<0, ')
        ref = numpy_ref.stft(x, n, hop, **kw)
        assert got.shape[:-1] == ref.shape
        assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6, (n, hop, kw)
    got = host(tac.stft(dev(x[0, :, :700]), 960, hop_length=240))          # rows shorter than a frame
    assert rel_err(got[..., 0] + 1j * got[..., 1], numpy_ref.stft(x[0, :, :700], 960, 240)) < 5e-6
    for n, hop in ((960, 240), (1200, 300), (882, 441), (6000, 1500)):
        mag2 = np.abs(numpy_ref.stft(x, n, hop)) ** 2
        for power in (2.0, 1.0, 0.7):
            before = launches(tac)
            got = host(tac.Spectrogram(n, hop, power=power).cuda()(dev(x)))
            assert launched_since(tac, before) == {'tac_spectrogram_f32': 1}
            assert rel_err(got, mag2 ** (power / 2)) < 1e-5, (n, power)
        chain = torch.nn.Sequential(*tac.Spectrogram(n, hop, power=2.), tac.AmplitudeToDb()).cuda()
        want_db = 10.0 * np.log10(np.maximum(mag2 ** 2, 1e-7))                 # (amplitude_to_db squares its input)
        big = mag2 > 1e-6 * mag2.max()
        assert np.abs(host(chain(dev(x))) - want_db)[big].max() < DB_ABS, n
        mel = tac.Melspectrogram(num_mels=40, sample_rate=48000, fft_length=n, hop_length=hop).cuda()
        want_mel = torch_ref.melspectrogram(torch.from_numpy(x), num_mels=40, sample_rate=48000, n_fft=n, hop=hop).numpy()
        assert rel_err(host(mel(dev(x))), want_mel) < 2e-5, n
    xg = dev(x).requires_grad_(True)                       # gradients keep the DFT-matrix adjoint: no stock-torch route
    routed = dict(tac._ops.composite_calls)
    (g1,) = torch.autograd.grad(tac.Spectrogram(960, 240, power=2.).cuda()(xg).sum(), xg)
    assert tac._ops.composite_calls == routed
    xr = torch.from_numpy(x).double().requires_grad_(True)
    (g0,) = torch.autograd.grad(torch_ref.spectrogram(xr, 960, 240, window=torch.hann_window(960, dtype=torch.float64), power=2.0).sum(), xr)
    assert rel_err(host(g1), g0.numpy()) < 1e-4


def test_fft_length_8192_to_32768_four_step_kernel(tac):
    """fft_length 8192 / 16384 / 32768 (csrc/stft_big.hip, round 5: one frame per workgroup, S = 4 / 8 / 16 wave-level 1024-point
    transforms, an S-point column DFT and the R2C split; reference functional.py:99-107 takes any fft_length) against the float64
    restatement (oracle/numpy_ref.py): complex rows one- and two-sided, every pad mode, centre off, short windows, hops that are
    not 16-byte aligned (the gathered load), normalisation; |X|^p rows with p = 2, 1, 0.7 and the dB epilogue; the Melspectrogram
    chain (two launches).  Strict mode is on: nothing here may touch torch's operators."""
    x = signals.audio_like((2, 1, 90000), seed=77)
    cases = [(8192, 2048, {}), (16384, 4096, {}), (32768, 8192, {}), (16384, 1000, dict(onesided=False)),
             (8192, 4099, dict(center=False, normalized=True)), (32768, 16384, dict(pad_mode='constant', win_length=30000)),
             (16384, 2050, dict(pad_mode='replicate', win_length=401)), (8192, 512, dict(pad_mode='circular', onesided=False)),
             (32768, 7001, dict(center=False))]
    for n, hop, kw in cases:
        before = launches(tac)
        got = host(tac.stft(dev(x), n, hop_length=hop, **kw))
        assert launched_since(tac, before) == {'tac_stft_f32': 1}, (n, hop, kw)
        assert tac._native.lib().tac_last_route().decode().startswith('stft_big_kernel<%d, 0, ' % (n // 2048))
        ref = numpy_ref.stft(x, n, hop, **kw)
        assert got.shape[:-1] == ref.shape
        assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6, (n, hop, kw)
    xs = x[0, :, :40000]                                   # a row shorter than the 32768 frame + padding on both sides
    got = host(tac.stft(dev(xs), 32768, hop_length=8192))
    assert rel_err(got[..., 0] + 1j * got[..., 1], numpy_ref.stft(xs, 32768, 8192)) < 5e-6
    for n, hop in ((8192, 2048), (16384, 4096), (32768, 8192)):
        mag2 = np.abs(numpy_ref.stft(x, n, hop)) ** 2
        for power in (2.0, 1.0, 0.7):
            before = launches(tac)
            got = host(tac.Spectrogram(n, hop, power=power).cuda()(dev(x)))
            assert launched_since(tac, before) == {'tac_spectrogram_f32': 1}
            assert rel_err(got, mag2 ** (power / 2)) < 1e-5, (n, power)
        chain = torch.nn.Sequential(*tac.Spectrogram(n, hop, power=2.), tac.AmplitudeToDb()).cuda()
        want_db = 10.0 * np.log10(np.maximum(mag2 ** 2, 1e-7))                 # (amplitude_to_db squares its input)
        big = mag2 > 1e-6 * mag2.max()
        assert np.abs(host(chain(dev(x))) - want_db)[big].max() < DB_ABS, n
        mel = tac.Melspectrogram(num_mels=64, sample_rate=44100, fft_length=n, hop_length=hop).cuda()
        want_mel = torch_ref.melspectrogram(torch.from_numpy(x), num_mels=64, sample_rate=44100, n_fft=n, hop=hop).numpy()
        assert rel_err(host(mel(dev(x))), want_mel) < 2e-5, n
    # gradients: 8192 through the generic Stockham adjoint (csrc/stft_smooth.hip: radix-4 / 2 passes, one frame per workgroup);
    # above that the op is differentiated through torch's operators (announced)
    xg = dev(x[:, :, :40000]).requires_grad_(True)
    before = launches(tac)
    (g1,) = torch.autograd.grad(tac.Spectrogram(8192, 2048, power=2.).cuda()(xg).sum(), xg)
    ran = launched_since(tac, before)
    assert ran.get('tac_stft_backward_f32') == 1 and 'tac_apply_filterbank_f32' not in ran, ran
    xr = torch.from_numpy(x[:, :, :40000]).double().requires_grad_(True)
    (g0,) = torch.autograd.grad(torch_ref.spectrogram(xr, 8192, 2048, window=torch.hann_window(8192, dtype=torch.float64), power=2.0).sum(), xr)
    assert rel_err(host(g1), g0.numpy()) < 1e-4
    with pytest.raises(RuntimeError, match='strict mode'):
        tac.stft(xg, 16384, hop_length=4096).square().sum().backward()


def test_rows_shorter_than_a_frame(tac):
    """Signals shorter than fft_length (centred: every frame touches the padding, the clamped whole-frame requests of the
    persistent kernels have nothing to read): complex stft, power dB and fused mel dB of every kernel family against the CPU
    route of the same modules (torch's CPU operators in the reference's operator order)."""
    torch.manual_seed(0)
    for n in (64, 256, 400, 512, 1024, 2048, 4096):
        for length in (n // 2 + 3, n - 1, n - 7):
            for rows in ((1, 1), (3, 2)):
                x = torch.rand(*rows, length) * 2 - 1
                z_want = tac.STFT(n, n // 4)(x)
                z = tac.realize(tac.STFT(n, n // 4).cuda()(x.cuda())).cpu()
                assert float((z - z_want).abs().max()) < 2e-6 * max(1.0, float(z_want.abs().max())) * 50, (n, length, rows)
                chains = [torch.nn.Sequential(*tac.Spectrogram(n, n // 4, power=2.), tac.AmplitudeToDb())]
                if n >= 256:
                    chains.append(torch.nn.Sequential(*tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=n, hop_length=n // 4),
                                                      tac.AmplitudeToDb()))
                for m in chains:
                    want = tac.realize(m(x))
                    got = tac.realize(m.cuda()(x.cuda())).cpu()
                    big = want > want.max() - 50.0                       # dB values within 50 dB of the peak (see the note in the test below)
                    assert float((got - want)[big].abs().max()) < DB_ABS, (n, length, rows, len(m))
                if n < 256 or length != n - 7:
                    continue
                # ... and their gradients (the backward kernels re-read the frames with the same clamped requests)
                for mk in (lambda: tac.Spectrogram(n, n // 4, power=2.),
                           lambda: tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=n, hop_length=n // 4)):
                    m = mk()
                    xc = x.clone().requires_grad_(True)
                    y = tac.realize(m(xc))
                    wgt = torch.rand_like(y)
                    (gw,) = torch.autograd.grad((y * wgt).sum(), xc)
                    xg = x.cuda().requires_grad_(True)
                    (gg,) = torch.autograd.grad((tac.realize(m.cuda()(xg)) * wgt.cuda()).sum(), xg)
                    assert float((gg.cpu() - gw).abs().max() / gw.abs().max()) < 1e-3, (n, length, rows)


def test_tiny_inputs_every_kernel_family(tac):
    """One frame, a handful of frames, fewer frames than waves, a single row: the persistent kernels' frame counters,
    clamped duplicate frames and sample-by-sample edge paths at their smallest sizes (every FFT size class; complex,
    |X|^2 dB and fused mel)."""
    for n in (64, 256, 400, 512, 1024, 2048, 4096):
        hop = n // 4
        for length, center in ((n, False), (n + hop, False), (n, True), (3 * n + 5, True)):
            for rows in ((1, 1), (3, 1)):
                x = signals.audio_like(rows + (length,), seed=n + length)
                got = host(tac.stft(dev(x), n, hop_length=hop, center=center))
                ref = numpy_ref.stft(x, n, hop, center=center)
                assert got.shape[:-1] == ref.shape, (n, length, center, rows)
                assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6, (n, length, center, rows)
                spec = host(torch.nn.Sequential(*tac.Spectrogram(n, hop, center=center, power=2.), tac.AmplitudeToDb()).cuda()(dev(x)))
                want = 10.0 * np.log10(np.maximum((np.abs(ref) ** 2) ** 2, 1e-7))
                # (amplitude_to_db squares the power: a bin 50 dB under the frame's peak carries ~3e-5 of relative fp32 FFT
                #  noise in |X|, 1.2e-4 in |X|^4, 5e-4 dB — bins further down are beyond DB_ABS for ANY float32 transform)
                big = np.abs(ref) ** 2 > 1e-5 * (np.abs(ref) ** 2).max()
                assert np.abs(spec - want)[big].max() < DB_ABS, (n, length, center, rows)
                if n >= 256:
                    mel = tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=n, hop_length=hop, center=center).cuda()
                    gm = host(tac.realize(mel(dev(x))))
                    wm = torch_ref.melspectrogram(torch.from_numpy(x), num_mels=40, sample_rate=16000, n_fft=n, hop=hop,
                                                  center=center).numpy()
                    assert gm.shape == wm.shape and rel_err(gm, wm) < 2e-5, (n, length, center, rows)


def test_short_input_raises_runtime_error(tac):
    # reference: strict-xfail RuntimeError for (1,100) with n_fft=512 reflect (tests/test_functional.py:31)
    with pytest.raises(RuntimeError):
        tac.stft(dev(signals.uniform((1, 100), seed=1)), 512, hop_length=256)
    with pytest.raises(RuntimeError):
        tac.STFT(512, hop_length=256).cuda()(dev(signals.uniform((1, 256), seed=1)))
    z = tac.stft(dev(signals.uniform((1, 257), seed=1)), 512, hop_length=256)     # L=257 is legal
    assert tuple(z.shape) == (1, 257, 2, 2)


# ------------------------------------------------------------------ live oracle comparisons
@pytest.mark.parametrize('shape,n,hop', [((1, 100000), 512, 256), ((1, 2, 100000), 512, 256),
                                         ((3, 2, 33333), 1024, 200), ((5, 1, 9000), 2048, 512)])
def test_stft_vs_oracles(tac, shape, n, hop):
    x = signals.audio_like(shape, seed=11)
    z = host(tac.stft(dev(x), n, hop_length=hop))
    want_t = torch_ref.stft(torch.from_numpy(x), n, hop).numpy()
    want_n = numpy_ref.stft(x, n, hop)
    assert z.shape == want_t.shape
    assert rel_err(z, want_t) < TIGHT
    assert rel_err(z[..., 0] + 1j * z[..., 1], want_n) < TIGHT
    mag, phase = tac.magphase(tac.stft(dev(x), n, hop_length=hop))
    assert rel_err(host(mag), np.abs(want_n)) < TIGHT


def test_stft_parameter_sweep_vs_oracle(tac):
    """Seeded sweep over the STFT argument space (size, hop parity, short windows, centring, every pad mode,
    normalisation, one/two-sided, odd lengths and row counts) — every kernel family and both load paths — against
    the torch-CPU restatement of functional.py:48-113."""
    rng = np.random.default_rng(2026)
    checked = 0
    for case in range(64):
        n = int(rng.choice([32, 64, 128, 256, 400, 512, 1024, 2048, 4096]))
        hop = int(rng.integers(1, n + 1)) if case % 3 else int(rng.choice([n // 4, n // 2, n // 8 or 1]))
        win_length = n if case % 4 else int(rng.integers(max(2, n // 4), n + 1))
        center = bool(case % 5)
        pad_mode = ['reflect', 'constant', 'replicate', 'circular'][case % 4]
        normalized = bool(case % 7 == 0)
        onesided = bool(case % 6)
        rows = (int(rng.integers(1, 4)), int(rng.integers(1, 3)))
        length = int(rng.integers(n + 1, 3 * n + 40)) + (case % 2)
        x = signals.audio_like(rows + (length,), seed=100 + case)
        window = None if case % 3 else torch.from_numpy(signals.uniform((win_length,), seed=200 + case) * 0.5 + 0.75)
        kw = dict(win_length=win_length, window=window, center=center, pad_mode=pad_mode, normalized=normalized,
                  onesided=onesided)
        want = torch_ref.stft(torch.from_numpy(x), n, hop, **kw).numpy()
        got = host(tac.stft(dev(x), n, hop_length=hop, win_length=win_length,
                            window=None if window is None else window.cuda(), center=center, pad_mode=pad_mode,
                            normalized=normalized, onesided=onesided))
        assert got.shape == want.shape, (case, n, hop, kw)
        assert rel_err(got, want) < 5e-6, (case, n, hop, win_length, center, pad_mode, normalized, onesided, length)
        checked += 1
    assert checked == 64


def test_hop_ring_rows_on_every_chunk_shape(tac):
    """The fft_length-2048 / hop-512 rows go through the LDS hop ring (csrc/stft_ring3.hpp): a loader wave, per-frame consumed
    marks, hop ids per audio row.  Shapes that stress its bookkeeping against the oracle — a single short row (every frame but a few
    touches the padding), rows of a handful of frames (a workgroup's chunk spans several rows: the ids jump, the ring's marks
    lap), many short rows, no centring (hops start at 0), a short window, and more frames per workgroup than ring slots and marks;
    complex rows, |X|^2 rows and dB rows; and the launch really is the ring kernel's where its conditions hold."""
    cases = [((1, 1, 4096), {}), ((1, 1, 2048 + 512 * 3), {}), ((7, 1, 5000), {}), ((3, 2, 9000), {'center': False}),
             ((40, 1, 6144), {}), ((2, 1, 70000), {'win_length': 1200}), ((1, 3, 40000), {'pad_mode': 'constant'}),
             ((5, 1, 2048 * 4 + 4), {})]
    for shape, kw in cases:
        x = signals.audio_like(shape, seed=sum(shape))
        want = torch_ref.stft(torch.from_numpy(x), 2048, 512, **kw)
        got = tac.stft(dev(x), 2048, 512, **kw)
        assert tac._native.lib().tac_last_route().startswith(b'stft_ring3_kernel<1024, 16, 0,'), (shape, kw)
        assert rel_err(host(got), want.numpy()) < 2e-6, (shape, kw)
        p_want = torch_ref.complex_norm(want, 2.0).numpy()
        spec = tac.Spectrogram(2048, 512, power=2., **kw).cuda()
        assert rel_err(host(spec(dev(x))), p_want) < 1e-5, (shape, kw)
        assert tac._native.lib().tac_last_route().startswith(b'stft_ring3_kernel<1024, 16, 1,'), (shape, kw)
        db = host(torch.nn.Sequential(tac.Spectrogram(2048, 512, power=2., **kw), tac.AmplitudeToDb()).cuda()(dev(x)))
        keep = p_want > 1e-6 * p_want.max()
        assert np.abs(db - torch_ref.amplitude_to_db(torch.from_numpy(p_want)).numpy())[keep].max() < DB_ABS, (shape, kw)
    # hop = fft_length / 8 takes the ring too (eight 1 KB hops per frame)
    for shape, kw in (((2, 1, 30000), {}), ((1, 2, 9000), {'center': False}), ((9, 1, 4100), {})):
        x = signals.audio_like(shape, seed=sum(shape) + 1)
        want = torch_ref.stft(torch.from_numpy(x), 2048, 256, **kw)
        assert rel_err(host(tac.stft(dev(x), 2048, 256, **kw)), want.numpy()) < 2e-6, (shape, kw)
        assert tac._native.lib().tac_last_route().startswith(b'stft_ring3_kernel<1024, 16, 0, 12, 8>'), (shape, kw)
        spec = tac.Spectrogram(2048, 256, power=1., **kw).cuda()
        assert rel_err(host(spec(dev(x))), torch_ref.complex_norm(want, 1.0).numpy()) < 1e-5, (shape, kw)
    # the same rows with a hop the ring does not take (and rows whose hops are not 16-byte aligned) still agree: the other kernel
    x = signals.audio_like((3, 1, 20000), seed=5)
    assert rel_err(host(tac.stft(dev(x), 2048, 500)), torch_ref.stft(torch.from_numpy(x), 2048, 500).numpy()) < 2e-6
    assert tac._native.lib().tac_last_route().startswith(b'stft_stream3_kernel<1024, 16, 0,')
    xo = dev(signals.audio_like((3, 1, 20001), seed=6))[..., 1:]                 # row starts off a 16-byte boundary
    assert rel_err(host(tac.stft(xo, 2048, 512)), torch_ref.stft(xo.cpu(), 2048, 512).numpy()) < 2e-6
    assert not tac._native.lib().tac_last_route().startswith(b'stft_ring3')


@pytest.mark.parametrize('power', [1, 2, 0.7])
@pytest.mark.parametrize('shape', [(1, 2, 1025, 400, 2), (1025, 400, 2)])
def test_complex_norm(tac, shape, power):
    z = signals.uniform(shape, seed=12, scale=4.0)
    got = host(tac.complex_norm(dev(z), power))
    want = (z.astype(np.float64) ** 2).sum(-1) ** (power / 2)
    assert np.abs(got - want).max() < 1e-5          # tests/test_functional.py:119-128 bar
    zt = dev(z).transpose(0, 1)                      # dense but permuted input keeps its layout
    got_t = tac.complex_norm(zt, power)
    assert got_t.shape == zt.shape[:-1]
    assert np.abs(host(got_t) - np.swapaxes(want, 0, 1)).max() < 1e-5


@pytest.mark.parametrize('new_len', [120, 36])
@pytest.mark.parametrize('shape', [(1, 257, 391), (1, 2, 257, 391), (257, 391), (3, 2, 2, 100, 70)])
def test_apply_filterbank_dense_random(tac, shape, new_len):
    spec = signals.uniform(shape, seed=13)
    fb = signals.uniform((shape[-2], new_len), seed=14)
    got = tac.apply_filterbank(dev(spec), dev(fb))
    assert got.shape[-1] == shape[-1] and got.shape[-2] == new_len and got.dim() == len(shape)
    want = np.einsum('...ft,fm->...mt', spec.astype(np.float64), fb.astype(np.float64))
    assert rel_err(host(got), want) < 1e-5
    want32 = torch_ref.apply_filterbank(torch.from_numpy(spec), torch.from_numpy(fb)).numpy()
    assert rel_err(host(got), want32) < 1e-5
    # frame-major strided view input (what Spectrogram hands over)
    sv = dev(np.ascontiguousarray(np.swapaxes(spec, -1, -2))).transpose(-1, -2)
    assert rel_err(host(tac.apply_filterbank(sv, dev(fb))), want) < 1e-5


def test_apply_filterbank_mel_sparse_plan(tac):
    fb = tac.create_mel_filter(1025, 128, 0.0, 8000, False)
    spec = np.abs(signals.uniform((2, 1025, 300), seed=15))
    got = host(tac.apply_filterbank(dev(spec), fb.cuda()))
    want = np.einsum('rft,fm->rmt', spec.astype(np.float64), fb.numpy().astype(np.float64))
    assert rel_err(got, want) < 1e-5


def test_apply_filterbank_streams_frame_major_spectrograms(tac):
    """apply_filterbank on the strided (…, F, T) views the spectrogram kernels return takes the band-sparse streaming
    kernels (the wave-autonomous lane-layout kernel where the bank fits it, the tile kernel otherwise — 13 bands, 2049
    bins): bin counts that are not multiples of four, ragged tiles, batches, odd band counts, and a sliced (non-packed)
    frame stride."""
    for n_fft, hop, n_mels, length, rows in ((2048, 512, 128, 30000, (3, 1)), (1024, 256, 40, 7777, (2, 2)),
                                              (512, 160, 13, 5000, (5,)), (2048, 512, 80, 160000, (4, 1)),
                                              (400, 160, 80, 9000, (2,)), (4096, 1024, 64, 40000, (2,)), (256, 64, 8, 3000, (3,))):
        x = signals.uniform(rows + (length,), seed=91 + n_mels)
        spec = tac.Spectrogram(n_fft, hop, power=2.).cuda()(dev(x))
        assert spec.stride(-2) == 1                                   # frame-major view
        fb = tac.create_mel_filter(n_fft // 2 + 1, n_mels, 0.0, 8000.0, bool(n_mels % 2)).cuda()
        got = host(tac.apply_filterbank(spec, fb))
        want = np.einsum('...ft,fm->...mt', host(spec).astype(np.float64), fb.cpu().numpy().astype(np.float64))
        assert got.shape == want.shape and rel_err(got, want) < 1e-5, (n_fft, n_mels)
        sliced = spec[..., 1:-2]                                      # frame stride != F*... still frame-major rows
        got2 = host(tac.apply_filterbank(sliced, fb))
        assert rel_err(got2, want[..., 1:-2]) < 1e-5, (n_fft, n_mels)


@pytest.mark.timeout(120)
def test_melbank_pack_bank_aware_placement_is_polynomial(tac):
    """tac_melbank_pack places the band starts of every sixteen-lane LDS group by bipartite matching (lane -> bank residue,
    smallest load per residue).  A slot in which fifteen narrow bands share ONE start next to a twelve-step band gives every
    lane the same twelve candidate residues: no matching exists below two lanes per residue, and refusing the smaller cap
    must not explore every eviction order (the visited set is shared by the whole augmenting search: Kuhn's algorithm).
    Results are checked against float64 — placement only moves zero-weight taps."""
    import time
    n_fft, hop, f_bins = 2048, 512, 1025
    fb = np.zeros((f_bins, 64), dtype=np.float32)
    rng = np.random.default_rng(3)
    for m in range(64):
        if m % 16 == 0:
            fb[300:348, m] = rng.random(48).astype(np.float32) + 0.1       # 12 four-tap steps: sets the slot's length
        else:
            fb[200:204, m] = rng.random(4).astype(np.float32) + 0.1        # one step, eleven steps of slack, all lanes alike
    x = signals.audio_like((2, 1, 12000), seed=43)
    chain = torch.nn.Sequential(tac.STFT(n_fft, hop), tac.ComplexNorm(2.0), tac.ApplyFilterbank(torch.from_numpy(fb))).cuda()
    before = launches(tac)
    t0 = time.perf_counter()
    y = tac.realize(chain(dev(x)))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 30.0
    assert launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
    p = np.abs(numpy_ref.stft(x, n_fft, hop)) ** 2
    assert rel_err(host(y), np.einsum('...ft,fm->...mt', p, fb.astype(np.float64))) < 1e-5


@pytest.mark.parametrize('n_fft', [1024, 400, 512, 2048, 4096])
@pytest.mark.parametrize('path', ['sparse', 'mfma'])
def test_fused_kernels_on_custom_filterbanks(tac, path, n_fft, monkeypatch):
    """Both fused contraction forms against float64 on banks that stress the packing: a band with no support, a band
    with interior zeros, n_mels not a multiple of 16, weights reaching the last bin; plus a dense random bank, which
    the band-sparse form rejects (falls back to the MFMA form or to spectrogram + MFMA GEMM kernels) — all through the layer chain."""
    monkeypatch.setattr(tac._hip, 'MEL_PATH', path)
    fused = 'tac_melspec_sparse_f32' if path == 'sparse' else 'tac_melspec_f32' 
    x = signals.audio_like((3, 2, 20000), seed=41)
    hop, f_bins = n_fft // 4, n_fft // 2 + 1
    rng = np.random.default_rng(7)
    fb = np.zeros((f_bins, 50), dtype=np.float32)
    for m in range(50):
        lo = int(rng.integers(0, f_bins - 40))
        ln = int(rng.integers(1, 40))
        fb[lo:lo + ln, m] = rng.random(ln).astype(np.float32) + 0.1
    fb[:, 7] = 0.0                                   # empty band
    if n_fft != 1024:                                # (1024 keeps its two bands that span most of the spectrum: the
        fb[:, 9] = 0.0                               # lane-layout form rejects them and the three-phase form takes over)
        fb[:, 49] = 0.0
    fb[100:140, 9] = 1.0
    fb[110:130, 9] = 0.0                             # interior zeros inside the support
    fb[f_bins - 5:, 49] = 2.0                        # support touching the Nyquist bin
    chain = torch.nn.Sequential(tac.STFT(n_fft, hop), tac.ComplexNorm(2.0), tac.ApplyFilterbank(torch.from_numpy(fb)),
                                tac.AmplitudeToDb()).cuda()
    before = launches(tac)
    y = chain(dev(x))
    if not (path == 'mfma' and n_fft in (400, 2048, 4096)):                    # (the MFMA fused form covers fft_length <= 1024, powers of two)
        assert launched_since(tac, before) == {fused: 1}                       # really fused
    p = np.abs(numpy_ref.stft(x, n_fft, hop)) ** 2
    mel = np.einsum('...ft,fm->...mt', p, fb.astype(np.float64))
    want = 10.0 * np.log10(np.maximum(mel ** 2, 1e-7))
    assert np.abs(host(y) - want).max() < DB_ABS
    lin = torch.nn.Sequential(*list(chain)[:3])(dev(x))
    assert rel_err(host(lin), mel) < 1e-5
    # magnitude (power = 1) chain takes the same kernels
    chain1 = torch.nn.Sequential(tac.STFT(n_fft, hop), tac.ComplexNorm(1.0), tac.ApplyFilterbank(torch.from_numpy(fb))).cuda()
    assert rel_err(host(chain1(dev(x))), np.einsum('...ft,fm->...mt', np.sqrt(p), fb.astype(np.float64))) < 1e-5
    # dense random bank: not fusable, still exact
    dense = signals.uniform((f_bins, 24), seed=42)
    chain2 = torch.nn.Sequential(tac.STFT(n_fft, hop), tac.ComplexNorm(2.0), tac.ApplyFilterbank(torch.from_numpy(dense))).cuda()
    before = launches(tac)
    y2 = tac.realize(chain2(dev(x)))
    if path == 'sparse' and n_fft == 1024:           # (a small dense bank still fits the MFMA form's step budget)
        assert launched_since(tac, before) == {'tac_spectrogram_f32': 1, 'tac_apply_filterbank_f32': 1}
    assert rel_err(host(y2), np.einsum('...ft,fm->...mt', p, dense.astype(np.float64))) < 1e-5


def test_filterbank_buffer_replaced_in_place_is_repacked(tac):
    """The pack / plan ride on the filterbank tensor and follow in-place edits (no stale cache)."""
    x = dev(signals.audio_like((2, 1, 9000), seed=43))
    mel = tac.Melspectrogram(num_mels=32, sample_rate=16000, fft_length=512, hop_length=128).cuda()
    y1 = host(mel(x))
    mel[2].filterbank.mul_(2.0)
    y2 = host(mel(x))
    assert rel_err(y2, 2.0 * y1) < 1e-6
    mel[2].filterbank[:, :16] = 0.0
    y3 = host(mel(x))
    assert np.abs(y3[:, :, :16]).max() == 0.0 and rel_err(y3[:, :, 16:], y2[:, :, 16:]) < 1e-6


def test_route_cache_survives_object_id_reuse(tac):
    """The fused-kernel route is remembered per filterbank OBJECT: a new tensor that happens to get the ``id`` of a
    collected one (CPython reuses addresses) must be routed on its own contents.  Forced here by planting an entry under
    the new tensor's id that claims the band-sparse form for a dense bank."""
    import weakref
    x = dev(signals.audio_like((2, 1, 9000), seed=44))
    window = torch.hann_window(512, device='cuda')
    dense = dev(signals.uniform((257, 24), seed=45))
    other = torch.zeros(1)
    g = tac._hip.geometry(x, 512, 128, 512, True, 'reflect', False, True)
    g.routes[(id(dense), dense._version, 2.0, tac._hip.MEL_PATH)] = (weakref.ref(other), 'sparse')
    y = tac._hip.melspectrogram(x, window, dense, 512, 128, 512, True, 'reflect', False, True, 2.0, False, 1.0, 1e-10)
    p = np.abs(numpy_ref.stft(host(x), 512, 128)) ** 2
    assert rel_err(host(y), np.einsum('...ft,fm->...mt', p, host(dense).astype(np.float64))) < 1e-5
    for _ in range(20):                              # and the natural form of it: banks created and dropped in a loop
        sparse_mod = tac.Melspectrogram(num_mels=32, sample_rate=16000, fft_length=512, hop_length=128).cuda()
        a = host(sparse_mod(x))
        del sparse_mod
        fb = dev(signals.uniform((257, 32), seed=46))
        b = host(tac.realize(tac.apply_filterbank(tac.complex_norm(tac.stft(x, 512, 128, window=window), 2.0), fb)))
        assert rel_err(b, np.einsum('...ft,fm->...mt', p, host(fb).astype(np.float64))) < 1e-5
        del fb
    assert np.isfinite(a).all()


def test_amplitude_db_known_answers(tac, golden):
    amp = torch.tensor([0.000001, 0.0001, 0.1, 1.0, 10.0, 1000000.0]).sqrt().cuda()
    db = torch.tensor([-60.0, -40.0, -10.0, 0.0, 10.0, 60.0])
    got = tac.amplitude_to_db(amp, ref=1.0).cpu()
    assert (got - db).abs().max() < 1e-5
    assert (tac.db_to_amplitude(db.cuda(), ref=1.0).cpu() - amp.cpu()).abs().div(amp.cpu()).max() < 1e-6
    back = tac.amplitude_to_db(tac.db_to_amplitude(db.cuda()))
    assert (back.cpu() - db).abs().max() < 1e-5
    g = golden('g5_mulaw')
    xa = dev(signals.audio_like((4, 5000), seed=10))
    assert np.abs(host(tac.amplitude_to_db(xa, ref=2.0, amin=1e-5)) - g['a2db_ref2']).max() < 1e-4
    assert rel_err(host(tac.db_to_amplitude(xa * 40, ref=2.0)), g['db2a_ref2']) < 1e-5


def test_pipelined_kernel_epilogues_n2048(tac):
    """n_fft = 2048 takes the software-pipelined kernel for complex rows and for |X| / |X|^2 rows with or without
    the fused dB epilogue; 1025 floats per power row exercises every 16-byte row misalignment, 7 frames per row
    the reflect-padded edge frames, 23 rows a ragged tail of the persistent grid."""
    x = signals.uniform((23, 1, 3072), seed=21)            # flat spectrum: no bins at fp32 cancellation level for the dB check
    xt = dev(x)
    want_c = torch_ref.stft(torch.from_numpy(x), 2048, 512).numpy()
    got_c = host(tac.realize(tac.STFT(2048, 512).cuda()(xt)))
    assert got_c.shape == want_c.shape and rel_err(got_c, want_c) < TIGHT
    for power in (1.0, 2.0):
        want = torch_ref.spectrogram(torch.from_numpy(x), 2048, 512, power=power)
        got = host(tac.Spectrogram(2048, 512, power=power).cuda()(xt))
        assert rel_err(got, want.numpy()) < 1e-5, power
        want_db = torch_ref.amplitude_to_db(want, ref=1.0, amin=1e-7).numpy()
        chain = torch.nn.Sequential(*tac.Spectrogram(2048, 512, power=power), tac.AmplitudeToDb()).cuda()
        got_db = host(chain(xt))
        # bins five orders of magnitude below the frame's level are fp32 cancellation noise in the oracle as well:
        # the 1e-3 dB bar applies to the rest, a loose one to those few
        mag = want.numpy() ** (1.0 / power)
        big = mag > 1e-3 * mag.max()
        assert np.abs(got_db - want_db)[big].max() < DB_ABS, power
        assert np.abs(got_db - want_db).max() < 0.1 and (~big).mean() < 1e-3, power
    # normalized=True scales inside the kernel; an unaligned row stride (odd length) takes the 8-byte load path
    x2 = signals.audio_like((5, 2, 4099), seed=22)
    want = torch_ref.spectrogram(torch.from_numpy(x2), 2048, 512, power=2.0, normalized=True).numpy()
    got = host(tac.Spectrogram(2048, 512, power=2.0, normalized=True).cuda()(dev(x2)))
    assert rel_err(got, want) < 1e-5


def test_g6_angle_magphase(tac, golden):
    """angle / magphase: one streaming kernel, golden values from the reference."""
    g = golden('g6_magphase')
    z = signals.audio_like((3, 65, 11, 2), seed=31)
    z[0, 0, :4] = np.array([[0.0, 0.0], [1.0, 0.0], [-1.0, 0.0], [0.0, -2.0]], np.float32)
    zt = dev(z)
    assert np.abs(host(tac.angle(zt)) - g['angle']).max() < 2e-6
    for p in (1.0, 2.0, 0.5):
        m, ph = tac.magphase(zt, power=p)
        assert rel_err(host(m), g['mag_p%g' % p]) < 1e-6
        assert np.abs(host(ph) - g['phase_p%g' % p]).max() < 2e-6
    # strided input (the STFT's (…, F, T, 2) view) and an odd element count (scalar tail of the vector kernel)
    spec = tac.realize(tac.STFT(512, 128).cuda()(dev(signals.audio_like((2, 1, 4000), seed=32))))
    m, ph = tac.magphase(spec, power=1.0)
    want = torch_ref.magphase(spec.cpu(), 1.0)
    assert m.shape == spec.shape[:-1] and rel_err(host(m), want[0].numpy()) < 1e-6
    big = want[0].numpy() > 1e-3 * want[0].numpy().max()          # phase of a cancelling bin is noise in both
    assert np.abs(host(ph) - want[1].numpy())[big].max() < 1e-5
    odd = dev(signals.audio_like((7, 2), seed=33))
    assert np.abs(host(tac.angle(odd)) - torch_ref.angle(odd.cpu()).numpy()).max() < 2e-6


def test_g7_phase_vocoder_and_time_stretch(tac, golden):
    """phase_vocoder against the reference's golden outputs, then the reference's own layer
    chain STFT -> TimeStretch -> ComplexNorm (tests/test_layers.py:98-101) against the oracle."""
    import math
    g = golden('g7_phase_vocoder')
    z = signals.audio_like((2, 1, 65, 40, 2), seed=41)
    adv = torch.linspace(0, math.pi * 32, 65)[..., None]
    for rate in (1.3, 0.7, 2.0):
        got = tac.phase_vocoder(dev(z), rate, adv.cuda())
        want = g['pv_rate%g' % rate]
        assert tuple(got.shape) == want.shape
        got = host(got)
        # (1) The kernel carries exp(i phase) as a unit phasor advanced by complex products (csrc/phase_vocoder.hip: modulo one
        # turn the reference's wrapped step is a1 - a0), so it never forms the ill-conditioned float32 sum and agrees with
        # the float64 evaluation of the reference's formula on the same float32 inputs to ~1e-6 ...
        want64 = torch_ref.phase_vocoder(torch.from_numpy(z).double(), rate, adv.double()).numpy()
        assert rel_err(got, want64) < 1e-5, rate
        # (2) ... while the golden is the reference's FLOAT32 evaluation, whose own rounding is what separates the two
        # (the reference notes it, tests/test_functional.py:85-88).  Worst-case bound of that rounding per (bin, frame):
        # every step rounds `a1 - a0 - pa`, the wrapped value and `+ pa` at magnitude |pa|+2pi (3 half-spacings),
        # multiplies float32(2 pi) (1.75e-7 off) by up to |pa|/2pi + 1 turns, and rounds the running sum at its own
        # magnitude |acc_j| <= (j+1)(|pa|+pi) + pi; our own error adds 3e-5 rad.  Magnitudes are plain interpolation.
        pa = np.abs(adv.numpy().ravel().astype(np.float64))
        j = np.arange(want.shape[-2])
        step = 1.5 * np.spacing((pa + 2 * np.pi).astype(np.float32)).astype(np.float64) + 1.75e-7 * (pa / (2 * np.pi) + 1)
        acc_mag = ((j[None, :] + 1) * (pa[:, None] + np.pi) + np.pi).astype(np.float32)
        phase_tol = np.cumsum(step[:, None] + 0.5 * np.spacing(acc_mag).astype(np.float64), axis=1) + 3e-5
        mag = np.hypot(want[..., 0], want[..., 1])
        assert np.abs(np.hypot(got[..., 0], got[..., 1]) - mag).max() < 2e-6, rate
        err = np.hypot(got[..., 0] - want[..., 0], got[..., 1] - want[..., 1])
        assert (err <= phase_tol[None, None] * mag + 2e-6 * mag.max()).all(), rate
        assert err[:, :, :4].max() < 1e-4 * mag.max(), rate          # low bins: north_star's 1e-4 even against float32
    x = signals.audio_like((3, 2, 6000), seed=42)
    hop, n_fft = 128, 512
    chain = torch.nn.Sequential(tac.STFT(n_fft, hop), tac.TimeStretch(hop, n_fft // 2 + 1, fixed_rate=1.3),
                                tac.ComplexNorm(power=2.)).cuda()
    got = host(chain(dev(x)))
    spec = torch_ref.stft(torch.from_numpy(x), n_fft, hop)
    want = torch_ref.complex_norm(torch_ref.phase_vocoder(spec, 1.3, torch.linspace(0, math.pi * hop, n_fft // 2 + 1)[..., None]), 2.0)
    assert got.shape == tuple(want.shape) and got.shape[-1] == math.ceil(spec.shape[-2] / 1.3)
    assert rel_err(got, want.numpy()) < 1e-5
    # rate 1 is the identity in the layer (layers.py:254-255); a missing rate raises like the reference
    stft = tac.realize(tac.STFT(n_fft, hop).cuda()(dev(x)))
    ts = tac.TimeStretch(hop, n_fft // 2 + 1).cuda()
    assert torch.equal(ts(stft, 1.0), stft)
    with pytest.raises(ValueError):
        ts(stft)


def test_phase_vocoder_infinite_input_poisons_only_its_frames(tac):
    """An infinite component has a finite angle in the reference (atan2: functional.py:221-229) and an infinite magnitude: only the
    output frames interpolated from it are non-finite, every later frame of the bin is ordinary.  The float32 kernel carries the
    running phase as a unit phasor — the value must not turn that phasor into NaN (round-4 advisor finding)."""
    import math
    z = signals.audio_like((1, 1, 33, 30, 2), seed=77)
    z[0, 0, 5, 7, 0] = np.inf                       # +inf real part, finite imaginary part: angle 0
    z[0, 0, 9, 11, 1] = -np.inf                     # -inf imaginary part: angle -pi/2
    z[0, 0, 12, 3, :] = (-np.inf, np.inf)           # both: angle 3 pi / 4
    adv = torch.linspace(0, math.pi * 16, 33)[..., None]
    for rate in (1.3, 0.6):
        got = host(tac.phase_vocoder(dev(z), rate, adv.cuda()))
        want = torch_ref.phase_vocoder(torch.from_numpy(z).double(), rate, adv.double()).numpy()
        fin = np.isfinite(want).all(-1)
        assert (np.isfinite(got).all(-1) == fin).all(), 'the same (bin, frame) positions are non-finite'
        assert not fin.all() and fin.mean() > 0.95
        assert np.abs(got[fin] - want[fin]).max() < 1e-5 * np.abs(want[fin]).max()


def test_non_finite_samples_poison_the_same_frames(tac):
    """A NaN / Inf sample makes every bin of the frames that contain it NaN in the reference (the FFT mixes it
    into all of them); the kernels must poison exactly those frames — across the window, the reflect padding and
    the fused chain — and leave the others bit-for-bit unaffected."""
    x = signals.uniform((2, 1, 9000), seed=51)
    clean = x.copy()
    x[0, 0, 4000] = np.nan
    x[1, 0, 10] = np.inf                                    # inside the reflected prefix of the first frames
    mel = tac.Melspectrogram(num_mels=64, sample_rate=16000, fft_length=1024, hop_length=256).cuda()
    chain = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
    want = torch_ref.melspectrogram_db(torch.from_numpy(x), n_fft=1024, hop=256, num_mels=64, sample_rate=16000).numpy()
    got = host(chain(dev(x)))
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.isnan(got).any() and not np.isnan(got).all()
    ok = ~np.isnan(want)
    assert np.abs(got[ok] - want[ok]).max() < DB_ABS
    ref_clean = host(chain(dev(clean)))
    assert np.array_equal(got[ok], ref_clean[ok])            # untouched frames do not depend on the poisoned ones
    # complex STFT: which bins of a poisoned frame come out NaN and which +-Inf (and whether the exactly-zero
    # imaginary parts of DC / Nyquist stay zero) depends on the summation order, so compare per frame
    z = host(tac.stft(dev(x), 1024, hop_length=256))
    zw = torch_ref.stft(torch.from_numpy(x), 1024, 256).numpy()
    bad, bad_w = ~np.isfinite(z).all(axis=(2, 4)), ~np.isfinite(zw).all(axis=(2, 4))
    assert np.array_equal(bad, bad_w) and bad.any()
    keep = np.broadcast_to(~bad[:, :, None, :, None], z.shape)
    assert rel_err(z[keep], zw[keep]) < TIGHT


def test_non_contiguous_and_odd_layout_inputs(tac):
    """Strided time axes, transposed leading dims, row strides that break 8-/16-byte alignment and storage offsets
    must give exactly what the dense copy gives (the kernels take a row stride; everything else is normalised on
    the host without touching the values)."""
    base = dev(signals.uniform((3, 4, 8192), seed=61))
    mel = torch.nn.Sequential(*tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=512, hop_length=128),
                              tac.AmplitudeToDb()).cuda()
    views = {
        'time-strided': base[..., ::2],
        'leading-transposed': base.transpose(0, 1),
        'row-stride-odd': base[..., :4097],               # rows start at odd sample offsets: scalar load path
        'offset-storage': base[1:, 1:, 3:],
        'expanded-channel': base[:, :1].expand(3, 4, 8192),
    }
    for name, v in views.items():
        dense = v.contiguous()
        assert torch.equal(tac.stft(v, 512, hop_length=128), tac.stft(dense, 512, hop_length=128)), name
        assert torch.equal(mel(v), mel(dense)), name
        want = torch_ref.stft(dense.cpu(), 512, 128).numpy()
        assert rel_err(host(tac.stft(v, 512, hop_length=128)), want) < TIGHT, name
    # elementwise ops on non-dense inputs
    spec = tac.Spectrogram(512, 128, power=2.).cuda()(base)           # strided (…, F, T) view of a frame-major buffer
    assert torch.equal(tac.amplitude_to_db(spec), tac.amplitude_to_db(spec.contiguous()))
    assert torch.equal(tac.mu_law_encoding(base[..., ::3], 256), tac.mu_law_encoding(base[..., ::3].contiguous(), 256))


def test_kernels_run_on_the_callers_stream(tac):
    """Every entry point launches on torch's current stream (the C ABI takes the stream as an argument): work queued
    on a side stream behind a long-running producer must see the producer's data, without a device-wide sync."""
    side = torch.cuda.Stream()
    x = dev(signals.uniform((8, 1, 40000), seed=71))
    mel = torch.nn.Sequential(*tac.Melspectrogram(num_mels=64, sample_rate=16000, fft_length=1024, hop_length=256),
                              tac.AmplitudeToDb()).cuda()
    want = mel(x).clone()
    want_codes = tac.mu_law_encoding(x, 256).clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        buf = torch.zeros_like(x)
        for _ in range(20):                                  # keep the side stream busy ahead of the copy
            buf = buf + 0.0
        buf.copy_(x, non_blocking=True)
        got = tac.realize(mel(buf))
        codes = tac.mu_law_encoding(buf, 256)
    side.synchronize()
    assert torch.equal(got, want) and torch.equal(codes, want_codes)


def test_fused_call_is_hip_graph_capturable(tac):
    """After a warm-up call the fused pipeline is one kernel launch on the current stream plus an allocator hit, so
    it can be captured in a HIP graph and replayed on new data in the captured buffer."""
    model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                                tac.AmplitudeToDb()).cuda()
    x = dev(signals.uniform((4, 1, 16000), seed=81))
    for _ in range(3):
        tac.realize(model(x))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = tac.realize(model(x))
    x.copy_(dev(signals.uniform((4, 1, 16000), seed=82)))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, tac.realize(model(x)))


# ------------------------------------------------------------------ mu-law: bit-exact integers
def test_mulaw_golden_bit_exact(tac, golden):
    g = golden('g5_mulaw')
    x2 = dev(signals.uniform((1000000,), seed=8, scale=1.0))
    enc = tac.mu_law_encoding(x2, 256)
    assert enc.dtype == torch.int64
    assert np.array_equal(host(enc), g['enc256_unit'].astype(np.int64))
    codes = dev((signals.uniform((4096,), seed=9) * 127.5 + 127.5).astype(np.int64).clip(0, 255))
    dec = tac.mu_law_decoding(codes, 256)
    assert np.array_equal(host(dec).view(np.uint32), g['dec256_codes'].view(np.uint32))
    assert np.array_equal(host(tac.mu_law_decoding(torch.arange(256).cuda(), 256)).view(np.uint32),
                          g['lut256'].view(np.uint32))
    # round trip enc(dec(c)) == c for every code (tests/test_functional.py:195-199)
    allc = torch.arange(256).cuda()
    assert torch.equal(tac.mu_law_encoding(tac.mu_law_decoding(allc, 256), 256), allc)
    layer_rt = tac.MuLawEncoding(256)(tac.MuLawDecoding(256)(allc))
    assert torch.equal(layer_rt, allc)


def test_mulaw_exhaustive_on_device(tac):
    """Every float32 in [-1, 1] (2 x 1 065 353 217 bit patterns) through the n_quantize = 256 encoder against a
    binary search in the reference's threshold tables: the estimate-and-correct kernel is exact everywhere."""
    import subprocess, sys, os
    tool = os.path.join(os.path.dirname(__file__), '..', 'tools', 'check_mulaw_exhaustive.py')
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'values: 0' in out.stdout, out.stdout + out.stderr


def test_mulaw_thresholds_edges(tac, golden):
    g = golden('g5_mulaw')
    pos = g['thr256_pos_bits'].astype(np.uint32)
    neg = g['thr256_neg_bits'].astype(np.uint32)
    # exactly at, and one ulp below, every threshold; plus +-0, +-1 (same vector as tests/golden/make_golden.py)
    mags = np.concatenate([pos, pos - 1, neg, neg - 1, [0, 0x3f800000]]).astype(np.uint32)
    xs = np.concatenate([mags.view(np.float32), -(mags.view(np.float32))])
    got = host(tac.mu_law_encoding(dev(xs), 256))
    assert np.array_equal(got, g['enc256_edges'].astype(np.int64))


def test_mulaw_out_of_range_and_other_nq_bit_exact(tac, golden):
    """Outside [-1, 1] and for n_quantize != 256 the encoder evaluates the closed form with the reference CPU
    path's exact float32 roundings (csrc/exact_math.hpp), so those codes are bit-exact too."""
    g = golden('g5_mulaw')
    x1 = signals.uniform((1000000,), seed=7, scale=4.0)
    assert np.array_equal(host(tac.mu_law_encoding(dev(x1), 256)), g['enc256_scale4'].astype(np.int64))
    x2 = signals.uniform((200000,), seed=8, scale=1.0)[:200000]
    for nq, key in ((65536, 'enc65536_unit'), (16, 'enc16_unit')):
        assert np.array_equal(host(tac.mu_law_encoding(dev(x2), nq)), g[key].astype(np.int64)), key
    x3 = signals.uniform((200000,), seed=11, scale=1000.0)
    for nq, key in ((1024, 'enc1024_scale1000'), (7, 'enc7_scale1000')):
        assert np.array_equal(host(tac.mu_law_encoding(dev(x3), nq)), g[key].astype(np.int64)), key
    # +-0, +-1, denormals, just outside the unit interval, huge, inf and NaN (x86 float->int64 "indefinite")
    sp = dev(g['special_inputs'])
    assert np.array_equal(host(tac.mu_law_encoding(sp, 256)), g['enc256_special'])
    assert np.array_equal(host(tac.mu_law_encoding(sp, 65536)), g['enc65536_special'])
    # unaligned / odd-length views take the scalar path of the same kernel
    assert np.array_equal(host(tac.mu_law_encoding(dev(x3)[1:99998], 1024)), g['enc1024_scale1000'][1:99998].astype(np.int64))


def test_mulaw_decode_float_codes_bit_exact(tac, golden):
    """reference tests/test_functional.py:182-193 restated: the reference's own decoding test feeds FLOAT-typed codes
    (`waveform_mu.float()`) and compares with torch.eq.  Integral float codes go through the reference's table, so
    the bits are the golden table's for every code, int64- or float-typed, any container dtype."""
    g = golden('g5_mulaw')
    lut = g['lut256'].view(np.uint32)
    codes = torch.randint(low=0, high=255, size=(1, 1024), generator=torch.Generator().manual_seed(5))
    for c in (codes, torch.arange(256)[None], torch.arange(256).repeat(33)[None, 3:]):    # last: unaligned pointer
        got_f = host(tac.mu_law_decoding(c.float().cuda(), 256))
        got_i = host(tac.mu_law_decoding(c.cuda(), 256))
        assert got_f.dtype == np.float32
        assert np.array_equal(got_f.view(np.uint32), lut[c.numpy()])
        assert np.array_equal(got_i.view(np.uint32), lut[c.numpy()])
    assert np.array_equal(host(tac.MuLawDecoding(256)(codes.float().cuda())).view(np.uint32), lut[codes.numpy()])
    # both ways (reference :195-199), from float-typed codes
    assert torch.equal(codes.cuda(), tac.mu_law_encoding(tac.mu_law_decoding(codes.float().cuda(), 256), 256))
    # half-typed codes keep their dtype, like the reference's elementwise chain
    assert tac.mu_law_decoding(codes.half().cuda(), 256).dtype == torch.float16
    # non-integral / out-of-table float codes take the closed form: continuous through the table entries
    frac = host(tac.mu_law_decoding(torch.tensor([127.5, -1.0, 255.0, 256.0, 300.5]).cuda(), 256))
    y = np.array([127.5, -1.0, 255.0, 256.0, 300.5]) / 255 * 2 - 1
    want = np.sign(y) * (np.exp(np.abs(y) * np.log1p(255.0)) - 1) / 255
    assert (np.abs(frac - want) <= 1e-6 * np.abs(want)).all()       # float32 evaluation of y*log1p(mu), amplified by exp


def test_mulaw_decode_other_quantisations_ulp_bound(tac, golden):
    """n_quantize != 256: the closed form sign(y)(exp(|y| log1p(mu)) - 1)/mu with the reference's float32 op order and
    the device's expf.  The reference's CPU `exp` (MKL / Sleef, host dependent: two hosts were seen to differ in 4
    of the 256 table entries) is not correctly rounded, so there is no single bit pattern to match; the bound is one
    ulp of the exponential: |got - want| <= 2^-23 * exp(|y| log1p(mu)) / mu  (+ one ulp of the result)."""
    g = golden('g5_mulaw')
    nq, mu = 65536, 65535
    codes, want = np.arange(0, 65536, 16), g['lut65536_sample']
    e = np.exp(np.abs(codes / mu * 2 - 1.0) * np.log1p(mu))
    tol = 2 * 2.0 ** -23 * e / mu + 2 * np.spacing(np.abs(want))        # both sides within one ulp of exp
    for c in (torch.from_numpy(codes).cuda(), torch.from_numpy(codes).float().cuda()):
        got = host(tac.mu_law_decoding(c, nq))
        assert got.dtype == np.float32 and np.all(np.abs(got.astype(np.float64) - want) <= tol)
    # encode(decode(c)) == c also holds for 16-bit codes (the round trip the reference tests at 8 bits)
    c16 = torch.arange(0, 65536, 7).cuda()
    assert torch.equal(tac.mu_law_encoding(tac.mu_law_decoding(c16, 65536), 65536), c16)


def test_deferred_chain_is_safe(tac):
    """Deferral must be unobservable or loud (reference layers are eager): (1) a chain finished by AmplitudeToDb has
    already launched when forward returns, so overwriting the input afterwards cannot change it; (2) a chain left
    pending raises if its input was modified before first use; (3) realised before the overwrite it is correct."""
    a = dev(signals.audio_like((3, 1, 20000), seed=51))
    b = dev(signals.audio_like((3, 1, 20000), seed=52))
    mel = tac.Melspectrogram(num_mels=64, sample_rate=16000, fft_length=1024, hop_length=256).cuda()
    full = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
    want_a = host(full(a.clone()))
    buf = a.clone()
    y = full(buf)
    buf.copy_(b)                                   # next batch lands in the same buffer
    assert type(y) is torch.Tensor and np.array_equal(host(y), want_a)
    # round 6: a user-owned container that ENDS in ApplyFilterbank hands back an ordinary tensor, launched before forward
    # returns (type(nn.Sequential(*Melspectrogram(...))(x)) is torch.Tensor, as with the reference's eager layers) ...
    ended = torch.nn.Sequential(*mel).cuda()
    buf = a.clone()
    y = ended(buf)
    buf.copy_(b)
    assert type(y) is torch.Tensor and np.array_equal(host(y), host(mel(a)))
    assert type(torch.nn.Sequential(torch.nn.Sequential(mel[0], mel[1]), mel[2]).cuda()(a)) is torch.Tensor    # ... through nesting,
    assert type(torch.nn.Sequential(mel[0], mel[1]).cuda()(a)) is torch.Tensor and type(torch.nn.Sequential(mel[0]).cuda()(a)) is torch.Tensor
    inner = torch.nn.Sequential(torch.nn.Sequential(*mel), tac.AmplitudeToDb()).cuda()      # ... while a container that goes on still fuses
    before = launches(tac)
    assert type(inner(a)) is torch.Tensor and sum(launched_since(tac, before).values()) == 1
    # layers called one by one (the reference's tests/test_layers.py:98-101 style) keep deferring: the safety net below is theirs
    plain = lambda t: mel[2](mel[1](mel[0](t)))
    buf = a.clone()
    pend = plain(buf)
    assert isinstance(pend, tac.DeferredSpectral) and pend.pending()
    buf.copy_(b)
    with pytest.raises(RuntimeError, match='modified in place'):
        pend.cpu()
    buf = a.clone()
    pend = plain(buf)
    val = tac.realize(pend)
    buf.copy_(b)
    assert np.array_equal(host(val), host(mel(a)))
    # the window / filterbank are watched too
    pend = plain(a)
    mel[2].filterbank.mul_(1.0)
    with pytest.raises(RuntimeError, match='filterbank'):
        pend + 1
    # a pending chain launches on the stream of its forward call; a consumer on another stream is ordered behind it
    side = torch.cuda.Stream()
    pend = plain(a)
    with torch.cuda.stream(side):
        out = pend * 1.0
    side.synchronize()
    assert np.array_equal(host(out), host(mel(a)))


def test_planned_chain_is_one_bound_launch(tac):
    """tac.planned(model, example) (round 6): the reference idiom (layers.py:307-381) bound to ONE call — layout / stamp checks,
    allocation, stream lookup, launch — for callers that run many small batches; same values as the module chain, an ordinary
    tensor, and anything the binding does not cover goes through the chain itself."""
    x = dev(signals.audio_like((4, 1, 20000), seed=61))
    mel = tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512).cuda()
    model = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
    want = model(x)
    fast = tac.planned(model, x)
    assert fast.fused()
    before = launches(tac)
    y = fast(x)
    assert type(y) is torch.Tensor and launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
    assert y.shape == want.shape and y.stride() == want.stride() and torch.equal(y, want)
    assert torch.equal(fast(x[:2]), model(x[:2]))                    # another layout: the chain itself
    lin = tac.planned(torch.nn.Sequential(*mel), x)                  # no dB stage
    assert lin.fused() and torch.equal(lin(x), mel(x))
    nested = tac.planned(torch.nn.Sequential(torch.nn.Sequential(*mel), tac.AmplitudeToDb(ref=2.0, amin=1e-6)).cuda(), x)
    assert nested.fused() and torch.equal(nested(x), torch.nn.Sequential(*mel, tac.AmplitudeToDb(ref=2.0, amin=1e-6)).cuda()(x))
    mel[2].filterbank.mul_(0.5)                                      # new contents of a bound constant: seen, not stale
    assert torch.equal(fast(x), model(x)) and not torch.equal(fast(x), want)
    xg = x.clone().requires_grad_(True)                              # a gradient: the chain's autograd path
    yg = fast(xg)
    assert yg.requires_grad
    (g1,) = torch.autograd.grad(yg.sum(), xg)
    xg2 = x.clone().requires_grad_(True)
    (g2,) = torch.autograd.grad(model(xg2).sum(), xg2)
    assert torch.equal(g1, g2)
    other = tac.planned(torch.nn.Sequential(tac.STFT(2048, hop_length=512), tac.ComplexNorm(2.)).cuda(), x)    # not a mel chain
    assert not other.fused() and type(other(x)) is torch.Tensor and torch.equal(other(x), tac.Spectrogram(2048, hop_length=512, power=2.).cuda()(x))


def test_dtype_and_device_routes(tac):
    """float16 / bfloat16 run the kernels (widened) and elementwise results come back in the input dtype; float64 has
    kernels of its own for the STFT chain (test_float64_chain_runs_on_the_f64_kernels) and phase_vocoder (the dtype the
    reference tests it in); what is left outside the kernels (float64 mu-law) is an error under strict mode and torch's
    GPU operators with a warning otherwise."""
    import math
    x = dev(signals.audio_like((2, 1, 8000), seed=61))
    before = launches(tac)
    zh = tac.stft(x.half(), 256, 64)
    assert zh.dtype == torch.float32 and launched_since(tac, before) == {'tac_stft_f32': 1}
    assert rel_err(host(zh), host(tac.stft(x.half().float(), 256, 64))) == 0.0
    assert tac.amplitude_to_db(x.bfloat16()).dtype == torch.bfloat16
    assert tac.complex_norm(zh.half(), 2.0).dtype == torch.float16
    # float64 mu-law has its own kernels since round 5 (strict mode accepts it): the formulas in double, as the reference's
    # CPU path evaluates them for double input — codes equal to the oracle's, decoded values within a few ulps
    before = launches(tac)
    xd = torch.from_numpy(signals.uniform((200000,), seed=77, scale=1.2)).double()
    c64 = tac.mu_law_encoding(xd.cuda())
    assert c64.dtype == torch.int64 and launched_since(tac, before) == {'tac_mulaw_encode_f64_i64': 1}
    want = torch_ref.mu_law_encoding(xd, 256)
    assert int((c64.cpu() != want).sum()) == 0
    assert int((tac.mu_law_encoding(xd.cuda(), 1024).cpu() != torch_ref.mu_law_encoding(xd, 1024)).sum()) == 0
    for codes in (want.cuda(), want.double().cuda() + 0.25):                      # integer and fractional float64 codes
        before = launches(tac)
        d64 = tac.mu_law_decoding(codes, 256, dtype=torch.float64) if not codes.is_floating_point() else tac.mu_law_decoding(codes, 256)
        assert d64.dtype == torch.float64 and launched_since(tac, before) == {'tac_mulaw_decode_f64': 1}
        ref64 = torch_ref.mu_law_decoding(codes.cpu(), 256, dtype=torch.float64)
        assert np.abs(host(d64) - ref64.numpy()).max() < 1e-14
    z = torch.randn(1, 2, 1025, 400, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    adv = torch.linspace(0, math.pi * 256, 1025, dtype=torch.float64)[..., None]
    for rate in (0.5, 1.01, 1.3):                                  # reference tests/test_functional.py:69
        before = launches(tac)
        got = tac.phase_vocoder(z.cuda(), rate, adv.cuda())
        assert got.dtype == torch.float64 and launched_since(tac, before) == {'tac_phase_vocoder_f64': 1}
        assert tuple(got.shape) == (1, 2, 1025, int(math.ceil(400 / rate)), 2)
        want = torch_ref.phase_vocoder(z, rate, adv)
        assert rel_err(host(got), want.numpy()) < 1e-9              # the reference's own bar is atol 1e-5
    # the float32 call on the same data never forms the ill-conditioned sum (unit phasor products): within 3e-5 of the float64
    # evaluation (the reference's own float32 evaluation is ~1e-3 away)
    got32 = tac.phase_vocoder(z.float().cuda(), 1.3, adv.float().cuda())
    want64 = torch_ref.phase_vocoder(z.float().double(), 1.3, adv.float().double())
    assert got32.dtype == torch.float32 and rel_err(host(got32), want64.numpy()) < 3e-5


def test_float64_chain_runs_on_the_f64_kernels(tac):
    """VERDICT r2 missing #4: the reference keeps float64 in -> float64 out (functional.py:48-113); float64 tensors on the
    device are evaluated by csrc/chain_f64.hip — launch counters under strict mode — and agree with the float64 oracle to
    1e-12 of the result's scale: power-of-two sizes (radix-4 passes with and without a radix-2 pass, 8192 = twiddles
    from global memory), mixed radix (400 = 2 x 4 x 2 x 5 x 5, 300, 6000), the direct transform (odd 77, 2 x 67), every pad mode, short windows, two-sided, normalised,
    non-contiguous rows; then complex_norm / angle / magphase / apply_filterbank (both layouts) / the dB pair and the
    Melspectrogram + AmplitudeToDb layers; the float32 kernels agree with it to float32 accuracy."""
    rng = np.random.default_rng(71)
    x = rng.standard_normal((2, 2, 9000))
    xd = torch.from_numpy(x).cuda()
    cases = [(2048, 512, None, True, 'reflect', False, True), (1024, 256, 800, True, 'constant', True, True),
             (512, 128, None, False, 'reflect', False, True), (256, 100, 200, True, 'replicate', False, False),
             (64, 16, None, True, 'circular', False, True), (8, 4, None, True, 'reflect', False, True),
             (4096, 1024, None, True, 'reflect', False, True), (8192, 2048, 5000, True, 'reflect', True, True),
             (400, 160, None, True, 'reflect', False, True), (300, 75, 256, True, 'reflect', False, False),
             (77, 20, None, False, 'reflect', True, True), (134, 50, None, True, 'reflect', False, True),
             (6000, 1500, None, True, 'reflect', False, True), (4, 1, None, True, 'reflect', False, False)]
    for n_fft, hop, wl, center, mode, normalized, onesided in cases:
        win = torch.hann_window(wl or n_fft, dtype=torch.float64) + 0.1
        kw = dict(win_length=wl, window=win, center=center, pad_mode=mode, normalized=normalized, onesided=onesided)
        want = torch_ref.stft(torch.from_numpy(x), n_fft, hop, **kw).numpy()
        before = launches(tac)
        got = tac.stft(xd, n_fft, hop, **{**kw, 'window': win.cuda()})
        assert launched_since(tac, before) == {'tac_stft_f64': 1}, (n_fft, launched_since(tac, before))
        assert got.dtype == torch.float64 and tuple(got.shape) == want.shape
        assert not got.is_contiguous() and got.transpose(-3, -2).is_contiguous()      # frame-major, like the float32 kernels
        assert np.abs(host(got) - want).max() < 1e-12 * np.abs(want).max(), n_fft
    # rows that are a strided view (every other channel of a wider buffer)
    wide = torch.from_numpy(rng.standard_normal((3, 4, 5000))).cuda()
    view = wide[:, ::2]
    want = torch_ref.stft(view.cpu(), 512, 128, window=torch.hann_window(512, dtype=torch.float64)).numpy()
    assert np.abs(host(tac.stft(view, 512, 128)) - want).max() < 1e-12 * np.abs(want).max()
    # against the float32 kernels
    z32 = host(tac.stft(xd.float(), 1024, 256))
    assert rel_err(z32, host(tac.stft(xd, 1024, 256))) < 2e-6
    # the pair ops, in the kernels' frame-major layout and on a plain contiguous tensor
    z = tac.stft(xd, 512, 128)
    zc = z.contiguous()
    for t in (z, zc):
        before = launches(tac)
        mag, ph = tac.magphase(t, 1.0)
        p07, ang = tac.complex_norm(t, 0.7), tac.angle(t)
        assert launched_since(tac, before) == {'tac_magphase_f64': 3}
        zz = host(t)
        assert np.abs(host(mag) - np.hypot(zz[..., 0], zz[..., 1])).max() < 1e-12 * np.abs(zz).max()
        assert np.abs(host(p07) - np.hypot(zz[..., 0], zz[..., 1]) ** 0.7).max() < 1e-12 * np.abs(zz).max()
        assert np.abs(host(ph) - np.arctan2(zz[..., 1], zz[..., 0])).max() < 1e-14 and np.array_equal(host(ph), host(ang))
    # apply_filterbank on both layouts + the dB pair
    fb = tac.create_mel_filter(257, 40, 0.0, 8000.0, False).double().cuda()
    spec = tac.complex_norm(z, 2.0)
    for t in (spec, spec.contiguous()):
        before = launches(tac)
        mel = tac.apply_filterbank(t, fb)
        assert launched_since(tac, before) == {'tac_apply_filterbank_f64': 1}
        want = np.einsum('...ft,fm->...mt', host(t), host(fb))
        assert mel.dtype == torch.float64 and np.abs(host(mel) - want).max() < 1e-13 * np.abs(want).max()
    before = launches(tac)
    db = tac.amplitude_to_db(mel, ref=2.0, amin=1e-7)
    back = tac.db_to_amplitude(db, ref=2.0)
    assert launched_since(tac, before) == {'tac_amplitude_to_db_f64': 1, 'tac_db_to_amplitude_f64': 1}
    m = host(mel)
    assert np.abs(host(db) - 10 * (np.log10(np.maximum(m * m, 1e-7)) - np.log10(2.0))).max() < 1e-11
    assert np.abs(host(back) - np.sqrt(np.maximum(m * m, 1e-7))).max() < 1e-12 * m.max()
    # the layers: Melspectrogram + AmplitudeToDb on float64 modules run op by op (deferral is a float32 feature); the fused
    # tac_amd::melspectrogram op = two launches; both give the values of the float64 oracle
    for n_fft, hop, mels in ((2048, 512, 128), (400, 160, 40)):
        chain = torch.nn.Sequential(*tac.Melspectrogram(num_mels=mels, sample_rate=16000, fft_length=n_fft, hop_length=hop),
                                    tac.AmplitudeToDb()).double().cuda()
        before = launches(tac)
        y = chain(xd)
        assert launched_since(tac, before) == {'tac_stft_f64': 1, 'tac_magphase_f64': 1, 'tac_apply_filterbank_f64': 1,
                                               'tac_amplitude_to_db_f64': 1}
        bank = chain[2].filterbank.cpu()
        ref = torch_ref.amplitude_to_db(torch_ref.apply_filterbank(torch_ref.complex_norm(
            torch_ref.stft(torch.from_numpy(x), n_fft, hop, window=chain[0].window.cpu()), 2.0), bank)).numpy()
        assert y.dtype == torch.float64 and np.abs(host(y) - ref).max() < 1e-9
        before = launches(tac)
        y2 = torch.ops.tac_amd.melspectrogram(xd, chain[0].window, chain[2].filterbank, n_fft, hop, n_fft, True, 'reflect',
                                              False, True, 2.0, True, 1.0, 1e-7)
        assert launched_since(tac, before) == {'tac_spectrogram_f64': 1, 'tac_apply_filterbank_f64': 1}
        assert y2.dtype == torch.float64 and np.abs(host(y2) - ref).max() < 1e-9
    # gradients of float64 calls differentiate the stock-torch evaluation: announced, an error under strict mode
    xg = xd.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match='strict mode'):
        tac.complex_norm(tac.stft(xg, 256, 64), 2.0).sum().backward()


@pytest.mark.parametrize('shape,n_fft,hop,mels', [((2, 1, 12000), 1024, 256, 64), ((3, 2, 9000), 512, 128, 40),
                                                  ((1, 1, 40000), 2048, 512, 128)])
def test_autograd_matches_the_reference_chain(tac, shape, n_fft, hop, mels):
    """The reference is differentiable end to end through stock torch; gradients of the product's chain on the GPU vs
    torch.autograd.grad through the CPU restatement of the reference chain, <= 1e-3 relative."""
    x = signals.audio_like(shape, seed=71)
    weight = signals.uniform(shape[:-1] + (mels, 1 + shape[-1] // hop), seed=72)
    xc = torch.from_numpy(x).requires_grad_(True)
    want_y = torch_ref.melspectrogram_db(xc, amin=1e-5, n_fft=n_fft, hop=hop, num_mels=mels, sample_rate=16000)
    (want,) = torch.autograd.grad((want_y * torch.from_numpy(weight)).sum(), xc)
    xg = dev(x).requires_grad_(True)
    mel = tac.Melspectrogram(num_mels=mels, sample_rate=16000, fft_length=n_fft, hop_length=hop).cuda()
    for idiom, chain in ((False, torch.nn.Sequential(mel, tac.AmplitudeToDb(amin=1e-5))),   # factory container + dB
                         (True, torch.nn.Sequential(*mel, tac.AmplitudeToDb(amin=1e-5)))):   # the unpacked (reference) idiom
        before = launches(tac)
        y = chain(xg)
        if idiom:       # deferred although the waveform requires grad: ONE fused forward kernel + the dB op, which keeps
            #             the linear mel values its gradient needs (fused in, backward would have to recompute them)
            assert launched_since(tac, before) == {'tac_melspec_sparse_f32': 1, 'tac_amplitude_to_db_f32': 1}
        assert y.requires_grad and np.abs(host(y) - want_y.detach().numpy()).max() < DB_ABS
        before = launches(tac)
        (got,) = torch.autograd.grad((y * dev(weight)).sum(), xg)
        ran = launched_since(tac, before)
        # the HIP gradient kernels of the fused op, either way: the backward kernel transforms the frames again itself and
        # folds the norm's adjoint into the inverse FFT's load — no stft launch, no spectrum or gradient spectrum in memory
        assert 'tac_stft_f32' not in ran and 'tac_complex_norm_backward_f32' not in ran, ran
        assert 'tac_stft_norm_backward_f32' not in ran, ran
        if (n_fft == 2048 and hop % 128 == 0) or (n_fft in (512, 1024) and hop in (n_fft // 8, n_fft // 4, n_fft // 2)):
            # ... the filterbank adjoint and the overlap-add happen inside it as well
            assert ran.get('tac_melspectrogram_backward_ola_f32') == 1 and 'tac_overlap_add_f32' not in ran, ran
            assert 'tac_apply_filterbank_adjoint_f32' not in ran and 'tac_spectrogram_backward_ola_f32' not in ran, ran
        elif n_fft in (256, 512, 1024) and hop % (n_fft // 16) == 0:         # ... and the overlap-add happens in its LDS
            assert ran.get('tac_spectrogram_backward_ola_f32') == 1 and 'tac_overlap_add_f32' not in ran, ran
        else:
            assert ran.get('tac_spectrogram_backward_f32') == 1 and ran.get('tac_overlap_add_f32') == 1, ran
        assert rel_err(host(got), want.numpy()) < 1e-3


@pytest.mark.parametrize('n_fft,hop,win_length,onesided,center,pad_mode,normalized', [
    (400, 160, None, True, True, 'reflect', False),       # the speech front end (mixed-radix forward kernel)
    (400, 100, 320, False, True, 'constant', True),       # ... two-sided, short window, normalized
    (300, 75, None, True, True, 'reflect', False),        # generic Stockham kernel (7-smooth half), forward and adjoint
    (250, 100, 200, True, False, 'reflect', False),       # ... not centred, short window
    (1001, 250, None, True, True, 'reflect', False),      # odd: DFT-matrix product, forward and adjoint
    (1018, 254, 800, False, True, 'constant', False),     # half = 509 (prime): DFT matrix, two-sided
    (512, 128, None, False, True, 'replicate', False),    # two-sided power of two
    (2048, 500, 1500, True, True, 'circular', False),     # window gradient at an FFT size, hop not a multiple of 128
    (4500, 1125, None, True, True, 'reflect', False),     # one frame per workgroup in the generic Stockham kernel
    (882, 441, 800, False, True, 'replicate', True),      # radix 7, two-sided, normalized
])
def test_general_gradient_routes_under_strict(tac, n_fft, hop, win_length, onesided, center, pad_mode, normalized):
    """The reference differentiates through every argument with stock torch (functional.py:99-107, 183-184).  Under
    set_strict(True) — the stock-torch route is an error — gradients w.r.t. the waveform, the WINDOW and the FILTERBANK
    come from the HIP entry points for every fft_length the forward kernels cover (FFT sizes, 400, DFT-matrix sizes),
    one- and two-sided, and agree with torch.autograd through the CPU oracle."""
    assert tac._ops.strict()
    shape = (2, 2, 3 * n_fft + 37)
    x = signals.audio_like(shape, seed=301)
    wl = n_fft if win_length is None else win_length
    win = (np.hanning(wl + 2)[1:-1] + 0.1).astype(np.float32)
    n_bins = n_fft // 2 + 1 if onesided else n_fft
    n_mels = 24
    fb = np.abs(signals.uniform((n_bins, n_mels), seed=302)).astype(np.float32)
    kw = dict(hop_length=hop, win_length=win_length, center=center, pad_mode=pad_mode, normalized=normalized, onesided=onesided)

    def chains(mod, xt, wt, ft):
        z = mod.stft(xt, n_fft, window=wt, **kw)
        p = mod.complex_norm(z, 2.0)
        m = mod.apply_filterbank(p, ft)
        return z, p, mod.amplitude_to_db(m, 1.0, 1e-3)

    class RefMod:                                          # the oracle under the product's argument names
        stft = staticmethod(lambda xt, n, window, hop_length, win_length, center, pad_mode, normalized, onesided:
                            torch_ref.stft(xt, n, hop_length, win_length, window, center, pad_mode, normalized, onesided))
        complex_norm = staticmethod(torch_ref.complex_norm)
        apply_filterbank = staticmethod(torch_ref.apply_filterbank)
        amplitude_to_db = staticmethod(torch_ref.amplitude_to_db)

    cx, cw, cf = (torch.from_numpy(a).double().requires_grad_(True) for a in (x, win, fb))
    gx, gw, gf = (dev(a).requires_grad_(True) for a in (x, win, fb))
    outs_ref = chains(RefMod, cx, cw, cf)
    outs_got = chains(tac, gx, gw, gf)
    for stage, (yr, yg) in enumerate(zip(outs_ref, outs_got)):
        wgt = signals.uniform(tuple(yr.shape), seed=310 + stage)
        ins_r, ins_g = ((cx, cw), (gx, gw)) if stage < 2 else ((cx, cw, cf), (gx, gw, gf))
        want = torch.autograd.grad((yr * torch.from_numpy(wgt).double()).sum(), ins_r, retain_graph=True)
        before = launches(tac)
        got = torch.autograd.grad((tac.realize(yg) * dev(wgt)).sum(), ins_g, retain_graph=True)
        ran = launched_since(tac, before)
        assert 'tac_overlap_add_f32' in ran and 'tac_window_grad_f32' in ran, ran
        for name, a, b in zip(('waveform', 'window', 'filterbank'), got, want):
            assert rel_err(host(a), b.numpy()) < 1e-4, (stage, name)
    # waveform only at fft_length 400: still no stock-torch route, whatever the op
    routed = dict(tac._ops.composite_calls)
    y = tac.Spectrogram(400, 160, power=2.).cuda()(gx)
    (g1,) = torch.autograd.grad(y.sum(), gx)
    assert tac._ops.composite_calls == routed
    # the fused op with a learnable filterbank and window
    mel = tac.Melspectrogram(num_mels=20, sample_rate=16000, fft_length=512, hop_length=128).cuda()
    mel[0].window.requires_grad_(True)
    mel[2].filterbank.requires_grad_(True)
    xs = dev(signals.audio_like((2, 1, 6000), seed=303))
    before = launches(tac)
    gwin, gbank = torch.autograd.grad(tac.realize(mel(xs)).sum(), (mel[0].window, mel[2].filterbank))
    cwin, cbank = mel[0].window.detach().cpu().double().requires_grad_(True), mel[2].filterbank.detach().cpu().double().requires_grad_(True)
    ref = torch_ref.apply_filterbank(torch_ref.complex_norm(torch_ref.stft(xs.cpu().double(), 512, 128, window=cwin), 2.0), cbank)
    rwin, rbank = torch.autograd.grad(ref.sum(), (cwin, cbank))
    assert rel_err(host(gwin), rwin.numpy()) < 1e-4 and rel_err(host(gbank), rbank.numpy()) < 1e-4


def test_fft_length_400_trains_on_the_mixed_radix_kernels(tac):
    """The 25 ms speech front end (fft_length 400, hop 160) under strict mode: Spectrogram, the 80-band Melspectrogram
    and the complex stft differentiate w.r.t. the waveform through the inverse form of the mixed-radix kernel
    (stft_n400_backward_kernel: the 8 x 25 transform on conjugated data; for |z|^p the frames are re-transformed inside
    it, no spectrum exists in memory) with the overlap-add of a unit's eight frames inside the kernel when hop and padding
    are multiples of four (else frame gradients + the gather overlap-add) — launch counters asserted — and agree with
    torch.autograd through the CPU oracle.  Short rows (every unit touches the padding), odd frame counts, every pad mode,
    hops from 52 to 400 and power 1 included."""
    assert tac._ops.strict()
    x = signals.audio_like((3, 1, 16000), seed=331)
    xc = torch.from_numpy(x).double().requires_grad_(True)
    xg = dev(x).requires_grad_(True)
    # power spectrogram
    for power in (2.0, 1.0):
        want_y = torch_ref.complex_norm(torch_ref.stft(xc, 400, 160), power)
        wgt = signals.uniform(tuple(want_y.shape), seed=332)
        (want,) = torch.autograd.grad((want_y * torch.from_numpy(wgt).double()).sum(), xc)
        y = tac.Spectrogram(400, 160, power=power).cuda()(xg)
        before = launches(tac)
        (got,) = torch.autograd.grad((y * dev(wgt)).sum(), xg)
        ran = launched_since(tac, before)
        assert ran == {'tac_spectrogram_backward_ola_f32': 1}, ran       # frames re-transformed AND overlap-added in the kernel
        assert rel_err(host(got), want.numpy()) < 1e-4, power
    xe = signals.audio_like((5, 2, 1234), seed=336)                   # 8 frames per row: every unit gathers its samples
    xec, xeg = torch.from_numpy(xe).double().requires_grad_(True), dev(xe).requires_grad_(True)
    for pad_mode, hop, center in (('reflect', 160, True), ('constant', 90, True), ('circular', 200, True), ('replicate', 52, True),
                                  ('reflect', 400, True), ('reflect', 160, False), ('constant', 120, True), ('reflect', 48, True)):
        wy = torch_ref.complex_norm(torch_ref.stft(xec, 400, hop, pad_mode=pad_mode, center=center), 2.0)
        wg = signals.uniform(tuple(wy.shape), seed=337)
        (want,) = torch.autograd.grad((wy * torch.from_numpy(wg).double()).sum(), xec)
        y = tac.Spectrogram(400, hop, pad_mode=pad_mode, center=center, power=2.).cuda()(xeg)
        before = launches(tac)
        (got,) = torch.autograd.grad((y * dev(wg)).sum(), xeg)
        fused = hop % 4 == 0 and hop >= 50
        assert launched_since(tac, before) == ({'tac_spectrogram_backward_ola_f32': 1} if fused else
                                               {'tac_spectrogram_backward_f32': 1, 'tac_overlap_add_f32': 1}), (pad_mode, hop)
        assert rel_err(host(got), want.numpy()) < 1e-4, (pad_mode, hop, center)
    # long rows without centring: the frames stop short of the row's end (positions past the last frame get no gradient)
    xl = signals.audio_like((2, 1, 9000), seed=339)
    xlc, xlg = torch.from_numpy(xl).double().requires_grad_(True), dev(xl).requires_grad_(True)
    wy = torch_ref.complex_norm(torch_ref.stft(xlc, 400, 160, center=False), 2.0)
    wg = signals.uniform(tuple(wy.shape), seed=340)
    (want,) = torch.autograd.grad((wy * torch.from_numpy(wg).double()).sum(), xlc)
    (got,) = torch.autograd.grad((tac.Spectrogram(400, 160, center=False, power=2.).cuda()(xlg) * dev(wg)).sum(), xlg)
    assert rel_err(host(got), want.numpy()) < 1e-4 and np.all(host(got)[..., 8880:] == 0.0)
    # the fused 80-band chain with its dB epilogue (the reference idiom)
    chain = torch.nn.Sequential(*tac.Melspectrogram(num_mels=80, sample_rate=16000, fft_length=400, hop_length=160),
                                tac.AmplitudeToDb(amin=1e-5)).cuda()
    xc32 = torch.from_numpy(x).requires_grad_(True)
    want_y = torch_ref.melspectrogram_db(xc32, amin=1e-5, n_fft=400, hop=160, num_mels=80, sample_rate=16000)
    wgt = signals.uniform(tuple(want_y.shape), seed=333)
    (want,) = torch.autograd.grad((want_y * torch.from_numpy(wgt)).sum(), xc32)
    y = chain(xg)
    before = launches(tac)
    (got,) = torch.autograd.grad((y * dev(wgt)).sum(), xg)
    ran = launched_since(tac, before)
    # (round 3: the filterbank adjoint is formed inside the backward kernel — no gradient spectrogram in memory)
    assert ran.get('tac_melspectrogram_backward_ola_f32') == 1 and 'tac_overlap_add_f32' not in ran, ran
    assert rel_err(host(got), want.numpy()) < 1e-3
    assert 'tac_apply_filterbank_adjoint_f32' not in ran and 'tac_apply_filterbank_f32' not in ran and 'tac_stft_f32' not in ran, ran
    # ... also for short rows whose every unit gathers its samples, power 1 and a bank with few bands
    for pad_mode, hop, mels, pw in (('reflect', 160, 40, 2.0), ('constant', 90, 17, 1.0), ('circular', 200, 128, 2.0)):
        xe32 = torch.from_numpy(xe).requires_grad_(True)
        want_m = torch_ref.apply_filterbank(torch_ref.complex_norm(torch_ref.stft(xe32, 400, hop, pad_mode=pad_mode), pw),
                                            torch_ref.create_mel_filter(201, mels, 0.0, 8000, False))
        bank = tac.MelFilterbank(num_mels=mels, sample_rate=16000, num_freqs=201).get_filterbank()
        mel_c = torch.nn.Sequential(*tac.Spectrogram(400, hop, pad_mode=pad_mode, power=pw), tac.ApplyFilterbank(bank))
        wg = signals.uniform(tuple(want_m.shape), seed=338)
        (want,) = torch.autograd.grad((want_m * torch.from_numpy(wg)).sum(), xe32)
        before = launches(tac)
        (got,) = torch.autograd.grad((mel_c.cuda()(xeg) * dev(wg)).sum(), xeg)
        assert launched_since(tac, before).get('tac_melspectrogram_backward_ola_f32' if hop % 4 == 0 else
                                               'tac_melspectrogram_backward_f32') == 1, (pad_mode, hop)
        assert rel_err(host(got), want.numpy()) < 1e-3, (pad_mode, hop, mels, pw)
    # complex stft, an odd number of frames per unit, short window, not centred
    xs = signals.audio_like((2, 2, 2011), seed=334)
    win = (np.hanning(302)[1:-1] + 0.2).astype(np.float32)
    xsc, wc = torch.from_numpy(xs).double().requires_grad_(True), torch.from_numpy(win).double()
    want_z = torch_ref.stft(xsc, 400, 77, 300, wc, False, 'reflect', True)
    wz = signals.uniform(tuple(want_z.shape), seed=335)
    (want,) = torch.autograd.grad((want_z * torch.from_numpy(wz).double()).sum(), xsc)
    xsg = dev(xs).requires_grad_(True)
    z = tac.stft(xsg, 400, 77, 300, dev(win), False, 'reflect', True)
    before = launches(tac)
    (got,) = torch.autograd.grad((z * dev(wz)).sum(), xsg)
    assert launched_since(tac, before) == {'tac_stft_backward_f32': 1, 'tac_overlap_add_f32': 1}
    assert rel_err(host(got), want.numpy()) < 1e-5


def test_backward_without_a_kernel_is_announced(tac):
    """What still differentiates through stock torch operators on the device (ops without gradient kernels, double
    backward) says so: an error under strict mode, a CompositeRouteWarning and a composite_calls entry otherwise."""
    x64 = dev(signals.audio_like((1, 1, 3000), seed=305).astype(np.float64)).requires_grad_(True)    # (float64: forward kernels, no gradient kernels)
    y = tac.Spectrogram(256, 64, power=2.).cuda().double()(x64)
    with pytest.raises(RuntimeError, match='strict mode'):
        y.sum().backward()
    x = dev(signals.audio_like((1, 1, 3000), seed=306)).requires_grad_(True)
    with pytest.raises(RuntimeError, match='strict mode'):
        torch.autograd.grad(torch.autograd.grad(tac.Spectrogram(256, 64, power=2.).cuda()(x).sum(), x, create_graph=True)[0].pow(2).sum(), x)
    tac.set_strict(False)
    try:
        tac._ops._warned.clear()
        with pytest.warns(tac.CompositeRouteWarning, match='backward'):
            (g,) = torch.autograd.grad(tac.Spectrogram(256, 64, power=2.).cuda()(x).sum(), x, create_graph=True)
        assert g.requires_grad
        (h,) = torch.autograd.grad(g.pow(2).sum(), x)
        xc = x.detach().cpu().double().requires_grad_(True)
        (gc,) = torch.autograd.grad(torch_ref.complex_norm(torch_ref.stft(xc, 256, 64), 2.0).sum(), xc, create_graph=True)
        (hc,) = torch.autograd.grad(gc.pow(2).sum(), xc)
        assert rel_err(host(h), hc.numpy()) < 1e-4
        assert any(k[1].startswith('backward: double backward') for k in tac._ops.composite_calls)
    finally:
        tac.set_strict(True)


def test_filterbank_adjoint_forms(tac):
    """Gradient of apply_filterbank w.r.t. the spectrogram (functional.py:183-184 under autograd): banks with at most two
    non-zero weights per bin (mel banks) take the per-bin table kernel, anything else the GEMM with the transposed bank;
    both against float64."""
    for n_freqs, n_mels, shape in ((1025, 128, (3, 1025, 77)), (257, 40, (2, 2, 257, 31)), (201, 23, (201, 9)),
                                   (2049, 80, (1, 2049, 40))):
        spec = dev(np.abs(signals.uniform(shape, seed=91)) + 0.1).requires_grad_(True)
        w = signals.uniform(shape[:-2] + (n_mels, shape[-1]), seed=92)
        mel_fb = tac.create_mel_filter(n_freqs, n_mels, 0.0, 8000.0, False).cuda()
        dense_fb = dev(signals.uniform((n_freqs, n_mels), seed=93))
        three = mel_fb.clone()
        three[5, :3] = torch.tensor([0.25, 0.5, 0.125])              # one bin with three bands: not a two-entry table
        for fb, entry in ((mel_fb, 'tac_apply_filterbank_adjoint_f32'), (dense_fb, 'tac_apply_filterbank_f32'),
                          (three, 'tac_apply_filterbank_f32')):
            y = tac.apply_filterbank(spec, fb)
            before = launches(tac)
            (got,) = torch.autograd.grad((tac.realize(y) * dev(w)).sum(), spec)
            assert launched_since(tac, before) == {entry: 1}, (n_freqs, entry)
            want = np.einsum('...mt,fm->...ft', w.astype(np.float64), host(fb).astype(np.float64))
            assert rel_err(host(got), want) < 1e-5, (n_freqs, entry)


def test_deferred_results_carry_gradients(tac):
    """A waveform that requires grad is deferred like any other (the reference idiom trains through the fused kernels);
    whatever consumes a pending result — a terminal layer, a torch function, a method, a view, an in-place-free
    expression, torch.autograd.grad on the pending result itself — gets the graph eager evaluation would have built."""
    x = signals.audio_like((2, 1, 9000), seed=81)
    n_fft, hop = 512, 128
    xc = torch.from_numpy(x).requires_grad_(True)
    zc = torch_ref.stft(xc, n_fft, hop)
    stft = tac.STFT(n_fft, hop).cuda()
    cases = [('pow-sum', lambda z: z.pow(2).sum()), ('transpose-mul', lambda z: (z.transpose(-2, -3) * 3.0).sum()),
             ('index', lambda z: z[..., 5:40, :, 0].abs().sum()), ('torch-fn', lambda z: torch.sum(torch.tanh(z))),
             ('reshape-matmul', lambda z: (z.reshape(z.shape[0], -1) @ torch.ones(z[0].numel(), 1, device=z.device, dtype=z.dtype)).sum()),
             ('norm-layer', None)]
    for name, fn in cases:
        xg = dev(x).requires_grad_(True)
        before = launches(tac)
        z = stft(xg)
        assert type(z) is tac._lazy.DeferredSpectral and z.pending() and z.requires_grad, name
        assert tuple(z.shape) == tuple(zc.shape) and launched_since(tac, before) == {}, name      # metadata only, nothing ran
        if fn is None:
            m = tac.ComplexNorm(power=2.0)(z)                       # still pending, still carrying the gradient
            assert type(m) is tac._lazy.DeferredSpectral and m.pending()
            loss, want_loss = m.sum(), torch_ref.complex_norm(zc, 2.0).sum()
            assert launched_since(tac, before).get('tac_spectrogram_f32') == 1
        else:
            loss, want_loss = fn(z), fn(zc)
        assert loss.requires_grad and abs(float(loss.detach()) - float(want_loss.detach())) <= 1e-4 * abs(float(want_loss.detach())) + 1e-3, name
        (got,) = torch.autograd.grad(loss, xg)
        (want,) = torch.autograd.grad(want_loss, xc, retain_graph=True)
        assert rel_err(host(got), want.numpy()) < 1e-4, name
    # torch.autograd.grad / backward on the pending result itself
    xg = dev(x).requires_grad_(True)
    y = torch.nn.Sequential(*tac.Spectrogram(n_fft, hop, power=2.0)).cuda()(xg)
    w = dev(signals.uniform(tuple(y.shape), seed=82))
    (got,) = torch.autograd.grad(y, xg, grad_outputs=w)
    yc = torch_ref.spectrogram(xc, n_fft, hop, power=2.0)
    (want,) = torch.autograd.grad(yc, xc, grad_outputs=torch.from_numpy(host(w)))
    assert rel_err(host(got), want.numpy()) < 1e-4
    # no_grad: an ordinary deferred result, nothing recorded
    with torch.no_grad():
        z = stft(dev(x).requires_grad_(True))
        assert not z.requires_grad and not (z * 2).requires_grad
    # a filterbank that requires grad behind a gradient-carrying pending spectrogram: both gradients, against the same
    # chain on CPU tensors (the filterbank gradient takes the documented torch-operator route, hence non-strict)
    tac.set_strict(False)
    try:
        xg = dev(x).requires_grad_(True)
        mel = tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=n_fft, hop_length=hop).cuda()
        mel[2].filterbank.requires_grad_(True)
        torch.nn.Sequential(*mel, tac.AmplitudeToDb())(xg).sum().backward()
        xr = torch.from_numpy(x).requires_grad_(True)
        melc = tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=n_fft, hop_length=hop)
        melc[2].filterbank.requires_grad_(True)
        torch.nn.Sequential(*melc, tac.AmplitudeToDb())(xr).sum().backward()
        assert rel_err(host(xg.grad), xr.grad.numpy()) < 1e-5
        assert rel_err(host(mel[2].filterbank.grad), melc[2].filterbank.grad.numpy()) < 1e-5
    finally:
        tac.set_strict(True)


@pytest.mark.parametrize('n_fft,hop,kw', [(512, 128, {}), (256, 64, dict(pad_mode='constant')), (1024, 300, dict(pad_mode='replicate')),
                                          (128, 32, dict(pad_mode='circular', normalized=True)), (4096, 1024, {}),
                                          (2048, 512, dict(center=False)), (512, 128, dict(win_length=400)), (64, 16, {})])
def test_stft_gradient_kernels(tac, n_fft, hop, kw):
    """Adjoint of the STFT kernels (inverse real FFT per frame + gather overlap-add) against torch.autograd through the
    CPU restatement of the reference, every pad mode (the images of the padding carry gradient too), several frames per
    wave (n_fft <= 1024), 32 elements per lane (n_fft = 4096); and of |.|^p and dB on their own."""
    x = signals.audio_like((2, 2, 9000), seed=91)
    gw = signals.uniform((2, 2, n_fft // 2 + 1, tac._hip.stft_frames(9000, n_fft, hop, kw.get('center', True)), 2), seed=92)
    xc = torch.from_numpy(x).requires_grad_(True)
    okw = {k: v for k, v in kw.items()}
    (want,) = torch.autograd.grad((torch_ref.stft(xc, n_fft, hop, **okw) * torch.from_numpy(gw)).sum(), xc)
    xg = dev(x).requires_grad_(True)
    before = launches(tac)
    (got,) = torch.autograd.grad((tac.stft(xg, n_fft, hop_length=hop, **kw) * dev(gw)).sum(), xg)
    assert launched_since(tac, before).get('tac_stft_backward_f32') == 1
    assert rel_err(host(got), want.numpy()) < 1e-5
    # |z|^p and dB adjoints (elementwise kernels), incl. the zero of the norm and the clamp of the dB
    z = signals.audio_like((3, 40, 17, 2), seed=93)
    z[0, :3, :2] = 0.0
    for power in (1.0, 2.0, 0.7):
        zc = torch.from_numpy(z).requires_grad_(True)
        (wz,) = torch.autograd.grad(torch_ref.complex_norm(zc, power).sum(), zc)
        zg = dev(z).requires_grad_(True)
        (gz,) = torch.autograd.grad(tac.complex_norm(zg, power).sum(), zg)
        ok = np.isfinite(wz.numpy())                                  # torch gives NaN at z == 0 for power < 1; the kernel gives 0
        assert rel_err(host(gz)[ok], wz.numpy()[ok]) < 1e-5 and np.isfinite(host(gz)).all()
    a = signals.audio_like((4, 700), seed=94)
    ac = torch.from_numpy(a).requires_grad_(True)
    (wa,) = torch.autograd.grad(torch_ref.amplitude_to_db(ac, 2.0, 1e-3).sum(), ac)
    ag = dev(a).requires_grad_(True)
    (ga,) = torch.autograd.grad(tac.amplitude_to_db(ag, 2.0, 1e-3).sum(), ag)
    assert rel_err(host(ga), wa.numpy()) < 1e-5


def test_library_ops_pass_opcheck(tac):
    """torch.library.opcheck: schema (no hidden mutation / aliasing) and FakeTensor agreement (shapes, strides, dtypes)
    of the registered ops on real device inputs."""
    from torch.library import opcheck
    x = dev(signals.audio_like((2, 1, 6000), seed=81))
    win = torch.hann_window(512).cuda()
    fb = tac.create_mel_filter(257, 20, 0.0, 8000, False).cuda()
    utils = ('test_schema', 'test_faketensor')
    opcheck(torch.ops.tac_amd.stft.default, (x, win, 512, 128, 512, True, 'reflect', False, True), test_utils=utils)
    opcheck(torch.ops.tac_amd.melspectrogram.default,
            (x, win, fb, 512, 128, 512, True, 'reflect', False, True, 2.0, True, 1.0, 1e-7), test_utils=utils)
    opcheck(torch.ops.tac_amd.amplitude_to_db.default, (x, 1.0, 1e-7), test_utils=utils)
    opcheck(torch.ops.tac_amd.mu_law_encoding.default, (x, 256), test_utils=utils)
    z = tac.stft(x, 512, 128)
    opcheck(torch.ops.tac_amd.complex_norm.default, (z, 2.0), test_utils=utils)
    opcheck(torch.ops.tac_amd.phase_vocoder.default, (z, torch.linspace(0, 3.14 * 128, 257).cuda(), 1.3), test_utils=utils)


def test_elementwise_gradient_kernels(tac):
    """angle, magphase and db_to_amplitude are differentiated by gfx950 kernels (round 6; strict mode is on: a fall-back to torch's
    operators would raise): against torch.autograd through the oracle on dense pairs, on the strided pairs the STFT kernels return,
    with either output of magphase unused; at z = 0 the kernels return 0 (torch's atan2 gradient is 0 there too, its norm gradient
    0 / 0: the convention of tac_complex_norm_backward_f32)."""
    rng = np.random.default_rng(77)
    z_np = rng.standard_normal((3, 2, 33, 17, 2)).astype(np.float32)
    z_np[0, 0, :4, :3] = 0.0
    for power in (1.0, 2.0, 0.7):
        gm_np = rng.standard_normal(z_np.shape[:-1]).astype(np.float32)
        gp_np = rng.standard_normal(z_np.shape[:-1]).astype(np.float32)
        zr = torch.from_numpy(z_np).requires_grad_(True)
        mr, pr = torch_ref.magphase(zr, power)
        (want,) = torch.autograd.grad([mr, pr], zr, [torch.from_numpy(gm_np), torch.from_numpy(gp_np)])
        (want_m,) = torch.autograd.grad(torch_ref.magphase(zr, power)[0], zr, torch.from_numpy(gm_np))
        (want_p,) = torch.autograd.grad(torch_ref.angle(zr), zr, torch.from_numpy(gp_np))
        z = dev(z_np).requires_grad_(True)
        before = launches(tac)
        m, ph = tac.magphase(z, power)
        (got,) = torch.autograd.grad([m, ph], z, [dev(gm_np), dev(gp_np)])
        assert launched_since(tac, before).get('tac_magphase_backward_f32') == 1
        live = np.ones(z_np.shape, dtype=bool)
        live[0, 0, :4, :3] = False
        scale = np.abs(want.numpy()[live]).max()
        assert np.abs(host(got) - want.numpy())[live].max() < 1e-5 * scale
        (got_m,) = torch.autograd.grad(tac.magphase(z, power)[0], z, dev(gm_np))
        assert np.abs(host(got_m) - want_m.numpy())[live].max() < 1e-5 * scale
        (got_p,) = torch.autograd.grad(tac.angle(z), z, dev(gp_np))
        assert np.abs(host(got_p) - want_p.numpy()).max() < 1e-5 * np.abs(want_p.numpy()).max()     # (atan2: 0 at the origin in both)
        assert np.all(host(got)[0, 0, :4, :3] == 0.0) and np.all(host(got_m)[0, 0, :4, :3] == 0.0)
    # the strided (frame-major) pairs of the STFT kernels, through a chain: d/dx of sum(angle(stft(x)) * c)
    x_np = signals.audio_like((2, 1, 6000), seed=78)
    c_np = rng.standard_normal((2, 1, 129, 1 + 6000 // 64)).astype(np.float32)
    xr = torch.from_numpy(x_np).requires_grad_(True)
    (want_x,) = torch.autograd.grad((torch_ref.angle(torch_ref.stft(xr, 256, 64)) * torch.from_numpy(c_np)).sum(), xr)
    x = dev(x_np).requires_grad_(True)
    zz = tac.stft(x, 256, 64)
    assert not zz.is_contiguous()
    (got_x,) = torch.autograd.grad((tac.angle(zz) * dev(c_np)).sum(), x)
    assert rel_err(host(got_x), want_x.numpy()) < 1e-3            # (1 / |z|^2 amplifies the forward's rounding at quiet bins)
    # db_to_amplitude
    d_np = (rng.standard_normal((4, 3, 50, 21)) * 30.0).astype(np.float32)
    g_np = rng.standard_normal(d_np.shape).astype(np.float32)
    for ref in (1.0, 3.5):
        dr = torch.from_numpy(d_np).requires_grad_(True)
        (want_d,) = torch.autograd.grad(torch_ref.db_to_amplitude(dr, ref), dr, torch.from_numpy(g_np))
        d = dev(d_np).requires_grad_(True)
        before = launches(tac)
        (got_d,) = torch.autograd.grad(tac.db_to_amplitude(d, ref), d, dev(g_np))
        assert launched_since(tac, before).get('tac_db_to_amplitude_backward_f32') == 1
        assert np.abs(host(got_d) - want_d.numpy()).max() < 2e-6 * np.abs(want_d.numpy()).max()
        dt = dev(d_np).transpose(-1, -2).requires_grad_(True)          # a non-contiguous dense layout
        (got_t,) = torch.autograd.grad(tac.db_to_amplitude(dt, ref), dt, dev(g_np).transpose(-1, -2))
        assert np.abs(host(got_t).swapaxes(-1, -2) - want_d.numpy()).max() < 2e-6 * np.abs(want_d.numpy()).max()


def test_phase_vocoder_gradient_kernel(tac):
    """d phase_vocoder / d spec on the gfx950 kernel (round 6; strict mode is on) against torch.autograd through the float64 oracle: rates
    below and above one (a source frame read by several / by no output frame), the strided pairs the STFT kernels return, a chain
    STFT -> TimeStretch -> ComplexNorm differentiated down to the waveform; the gradient of phase_advance is zero, as autograd finds."""
    rng = np.random.default_rng(91)
    n_freqs, n_frames, hop = 65, 47, 32
    z_np = rng.standard_normal((2, 3, n_freqs, n_frames, 2)).astype(np.float32)
    pa_np = np.linspace(0, np.pi * hop, n_freqs, dtype=np.float32)[:, None]
    for rate in (0.6, 1.0, 1.3, 2.5):
        zr = torch.from_numpy(z_np).double().requires_grad_(True)
        par = torch.from_numpy(pa_np).double().requires_grad_(True)
        outr = torch_ref.phase_vocoder(zr, rate, par)
        g_np = rng.standard_normal(tuple(outr.shape)).astype(np.float32)
        want, want_pa = torch.autograd.grad(outr, [zr, par], torch.from_numpy(g_np).double())
        assert float(want_pa.abs().max()) < 1e-9
        z = dev(z_np).requires_grad_(True)
        pa = dev(pa_np).requires_grad_(True)
        before = launches(tac)
        out = tac.phase_vocoder(z, rate, pa)
        got, got_pa = torch.autograd.grad(out, [z, pa], dev(g_np))
        assert launched_since(tac, before).get('tac_phase_vocoder_backward_f32') == 1
        assert float(got_pa.abs().max()) == 0.0
        assert rel_err(host(got), want.numpy()) < 2e-4, rate          # (1 / |z| amplifies float32 rounding at the smallest bins)
    x_np = signals.audio_like((2, 1, 5000), seed=92)
    xr = torch.from_numpy(x_np).double().requires_grad_(True)
    adv = torch.linspace(0, np.pi * 64, 129, dtype=torch.float64)[:, None]
    yr = torch_ref.complex_norm(torch_ref.phase_vocoder(torch_ref.stft(xr, 256, 64), 1.25, adv), 2.0)
    c_np = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    (want_x,) = torch.autograd.grad((yr * torch.from_numpy(c_np).double()).sum(), xr)
    x = dev(x_np).requires_grad_(True)
    chain = torch.nn.Sequential(tac.STFT(256, 64), tac.TimeStretch(64, 129, fixed_rate=1.25), tac.ComplexNorm(2.0)).cuda()
    (got_x,) = torch.autograd.grad((tac.realize(chain(x)) * dev(c_np)).sum(), x)
    assert rel_err(host(got_x), want_x.numpy()) < 1e-3


def test_hpss_gradient_kernel(tac):
    """d hpss / d mag on the gfx950 kernel (round 6; strict mode is on) against torch.autograd through the float64 oracle: soft masks at
    powers 1 / 2 / 0.7 with all four outputs, each output alone, the masks-only op, unequal widths, the strided spectrogram the STFT
    kernels return; hard masks (only harm = mag * mask and perc = mag * mask are differentiable)."""
    rng = np.random.default_rng(131)
    mag_np = (np.abs(rng.standard_normal((2, 2, 37, 29))) + 0.05).astype(np.float32)          # (continuous values: no ties in a window)
    for ks, power in (((5, 5), 2.0), ((7, 3), 1.0), ((9, 9), 0.7), ((3, 11), 2.0)):
        g_np = [rng.standard_normal(mag_np.shape).astype(np.float32) for _ in range(4)]
        mr = torch.from_numpy(mag_np).double().requires_grad_(True)
        outs_r = torch_ref.hpss(mr, ks[0], power) if ks[0] == ks[1] else None
        if outs_r is None:                                               # unequal widths: the documented behaviour, restated with torch ops
            kf, kt = ks
            pf = torch.nn.functional.pad(mr, (0, 0, kf // 2, kf // 2), mode='reflect')
            pt = torch.nn.functional.pad(mr, (kt // 2, kt // 2, 0, 0), mode='reflect')
            perc = pf.unfold(2, kf, 1).median(dim=-1)[0]
            harm = pt.unfold(3, kt, 1).median(dim=-1)[0]
            if power != 1.0:
                perc, harm = perc.pow(power), harm.pow(power)
            mh, mp = (harm + 1e-6) / (harm + perc + 1e-6), (perc + 1e-6) / (harm + perc + 1e-6)
            outs_r = (mr * mh, mr * mp, mh, mp)
        (want,) = torch.autograd.grad(outs_r, mr, [torch.from_numpy(g).double() for g in g_np], retain_graph=True)
        m = dev(mag_np).requires_grad_(True)
        before = launches(tac)
        outs = tac.hpss(m, ks if ks[0] != ks[1] else ks[0], power)
        (got,) = torch.autograd.grad(outs, m, [dev(g) for g in g_np])
        assert launched_since(tac, before).get('tac_hpss_backward_f32') == 1
        assert rel_err(host(got), want.numpy()) < 1e-4, (ks, power)
        for i in range(4):                                               # each output alone
            (w1,) = torch.autograd.grad(outs_r[i], mr, torch.from_numpy(g_np[i]).double(), retain_graph=True)
            (g1,) = torch.autograd.grad(tac.hpss(m, ks if ks[0] != ks[1] else ks[0], power)[i], m, dev(g_np[i]))
            assert rel_err(host(g1), w1.numpy()) < 1e-4, (ks, power, i)
    # masks only, on the strided spectrogram of the STFT kernels, down to the waveform
    x_np = signals.audio_like((2, 1, 4000), seed=132)
    xr = torch.from_numpy(x_np).double().requires_grad_(True)
    sr = torch_ref.complex_norm(torch_ref.stft(xr, 128, 32), 1.0)
    c_np = rng.standard_normal(tuple(sr.shape)).astype(np.float32)
    (want_x,) = torch.autograd.grad((torch_ref.hpss(sr, 5, 2.0)[2] * torch.from_numpy(c_np).double()).sum(), xr)
    x = dev(x_np).requires_grad_(True)
    spec = tac.Spectrogram(128, 32).cuda()(x)
    mh = tac.hpss(spec, 5, 2.0, mask_only=True)[2]
    (got_x,) = torch.autograd.grad((mh * dev(c_np)).sum(), x)
    assert rel_err(host(got_x), want_x.numpy()) < 1e-3
    # hard masks
    mr = torch.from_numpy(mag_np).double().requires_grad_(True)
    hr = torch_ref.hpss(mr, 5, 2.0, True)
    g0, g1 = rng.standard_normal(mag_np.shape).astype(np.float32), rng.standard_normal(mag_np.shape).astype(np.float32)
    (want_h,) = torch.autograd.grad([hr[0], hr[1]], mr, [torch.from_numpy(g0).double(), torch.from_numpy(g1).double()])
    m = dev(mag_np).requires_grad_(True)
    hh = tac.hpss(m, 5, 2.0, True)
    (got_h,) = torch.autograd.grad([hh[0], hh[1]], m, [dev(g0), dev(g1)])
    assert rel_err(host(got_h), want_h.numpy()) < 1e-6


def test_g8_hpss(tac, golden):
    """hpss on the HIP kernel: golden outputs of the reference (the medians select existing values, so
    harm / perc are exact up to pow and the mask quotient), the strided spectrogram the STFT kernels return, unequal
    filter widths against a numpy restatement of the documented behaviour, and the stock-torch route for wide kernels."""
    from test_oracle_golden import HPSS_CASES, hpss_input
    g = golden('g8_hpss')
    mag = hpss_input()
    for k, power, hard in HPSS_CASES:
        before = launches(tac)
        res = tac.hpss(dev(mag), k, power, hard)
        assert launched_since(tac, before) == {'tac_hpss_f32': 1}
        tag = 'k%d_p%g_%s' % (k, power, 'hard' if hard else 'soft')
        for name, r in zip(('harm', 'perc', 'mask_harm', 'mask_perc'), res):
            want = g[tag + '_' + name].astype(np.float32)
            assert r.dtype == (torch.bool if hard and name.startswith('mask') else torch.float32)
            tol = (1e-6 if power in (1.0, 2.0) else 2e-6) * max(1.0, np.abs(want).max())     # powf within a few ulp
            assert np.abs(host(r).astype(np.float32) - want).max() <= tol, (tag, name)
    assert tac.HPSS(7, 1.0, mask_only=True).cuda()(dev(mag))[0] is None
    # frame-major strided input (what Spectrogram returns) and unequal widths
    x = dev(signals.audio_like((2, 1, 12000), seed=53))
    spec = tac.Spectrogram(256, 64).cuda()(x)
    assert not spec.is_contiguous()
    h, p, mh, mp = tac.hpss(spec, (5, 9), 1.0)
    s = host(spec).astype(np.float64)
    padf = np.pad(s, ((0, 0), (0, 0), (2, 2), (0, 0)), mode='reflect')
    padt = np.pad(s, ((0, 0), (0, 0), (0, 0), (4, 4)), mode='reflect')
    perc = np.median(np.stack([padf[:, :, i:i + s.shape[2]] for i in range(5)], -1), -1)
    harm = np.median(np.stack([padt[..., i:i + s.shape[3]] for i in range(9)], -1), -1)
    want_mh = (harm + 1e-6) / (harm + perc + 1e-6)
    assert np.abs(host(mh) - want_mh).max() < 1e-6 and np.abs(host(h) - s * want_mh).max() < 1e-6 * s.max()
    assert mh.stride() == spec.stride()
    tac.set_strict(False)
    try:
        tac._ops._warned.clear()
        with pytest.warns(tac.CompositeRouteWarning, match='kernel_size'):
            wide = tac.hpss(dev(mag), 65, 2.0)
        want = torch_ref.hpss(torch.from_numpy(mag), 65, 2.0)
        assert np.abs(host(wide[2]) - want[2].numpy()).max() < 1e-6
    finally:
        tac.set_strict(True)


def test_hpss_tile_kernel_every_width_both_layouts_and_nan(tac):
    """The 64 x 64 tile kernel (equal odd widths 9 ... 31, shared-sort medians of eight windows, csrc/hpss.hip) against a numpy restatement
    of beta_hpss.py:104-127: every width it is instantiated for, sizes that are not multiples of the tile (and smaller
    than it), the contiguous (F, T) layout and the frame-major strided layout the STFT kernels return; medians select
    existing values, so the enhanced spectrograms are compared exactly.  A NaN poisons exactly the windows that hold
    it, as torch.median does (the CPU route of the same call is the witness)."""
    rng = np.random.default_rng(7)

    def ref_medians(s, k):
        h = k // 2
        padf = np.pad(s, ((0, 0), (h, h), (0, 0)), mode='reflect')
        padt = np.pad(s, ((0, 0), (0, 0), (h, h)), mode='reflect')
        perc = np.sort(np.stack([padf[:, i:i + s.shape[1]] for i in range(k)], -1), -1)[..., h]
        harm = np.sort(np.stack([padt[..., i:i + s.shape[2]] for i in range(k)], -1), -1)[..., h]
        return harm, perc

    for k, (rows, F, T) in zip(range(9, 33, 2), [(2, 70, 131), (1, 16, 200), (3, 129, 64), (1, 65, 65), (2, 40, 33),
                                                  (1, 257, 90), (2, 64, 64), (1, 100, 17), (1, 33, 300), (2, 127, 129),
                                                  (1, 513, 70), (2, 150, 97)]):
        if k // 2 >= min(F, T):
            continue
        s = (rng.random((rows, F, T), dtype=np.float32) * rng.integers(1, 4, (rows, F, T))).astype(np.float32)   # ties included
        harm, perc = ref_medians(s, k)
        for layout in ('contiguous', 'frame-major'):
            x = dev(s) if layout == 'contiguous' else dev(np.ascontiguousarray(s.transpose(0, 2, 1))).transpose(1, 2)
            assert x.is_contiguous() == (layout == 'contiguous')
            before = launches(tac)
            h, p, mh, mp = tac.hpss(x, k, 1.0, False)
            assert launched_since(tac, before) == {'tac_hpss_f32': 1}
            assert mh.stride() == x.stride()
            want_mh = (harm + np.float32(1e-6)) / (harm + perc + np.float32(1e-6))
            want_mp = (perc + np.float32(1e-6)) / (harm + perc + np.float32(1e-6))
            assert np.abs(host(mh) - want_mh).max() <= 2e-7 and np.abs(host(mp) - want_mp).max() <= 2e-7, (k, layout)
            assert np.abs(host(h) - s * want_mh).max() <= 4e-7 * s.max(), (k, layout)
            hard = tac.hpss(x, k, 2.0, True)
            assert np.array_equal(host(hard[2]), harm > perc) and np.array_equal(host(hard[3]), harm < perc), (k, layout)
    # NaN: the windows that contain it, and only those
    s = rng.random((1, 90, 80), dtype=np.float32)
    s[0, 40, 33] = np.nan
    s[0, 2, 70] = np.nan
    for k in (9, 31):
        got = tac.hpss(dev(s), k, 2.0, False)
        want = tac.hpss(torch.from_numpy(s), k, 2.0, False)                  # CPU route: torch.median
        for a, b in zip(got, want):
            assert np.array_equal(np.isnan(host(a)), np.isnan(b.numpy())), k
            ok = ~np.isnan(b.numpy())
            assert np.abs(host(a)[ok] - b.numpy()[ok]).max() <= 1e-6
        got5 = tac.hpss(dev(s), (5, 9), 2.0, False)                          # unequal widths: the two-launch route
        want5 = tac.hpss(torch.from_numpy(s), (5, 9), 2.0, False)
        assert np.array_equal(np.isnan(host(got5[2])), np.isnan(want5[2].numpy()))


def test_hpss_unequal_and_small_widths_both_layouts_and_nan(tac):
    """Unequal or small filter widths (csrc/hpss.hip: hpss_axis_a_kernel + hpss_axis_b_kernel, one halo axis each, the first
    launch's medians parked in the mask_perc buffer) against the same numpy restatement of beta_hpss.py:104-127, both
    layouts, sizes off the tile grid; medians select existing values, so hard masks are compared exactly."""
    rng = np.random.default_rng(11)
    cases = [((1, 3), (2, 70, 131)), ((3, 1), (1, 65, 64)), ((5, 9), (2, 129, 200)), ((7, 31), (1, 257, 90)),
             ((31, 9), (2, 64, 33)), ((13, 5), (1, 100, 17)), ((3, 3), (1, 513, 70)), ((1, 1), (1, 9, 9)), ((29, 31), (1, 40, 150)),
             # widths 33 ... 63 (round 5: the same two launches, a 32-column halo): equal and unequal, shortest legal axes
             ((33, 33), (2, 70, 131)), ((63, 63), (1, 129, 200)), ((41, 5), (1, 257, 90)), ((7, 47), (2, 64, 33)),
             ((63, 33), (1, 32, 17)), ((55, 61), (1, 100, 300)), ((37, 63), (1, 513, 32))]
    for (kf, kt), (rows, F, T) in cases:
        s = (rng.random((rows, F, T), dtype=np.float32) * rng.integers(1, 4, (rows, F, T))).astype(np.float32)
        padf = np.pad(s, ((0, 0), (kf // 2, kf // 2), (0, 0)), mode='reflect')
        padt = np.pad(s, ((0, 0), (0, 0), (kt // 2, kt // 2)), mode='reflect')
        perc = np.sort(np.stack([padf[:, i:i + F] for i in range(kf)], -1), -1)[..., kf // 2]
        harm = np.sort(np.stack([padt[..., i:i + T] for i in range(kt)], -1), -1)[..., kt // 2]
        for layout in ('contiguous', 'frame-major'):
            x = dev(s) if layout == 'contiguous' else dev(np.ascontiguousarray(s.transpose(0, 2, 1))).transpose(1, 2)
            before = launches(tac)
            h, p, mh, mp = tac.hpss(x, (kf, kt), 1.0, False)
            assert launched_since(tac, before) == {'tac_hpss_f32': 1}
            assert mh.stride() == x.stride()
            want_mh = (harm + np.float32(1e-6)) / (harm + perc + np.float32(1e-6))
            want_mp = (perc + np.float32(1e-6)) / (harm + perc + np.float32(1e-6))
            assert np.abs(host(mh) - want_mh).max() <= 2e-7 and np.abs(host(mp) - want_mp).max() <= 2e-7, (kf, kt, layout)
            assert np.abs(host(h) - s * want_mh).max() <= 4e-7 * s.max() and np.abs(host(p) - s * want_mp).max() <= 4e-7 * s.max()
            hard = tac.hpss(x, (kf, kt), 2.0, True)
            assert np.array_equal(host(hard[2]), harm > perc) and np.array_equal(host(hard[3]), harm < perc), (kf, kt, layout)
    s = rng.random((1, 90, 80), dtype=np.float32)
    s[0, 40, 33] = np.nan
    s[0, 2, 70] = np.nan
    for ks in ((5, 9), (31, 3), (1, 7), (45, 45), (63, 35)):
        got = tac.hpss(dev(s), ks, 2.0, False)
        want = tac.hpss(torch.from_numpy(s), ks, 2.0, False)                 # CPU route: torch.median
        for a, b in zip(got, want):
            assert np.array_equal(np.isnan(host(a)), np.isnan(b.numpy())), ks
            ok = ~np.isnan(b.numpy())
            assert np.abs(host(a)[ok] - b.numpy()[ok]).max() <= 1e-6


def test_hpss_overlapping_buffers_are_refused(tac):
    """include/tac_amd.h (10): outputs overlapping the input or one another are an error (TAC_E_INVALID), not silent corruption —
    the two-launch route of unequal widths parks its first medians in mask_perc."""
    import ctypes
    P = ctypes.c_void_p
    lib = tac._native.lib()
    x = torch.rand(2, 40, 50, device='cuda')
    a, b = torch.empty_like(x), torch.empty_like(x)
    stream = tac._native.stream_ptr(x.device)
    args = lambda mh, mp: (P(x.data_ptr()), 2, 40, 50, 2000, 50, 1, 9, 5, 2.0, 0, None, None, P(mh), P(mp), stream)
    assert lib.tac_hpss_f32(*args(a.data_ptr(), b.data_ptr())) == tac._native.TAC_OK
    assert lib.tac_hpss_f32(*args(a.data_ptr(), x.data_ptr())) == tac._native.TAC_E_INVALID          # mask_perc is the input
    assert lib.tac_hpss_f32(*args(a.data_ptr(), a.data_ptr() + 400)) == tac._native.TAC_E_INVALID    # the two masks overlap
    torch.cuda.synchronize()


def test_float64_long_non_smooth_length_is_a_composite_route(tac):
    """Since round 4 the float64 route takes even lengths whose half is 5-smooth (<= 8192) and any length <= 512; other float64
    lengths (here 1001) go to torch's GPU operators: an error under strict mode, a CompositeRouteWarning otherwise — and the
    result is still the reference's."""
    x = torch.from_numpy(signals.audio_like((2, 1, 6000), seed=91)).double().cuda()
    win = torch.hann_window(1001, dtype=torch.float64)
    with pytest.raises(RuntimeError, match='strict mode'):
        tac.stft(x, 1001, 250, window=win.cuda())
    tac.set_strict(False)
    try:
        tac._ops._warned.clear()
        with pytest.warns(tac.CompositeRouteWarning):
            z = tac.stft(x, 1001, 250, window=win.cuda())
        assert z.dtype == torch.float64
        assert rel_err(host(z), torch_ref.stft(x.cpu(), 1001, 250, window=win).numpy()) < 1e-12
    finally:
        tac.set_strict(True)


def test_hpss_mask_only_skips_the_masked_spectrograms(tac):
    """mask_only=True (beta_hpss.py:123-124) takes the `hpss_masks` op: the same kernels with NULL harm / perc pointers, i.e.
    without the two masked-spectrogram stores; masks bit-equal to the full call, equal widths (tile kernel) and unequal
    (two launches), soft and hard, both layouts, and through the HPSS layer."""
    rng = np.random.default_rng(13)
    s = rng.random((2, 130, 97), dtype=np.float32)
    for layout in ('contiguous', 'frame-major'):
        x = dev(s) if layout == 'contiguous' else dev(np.ascontiguousarray(s.transpose(0, 2, 1))).transpose(1, 2)
        for ks in (31, 9, (5, 9), (3, 31)):
            for hard in (False, True):
                full = tac.hpss(x, ks, 2.0, hard)
                before = launches(tac)
                only = tac.hpss(x, ks, 2.0, hard, True)
                assert launched_since(tac, before) == {'tac_hpss_f32': 1}
                assert only[0] is None and only[1] is None
                assert torch.equal(only[2], full[2]) and torch.equal(only[3], full[3]), (layout, ks, hard)
                assert only[2].stride() == x.stride()
    layer = tac.HPSS(17, 1.0, mask_only=True).cuda()
    out = layer(dev(s))
    assert out[0] is None and torch.equal(out[2], tac.hpss(dev(s), 17, 1.0)[2])


def test_g10_melspectrogram_fft_length_4096(tac, golden):
    """Melspectrogram (-> AmplitudeToDb) at fft_length 4096 against the reference's outputs (golden g10): ONE launch since
    round 6 — the twelve-wave form of the 4096 kernel runs both 1024-point transforms through one exchange area, which leaves
    the LDS for the bank's 42 KB of weights beside eleven waves (rounds 2 - 5: two launches, the |X|^p rows crossing HBM).  A
    spectrogram the caller made first still takes the standalone band-sparse filterbank kernel on 2049-bin rows."""
    g = golden('g10_mel4096')
    x = dev(signals.audio_like((2, 2, 30000), seed=71))
    mel = tac.Melspectrogram(num_mels=128, sample_rate=44100, fft_length=4096, hop_length=1024).cuda()
    before = launches(tac)
    got = tac.realize(mel(x))
    assert launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
    assert rel_err(host(got), g['mel']) < 1e-5
    chain = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
    before = launches(tac)
    got_db = chain(x)
    assert launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
    assert np.abs(host(got_db) - g['mel_db']).max() < DB_ABS
    # the unfused pieces: the spectrogram rows, then the standalone filterbank kernel (with and without the dB epilogue)
    spec = tac.realize(tac.Spectrogram(4096, hop_length=1024, power=2.0).cuda()(x))
    before = launches(tac)
    got2 = tac.apply_filterbank(spec, mel[-1].filterbank)
    assert launched_since(tac, before) == {'tac_apply_filterbank_sparse_f32': 1}
    assert rel_err(host(got2), g['mel']) < 1e-5
    mel80 = tac.Melspectrogram(num_mels=80, sample_rate=48000, fft_length=4096, hop_length=1024, htk=True, min_freq=50.0).cuda()
    assert rel_err(host(tac.realize(mel80(x))), g['mel80_htk']) < 1e-5
    # a longer input: more frames than waves, rows of different phase
    xl = signals.audio_like((3, 1, 250000), seed=72)
    want = torch_ref.melspectrogram_db(torch.from_numpy(xl), num_mels=128, sample_rate=44100, n_fft=4096, hop=1024).numpy()
    assert np.abs(host(chain(dev(xl))) - want).max() < DB_ABS


def test_melspectrogram_2048_common_banks_take_the_band_sparse_kernel(tac):
    """Banks of 40 ... 100 bands at fft_length 2048 (round 6): their lane tables (24 - 40 steps) fit beside the twelve waves of
    melspec_stream3_kernel, but until round 6 the packer asked the older two-waves-per-SIMD kernel's LDS formula and refused them, so
    the 2.1 x slower MFMA form ran.  Banks whose band count is not a multiple of 64 are laid out from their widest end (info[2] carries
    ST_REV_MARK = 256): outputs must land on their own bands.  Against the float64 oracle, float32 and int16 PCM input."""
    x = signals.audio_like((3, 1, 30000), seed=620)
    for n_mels, sr, rev in ((40, 16000, True), (64, 16000, False), (80, 16000, True), (80, 22050, True), (100, 22050, True), (160, 16000, True)):
        chain = torch.nn.Sequential(*tac.Melspectrogram(num_mels=n_mels, sample_rate=sr, fft_length=2048, hop_length=512),
                                    tac.AmplitudeToDb()).cuda()
        pack = tac._hip._melbank_pack(chain[2].filterbank, 2048)
        assert pack is not None and (int(pack[2][2]) == 64 + (256 if rev else 0)), (n_mels, [int(v) for v in pack[2]])
        before = launches(tac)
        got = host(chain(dev(x)))
        assert launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
        fb = chain[2].filterbank.double().cpu().numpy()
        mel = np.einsum('...ft,fm->...mt', np.abs(numpy_ref.stft(x, 2048, 512)) ** 2, fb)
        want = 10.0 * np.log10(np.maximum(mel ** 2, 1e-7))
        live = mel > 1e-5 * mel.max()
        assert np.abs(got - want)[live].max() < DB_ABS, n_mels
        lin = host(tac.realize(torch.nn.Sequential(*list(chain)[:3])(dev(x))))
        assert rel_err(lin, mel) < 1e-5, n_mels
    # int16 PCM through the coded-input form of the same kernel, 80 bands (reversed cells)
    pcm = np.round(x * 32767.0).astype(np.int16)
    chain = torch.nn.Sequential(*tac.Melspectrogram(num_mels=80, sample_rate=16000, fft_length=2048, hop_length=512)).cuda()
    got = host(tac.realize(chain(torch.from_numpy(pcm).cuda())))
    fb = chain[2].filterbank.double().cpu().numpy()
    want = np.einsum('...ft,fm->...mt', np.abs(numpy_ref.stft(pcm.astype(np.float64) / 32768.0, 2048, 512)) ** 2, fb)
    assert rel_err(got, want) < 1e-5


def test_melspectrogram_4096_one_launch_geometries(tac):
    """The one-launch chain at fft_length 4096 (csrc/stft_n4096_s3.hpp) against the float64 oracle over what its launcher and its
    table builder decide on: band counts that fill one to four lane slots, both powers, no centring, every padding mode, a hop
    that shares two of four hops, frames the wave count does not divide, rows of exactly one frame — and the geometries it declines
    (frames that are not 16-byte aligned, rows shorter than a frame), which take the two-launch chain with the same results."""
    def check(x, n_mels, sr, hop, power, expect_fused, db=True, **kw):
        layers = list(tac.Melspectrogram(num_mels=n_mels, sample_rate=sr, fft_length=4096, hop_length=hop, **kw))
        layers[1] = tac.ComplexNorm(power)
        chain = torch.nn.Sequential(*layers, *([tac.AmplitudeToDb()] if db else [])).cuda()
        before = launches(tac)
        got = host(tac.realize(chain(dev(x))))
        calls = launched_since(tac, before)
        assert (calls == {'tac_melspec_sparse_f32': 1}) == expect_fused, (calls, n_mels, hop, kw)
        fb = layers[2].filterbank.double().cpu().numpy()
        p = np.abs(numpy_ref.stft(x, 4096, hop, center=kw.get('center', True), pad_mode=kw.get('pad_mode', 'reflect'))) ** power
        mel = np.einsum('...ft,fm->...mt', p, fb)
        if db:
            want = 10.0 * np.log10(np.maximum(mel ** 2, 1e-7))
            live = mel > 1e-5 * mel.max()                                   # (below that the fp32 FFT's own rounding decides the value)
            assert np.abs(got - want)[live].max() < DB_ABS
        else:
            assert rel_err(got, mel) < 1e-5
    x = signals.audio_like((2, 3, 41000), seed=611)
    for n_mels, sr in ((40, 16000), (64, 22050), (80, 44100), (128, 44100), (200, 48000), (256, 48000)):
        check(x, n_mels, sr, 1024, 2.0, True)
    check(x, 128, 48000, 1024, 1.0, True)
    check(x, 128, 48000, 1024, 2.0, True, db=False)
    check(x, 128, 48000, 2048, 2.0, True, center=False)
    for mode in ('constant', 'replicate', 'circular'):
        check(x, 80, 44100, 512, 2.0, True, pad_mode=mode)
    check(signals.audio_like((1, 1, 4096), seed=612), 128, 44100, 1024, 2.0, True)          # five frames, four of them padded
    check(signals.audio_like((5, 1, 4096 + 1024 * 13), seed=613), 128, 44100, 1024, 2.0, True, center=False)   # 14 frames per row
    check(x, 128, 48000, 1000, 2.0, True)                                   # hop a multiple of four samples: still aligned
    check(x, 128, 48000, 1023, 2.0, False)                                  # frames off the 16-byte grid: the two-launch chain
    check(signals.audio_like((2, 1, 3000), seed=614), 128, 44100, 1024, 2.0, False)          # rows shorter than one frame


def test_compiled_binding_carries_the_fused_call(tac):
    """The steady state of the reference idiom launches through the compiled binding (one C++ call: layout and stamp checks,
    allocation, stream, tac_melspec_sparse_f32) — same bits as the ctypes launcher of the same plan, also through the
    dispatcher op, and a plan stops matching when its filterbank changes."""
    from torchaudio_contrib_amd import _lazy
    assert tac._native.binding() == 'compiled'
    x = dev(signals.audio_like((3, 1, 30000), seed=501))
    model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                                tac.AmplitudeToDb()).cuda()
    y0 = model(x)                                                     # general path + plan
    plans = [p for p in _lazy._plans.values() if type(p) is not tuple and p.fb is model[2].filterbank]
    assert len(plans) == 1 and plans[0].cplan is not None
    plan = plans[0]
    n0, before = plan.cplan.launches, launches(tac)
    y1 = model(x)
    assert plan.cplan.launches == n0 + 1 and launched_since(tac, before) == {'tac_melspec_sparse_f32': 1}
    assert type(y1) is torch.Tensor and y1.stride() == y0.stride() and torch.equal(y1, y0)
    assert torch.equal(plan.launch(x), y0)                            # the ctypes launcher of the same plan
    assert torch.equal(torch.ops.tac_amd.melspec_planned(x, plan.cplan.register()), y0)     # ... and the dispatcher op
    meta = torch.ops.tac_amd.melspec_planned(x.to('meta'), plan.cplan.register())
    assert meta.shape == y0.shape and meta.stride() == y0.stride()
    assert plan.cplan.launch(x[:2]) is None                           # another layout: not this plan's call
    model[2].filterbank.mul_(2.0)                                     # recorded by the version counter: the plan stops matching
    assert plan.cplan.launch(x) is None
    y2 = model(x)
    fresh = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                                tac.AmplitudeToDb()).cuda()
    fresh[2].filterbank.mul_(2.0)
    assert torch.equal(y2, fresh(x)) and not torch.equal(y2, y0)      # the new contents, not the stale tables


def test_tables_follow_the_filterbank_and_window(tac):
    """The packed weights / plans / transposes derived from a filterbank or window are cached on the tensor, stamped with
    its version counter and data pointer: every in-place torch op is seen.  A write PyTorch itself does not record —
    ``fb.data.mul_(2)`` (``.data`` has a version counter of its own) — needs ``tac.invalidate(fb)``; after it the kernels
    use the new values (without it they would keep the stale tables: the documented limit of the cheap stamps)."""
    x = dev(signals.audio_like((2, 1, 20000), seed=341))
    mel = tac.Melspectrogram(num_mels=64, sample_rate=16000, fft_length=2048, hop_length=512).cuda()
    fb = mel[2].filterbank
    y0 = host(tac.realize(mel(x)))
    fb.mul_(2.0)                                                      # recorded by the version counter
    assert rel_err(host(tac.realize(mel(x))), 2.0 * y0) < 1e-6
    fb.data.mul_(0.5)                                                 # NOT recorded
    assert fb._version == mel[2].filterbank._version
    tac.invalidate(fb)
    assert rel_err(host(tac.realize(mel(x))), y0) < 1e-6
    win = mel[0].window
    win.data.mul_(3.0)
    tac.invalidate()                                                  # everything
    assert rel_err(host(tac.realize(mel(x))), 9.0 * y0) < 1e-6
    fb.data.mul_(2.0)
    tac.invalidate(mel[2].filterbank.data)                            # through an alias that carries no tables: everything goes
    assert rel_err(host(tac.realize(mel(x))), 18.0 * y0) < 1e-6
    spec = tac.Spectrogram(2048, 512, power=2.).cuda()(x)
    fb2 = tac.create_mel_filter(1025, 40, 0.0, 8000.0, False).cuda()
    a = host(tac.apply_filterbank(spec, fb2))
    fb2.data.mul_(4.0)
    pack_of_fb = fb._tac_pack                                         # (another tensor's tables ...)
    tac.invalidate(fb2)
    assert rel_err(host(tac.apply_filterbank(spec, fb2)), 4.0 * a) < 1e-6
    tac.realize(mel(x))
    assert fb._tac_pack is pack_of_fb                                 # ... survive invalidate(fb2): no global epoch bump
    # layers built under inference_mode hold tensors without a version counter: no caching for those, but they work
    with torch.inference_mode():
        mel_inf = tac.Melspectrogram(num_mels=64, sample_rate=16000, fft_length=2048, hop_length=512).cuda()
        assert mel_inf[2].filterbank.is_inference()
        y_inf = host(tac.realize(mel_inf(x)))
        assert rel_err(y_inf, y0) < 1e-6                             # (a fresh layer: the original window and bank)
        mel_inf[2].filterbank.mul_(2.0)                               # an in-place write nothing records
        assert rel_err(host(tac.realize(mel_inf(x))), 2.0 * y_inf) < 1e-6


def test_coded_waveforms_fused_into_the_frame_load(tac, golden):
    """int16 PCM and 8-bit mu-law codes (uint8 or the int64 mu_law_encoding returns) are converted in
    registers inside the fused kernel's frame load — ONE launch, the decoded waveform never exists — and agree with the
    reference chain MuLawDecoding -> Melspectrogram -> AmplitudeToDb (golden g9), at fft_length 2048 and (round 3) 256 / 400 /
    512 / 1024; other fft sizes convert first."""
    g = golden('g9_mulaw_mel')
    for dtype in (torch.int64, torch.uint8):
        codes = torch.from_numpy(g['codes']).to(dtype).cuda()
        for n_fft, hop, mels, fused in ((2048, 512, 128, True), (512, 128, 40, True)):
            model = torch.nn.Sequential(tac.MuLawDecoding(256),
                                        *tac.Melspectrogram(num_mels=mels, sample_rate=16000, fft_length=n_fft, hop_length=hop),
                                        tac.AmplitudeToDb()).cuda()
            before = launches(tac)
            y = model(codes)
            ran = launched_since(tac, before)
            assert type(y) is torch.Tensor
            if fused:
                assert ran == {'tac_melspec_sparse_coded_f32': 1}, ran
            else:
                assert 'tac_melspec_sparse_f32' in ran and any(k.startswith('tac_mulaw_decode') for k in ran), ran
            assert np.abs(host(y) - g['mel_db_n%d' % n_fft]).max() < DB_ABS
    # a decoded waveform used by anything else is an ordinary tensor with the table's bits
    w = tac.MuLawDecoding(256)(torch.from_numpy(g['codes']).cuda())
    assert isinstance(w, tac.DeferredWave) and np.array_equal(host(w + 0.0).view(np.uint32),
                                                              golden('g5_mulaw')['lut256'].view(np.uint32)[g['codes']])
    # int16 PCM: every frame position incl. the reflected edges, odd row offsets (unaligned pairs -> gather path), vs the
    # float32 path on sample * 2^-15
    pcm = (signals.audio_like((3, 2, 30001), seed=63) * 25000).astype(np.int16)
    mel = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                              tac.AmplitudeToDb()).cuda()
    before = launches(tac)
    got = mel(dev(pcm))
    assert launched_since(tac, before) == {'tac_melspec_sparse_coded_f32': 1}
    want = mel(dev(pcm.astype(np.float32) / 32768.0))
    assert np.abs(host(got) - host(want)).max() < 1e-4
    odd = dev(pcm)[:, :, 1:]                                            # row base no longer 4-byte aligned
    assert np.abs(host(mel(odd)) - host(mel(dev(pcm.astype(np.float32)[:, :, 1:] / 32768.0)))).max() < 1e-4
    # the same at the sizes of the 2 / 4 / 8-frames-per-wave kernels and the mixed-radix 400 one
    for n_fft, hop, mels in ((1024, 256, 80), (512, 128, 80), (256, 64, 40), (400, 160, 80)):
        chain = torch.nn.Sequential(*tac.Melspectrogram(num_mels=mels, sample_rate=16000, fft_length=n_fft, hop_length=hop),
                                    tac.AmplitudeToDb()).cuda()
        before = launches(tac)
        got = chain(dev(pcm))
        ran = launched_since(tac, before)
        assert ran == {'tac_melspec_sparse_coded_f32': 1}, (n_fft, ran)
        assert np.abs(host(got) - host(chain(dev(pcm.astype(np.float32) / 32768.0)))).max() < 1e-4, n_fft
        assert np.abs(host(chain(odd)) - host(chain(dev(pcm.astype(np.float32)[:, :, 1:] / 32768.0)))).max() < 1e-4, n_fft
    # mu-law codes at the speech configuration (400 / 160 / 40 mels), both storage types, against decode-then-float32
    codes = torch.from_numpy(g['codes'])
    for dtype in (torch.int64, torch.uint8):
        chain = torch.nn.Sequential(tac.MuLawDecoding(256),
                                    *tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=400, hop_length=160),
                                    tac.AmplitudeToDb()).cuda()
        before = launches(tac)
        got = chain(codes.to(dtype).cuda())
        assert launched_since(tac, before) == {'tac_melspec_sparse_coded_f32': 1}
        wave = tac.MuLawDecoding(256)(codes.cuda()) + 0.0
        assert np.abs(host(got) - host(chain[1:](wave))).max() < 1e-4
        assert np.abs(host(chain(codes.to(dtype).cuda()[..., 3:])) - host(chain[1:](wave[..., 3:].contiguous()))).max() < 1e-4
    # rows shorter than a frame (every frame gathers): the coded kernels decline, the chain converts first — same values
    tiny = (signals.audio_like((2, 1, 333), seed=64) * 20000).astype(np.int16)
    for n_fft, hop in ((400, 160), (512, 128)):
        chain = torch.nn.Sequential(*tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=n_fft, hop_length=hop),
                                    tac.AmplitudeToDb()).cuda()
        assert np.abs(host(chain(dev(tiny))) - host(chain(dev(tiny.astype(np.float32) / 32768.0)))).max() < 1e-4, n_fft
    z = tac.stft(dev(pcm), 512, 128)                                    # no coded frame load there: converted by a kernel first
    assert rel_err(host(z), host(tac.stft(dev(pcm.astype(np.float32) / 32768.0), 512, 128))) < 1e-6


# ------------------------------------------------------------------ size-independent properties at BASELINE sizes
def test_cfg2_full_size_properties(tac):
    """cfg-2 (256x1x160000, 2048/512/128): linearity of the power-mel map in amplitude², agreement of
    fused vs unfused evaluation, frame-shift consistency, and a spot check of rows against the oracle."""
    torch.manual_seed(0)
    x = torch.rand(256, 1, 160000, device='cuda') * 2 - 1
    mel = tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512).cuda()
    y = mel(x)
    assert tuple(y.shape) == (256, 1, 128, 313)
    assert torch.isfinite(y).all()
    y2 = mel(x * 0.5)
    assert rel_err(host(y2), host(y) * 0.25) < 1e-5                  # power scales with gain²
    tac.set_lazy_fusion(False)
    try:
        y3 = mel(x)
    finally:
        tac.set_lazy_fusion(True)
    assert rel_err(host(y3), host(y)) < 1e-5                         # fused == op-by-op
    # shifting the signal by one hop shifts interior frames by one
    ys = mel(x[..., 512:])
    assert rel_err(host(ys[..., 2:300]), host(y[..., 3:301])) < 1e-5
    rows = [0, 77, 255]
    want = torch_ref.melspectrogram_db(x[rows].cpu(), n_fft=2048, hop=512, num_mels=128, sample_rate=16000)
    chain = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
    got = host(chain(x))[rows]
    assert np.abs(got - want.numpy()).max() < DB_ABS


def test_cfg4_multichannel_stress_slice(tac):
    """cfg-4 shape family (8 channels, 4096/1024) at a length the oracle finishes in seconds."""
    x = signals.audio_like((2, 8, 120000), seed=21)
    got = tac.Spectrogram(4096, hop_length=1024).cuda()(dev(x))
    want = torch_ref.spectrogram(torch.from_numpy(x), 4096, 1024).numpy()
    assert got.shape == want.shape
    assert rel_err(host(got), want) < TIGHT


def test_cfg3_per_gpu_shard_full_size(tac):
    """cfg-3 per-GPU shard (256 rows x 44.1 kHz x 30 s, 2048/512/128 + dB): one launch over 1.35 GB of input,
    32-bit tile indices and 64-bit byte offsets hold, rows spot-checked against the oracle."""
    torch.manual_seed(3)
    x = torch.rand(256, 1, 1323000, device='cuda') * 2 - 1
    chain = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=44100, fft_length=2048,
                                                    hop_length=512), tac.AmplitudeToDb()).cuda()
    y = tac.realize(chain(x))
    assert tuple(y.shape) == (256, 1, 128, 2584)
    assert torch.isfinite(y).all()
    rows = [0, 255]
    want = torch_ref.melspectrogram_db(x[rows].cpu(), n_fft=2048, hop=512, num_mels=128, sample_rate=44100)
    assert np.abs(host(y[rows]) - want.numpy()).max() < DB_ABS
    # batch-axis sharding property: any row block equals the same rows of the whole-batch result
    y_blk = tac.realize(chain(x[100:132]))
    assert torch.equal(y_blk, y[100:132])


def test_cfg4_full_size_multichannel(tac):
    """cfg-4 (64 x 8ch x 48 kHz x 60 s, STFT 4096/1024 + ComplexNorm): 5.9 GB in, 11.8 GB out, streamed in one
    launch with no padded or framed copy; first/last rows checked against the oracle."""
    torch.manual_seed(4)
    x = torch.rand(64, 8, 2880000, device='cuda') * 2 - 1
    mag = tac.Spectrogram(4096, hop_length=1024).cuda()(x)
    assert tuple(mag.shape) == (64, 8, 2049, 2813)
    for b, c in ((0, 0), (63, 7)):
        want = torch_ref.spectrogram(x[b, c][None].cpu(), 4096, 1024).numpy()[0]
        assert rel_err(host(mag[b, c]), want) < TIGHT
    del mag
    torch.cuda.empty_cache()


def test_cfg5_mulaw_roundtrip_full_size(tac, mulaw_oracle_pinned):
    """cfg-5: 1024x1x120000 @ n_quantize=256 — encode → decode → encode is idempotent and a checksum
    of the codes matches the oracle's on a slice."""
    torch.manual_seed(5)
    x = torch.rand(1024, 1, 120000, device='cuda') * 2 - 1
    c1 = tac.mu_law_encoding(x, 256)
    c2 = tac.mu_law_encoding(tac.mu_law_decoding(c1, 256), 256)
    assert torch.equal(c1, c2)
    assert int(c1.min()) >= 0 and int(c1.max()) <= 255
    if mulaw_oracle_pinned:
        sl = x[:8].cpu()
        assert torch.equal(c1[:8].cpu(), torch_ref.mu_law_encoding(sl, 256))


def test_state_dict_and_buffers_follow_device(tac):
    m = torch.nn.Sequential(*tac.Melspectrogram(fft_length=1024, hop_length=256, sample_rate=16000),
                            tac.AmplitudeToDb()).cuda()
    assert list(m.state_dict().keys()) == []
    assert m[0].window.is_cuda and m[2].filterbank.is_cuda
    assert list(m.parameters()) == []


def test_fft_length_400_kernel(tac, golden):
    """fft_length 400 (25 ms at 16 kHz) runs on its own mixed-radix kernel (csrc/stft_n400.hip, 200 = 8 x 25: a
    25-point transform per lane in registers, the 8-point one across lanes through DPP): complex rows, |X|, |X|^2 and
    their dB forms against the float64 restatement and the reference's golden vector; every pad mode, short / centred
    windows, odd lengths and strides (the sample-by-sample path), rows shorter than a unit; the forms the kernel does
    not have (two-sided, |X|^p) still take the DFT-matrix route."""
    base = signals.audio_like((1, 2, 20000), seed=4)
    before = launches(tac)
    z = tac.stft(dev(base[..., :6000]), 400, hop_length=160)
    assert launched_since(tac, before) == {'tac_stft_f32': 1}
    want = golden('g4_variants')['n400_h160']
    assert tuple(z.shape) == want.shape
    assert rel_err(host(z), want) < 5e-6
    cases = [((3, 2, 16000), 160, {}), ((2, 1, 16001), 160, dict(pad_mode='constant')),
             ((1, 3, 7777), 100, dict(pad_mode='replicate', win_length=320)),
             ((2, 2, 9000), 200, dict(pad_mode='circular', normalized=True)),
             ((1, 1, 5000), 160, dict(center=False)), ((5, 1, 999), 37, {}),          # odd hop: sample-by-sample path
             ((1, 1, 400), 160, {}), ((2, 1, 1100), 160, dict(center=False))]
    for shape, hop, kw in cases:
        x = signals.audio_like(shape, seed=41 + hop)
        before = launches(tac)
        got = host(tac.stft(dev(x), 400, hop_length=hop, **kw))
        assert launched_since(tac, before) == {'tac_stft_f32': 1}, (shape, hop, kw)
        ref = numpy_ref.stft(x, 400, hop, **kw)
        assert got.shape[:-1] == ref.shape, (shape, hop, kw)
        assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6, (shape, hop, kw)
    # a strided view of the rows (row stride odd: unaligned pairs)
    buf = dev(signals.audio_like((4, 12001), seed=77))
    view = buf[:, :12000]
    got = host(tac.stft(view, 400, hop_length=160))
    ref = numpy_ref.stft(host(view), 400, 160)
    assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6
    x = signals.audio_like((2, 2, 24000), seed=43)
    for power in (1.0, 2.0):
        before = launches(tac)
        spec = tac.Spectrogram(400, hop_length=160, power=power).cuda()
        got = host(spec(dev(x)))
        assert launched_since(tac, before) == {'tac_spectrogram_f32': 1}
        want_s = torch_ref.spectrogram(torch.from_numpy(x), 400, 160, power=power).numpy()
        assert rel_err(got, want_s) < 1e-5, power
        chain = torch.nn.Sequential(*tac.Spectrogram(400, hop_length=160, power=power), tac.AmplitudeToDb()).cuda()
        got_db = host(chain(dev(x)))
        want_db = torch_ref.amplitude_to_db(torch.from_numpy(want_s)).numpy()
        big = want_s > 1e-6 * want_s.max()
        assert np.abs(got_db - want_db)[big].max() < DB_ABS, power
    # the fused chain Melspectrogram (-> AmplitudeToDb) at fft_length 400: one launch
    for shape, n_mels, hop, kw in (((2, 2, 24000), 80, 160, {}), ((3, 1, 16001), 40, 160, dict(pad_mode='constant')),
                                   ((1, 1, 5000), 128, 100, {}), ((4, 1, 3000), 5, 77, {}), ((1, 2, 8000), 23, 160, dict(center=False))):
        xm = signals.audio_like(shape, seed=50 + n_mels)
        mel = tac.Melspectrogram(num_mels=n_mels, sample_rate=16000, fft_length=400, hop_length=hop, **kw).cuda()
        before = launches(tac)
        got = host(tac.realize(mel(dev(xm))))
        # (fewer than eight bands, or bands wider than the fused form's 48 taps: the three-kernel chain)
        fused = {'tac_melspec_sparse_f32': 1} if n_mels not in (5, 23) else {'tac_spectrogram_f32': 1, 'tac_apply_filterbank_sparse_f32': 1}
        assert launched_since(tac, before) == fused, (shape, n_mels)
        want_m = torch_ref.melspectrogram(torch.from_numpy(xm), num_mels=n_mels, sample_rate=16000, n_fft=400, hop=hop,
                                          **kw).numpy()
        assert got.shape == want_m.shape
        assert rel_err(got, want_m) < 1e-5, (shape, n_mels)
        chain = torch.nn.Sequential(*mel, tac.AmplitudeToDb()).cuda()
        before = launches(tac)
        got_db = host(chain(dev(xm)))
        assert ('tac_melspec_sparse_f32' in launched_since(tac, before)) == (n_mels not in (5, 23))
        want_db = torch_ref.amplitude_to_db(torch.from_numpy(want_m)).numpy()
        big = want_m > 1e-6 * want_m.max()
        assert np.abs(got_db - want_db)[big].max() < DB_ABS, (shape, n_mels)
    # forms outside the kernel: DFT-matrix route
    before = launches(tac)
    got = host(tac.stft(dev(x), 400, hop_length=160, onesided=False))
    assert 'tac_stft_f32' not in launched_since(tac, before)
    ref = numpy_ref.stft(x, 400, 160, onesided=False)
    assert rel_err(got[..., 0] + 1j * got[..., 1], ref) < 5e-6
    got = host(tac.Spectrogram(400, hop_length=160, power=0.7).cuda()(dev(x)))
    assert rel_err(got, torch_ref.spectrogram(torch.from_numpy(x), 400, 160, power=0.7).numpy()) < 1e-5
