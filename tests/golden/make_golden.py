#!/usr/bin/env python
"""Generate tests/golden/*.npz from the UNMODIFIED reference (this container only).

    PYTHONPATH=/root/reference python tests/golden/make_golden.py [--skip-scan]

The reference calls ``torch.stft`` without ``return_complex`` (functional.py:99-107), which
torch>=2 rejects; the shim below (installed in THIS process only, reference untouched)
forwards to ``return_complex=True`` and returns ``view_as_real`` — the legacy layout the
reference expects.  Inputs come from ``oracle.signals`` (pure function of index+seed), so
only outputs are stored.  The reference's files never enter the repo or the GPU box.
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')

_orig_stft = torch.stft


def _stft_legacy(input, n_fft, hop_length=None, win_length=None, window=None, center=True,
                 pad_mode='reflect', normalized=False, onesided=None, return_complex=None):
    out = _orig_stft(input, n_fft, hop_length=hop_length, win_length=win_length, window=window,
                     center=center, pad_mode=pad_mode, normalized=normalized, onesided=onesided,
                     return_complex=True)
    return torch.view_as_real(out)


torch.stft = _stft_legacy
sys.path.insert(0, '/root/reference')
import torchaudio_contrib as ref  # noqa: E402  (the reference)

from oracle import signals  # noqa: E402

T = torch.from_numpy


def np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def g1():
    x = T(signals.audio_like((4, 1, 16000), seed=1))
    win = torch.hann_window(512)
    cplx = ref.stft(x, 512, hop_length=256, window=win)
    mag = ref.complex_norm(cplx, 1.0)
    seq = torch.nn.Sequential(*ref.Spectrogram(512, hop_length=256, window=win),
                              ref.AmplitudeToDb(ref=1.0, amin=1e-7))
    np.savez_compressed(os.path.join(GOLD, 'g1_cfg1.npz'),
                        stft=np32(cplx), mag=np32(mag), spec_db=np32(seq(x)))


G2_FRAMES = [0, 1, 2, 3, 155, 156, 309, 310, 311, 312]


def g2():
    x = T(signals.audio_like((2, 1, 160000), seed=2))
    mel_mod = ref.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512)
    full = torch.nn.Sequential(*mel_mod, ref.AmplitudeToDb())
    power = torch.nn.Sequential(*list(mel_mod)[:2])(x)
    np.savez_compressed(os.path.join(GOLD, 'g2_cfg2_slice.npz'),
                        mel=np32(mel_mod(x)), mel_db=np32(full(x)),
                        power_frames=np32(power[..., G2_FRAMES]),
                        frame_index=np.array(G2_FRAMES))


FB_CASES = {
    'slaney_1025_128_8000': (1025, 128, 0.0, 8000, False),
    'slaney_1025_128_22050': (1025, 128, 0.0, 22050, False),
    'htk_1025_128_8000': (1025, 128, 0.0, 8000, True),
    'slaney_257_128_1': (257, 128, 0.0, 1.0, False),
    'slaney_2049_128_24000': (2049, 128, 0.0, 24000, False),
    'htk_257_40_20_4000': (257, 40, 20.0, 4000.0, True),
}


def g3():
    out = {}
    for name, (f, m, lo, hi, htk) in FB_CASES.items():
        out[name] = np32(ref.create_mel_filter(f, m, lo, hi, htk))
    # the layer path (MelFilterbank defaults: max_freq = sample_rate // 2)
    out['layer_default_sr16000'] = np32(ref.MelFilterbank(sample_rate=16000).get_filterbank())
    np.savez_compressed(os.path.join(GOLD, 'g3_filterbanks.npz'), **out)


def g4():
    base = signals.audio_like((1, 2, 20000), seed=4)
    x = T(base)
    xs = T(np.ascontiguousarray(base[..., :6000]))
    custom_win = T(np.abs(signals.uniform((512,), seed=44)) + np.float32(0.25))
    out = {}
    out['n4096_h1024'] = np32(ref.stft(x, 4096, hop_length=1024))
    out['n512_h128_win400'] = np32(ref.stft(xs, 512, hop_length=128, win_length=400))
    out['n256_h64_normalized'] = np32(ref.stft(xs, 256, hop_length=64, normalized=True))
    out['n256_h100_twosided'] = np32(ref.stft(xs, 256, hop_length=100, onesided=False))
    out['n1024_h256_nocenter'] = np32(ref.stft(xs, 1024, hop_length=256, center=False))
    for mode in ('constant', 'replicate', 'circular'):
        out['n512_h256_' + mode] = np32(ref.stft(xs, 512, hop_length=256, pad_mode=mode))
    out['n512_h256_customwin'] = np32(ref.stft(xs, 512, hop_length=256, window=custom_win))
    out['n1024_hdefault'] = np32(ref.stft(xs, 1024))
    out['n400_h160'] = np32(ref.stft(xs, 400, hop_length=160))
    out['n2048_h512'] = np32(ref.stft(xs, 2048, hop_length=512))
    out['n128_h32'] = np32(ref.stft(xs[..., :2000], 128, hop_length=32))
    out['n64_h16'] = np32(ref.stft(xs[..., :1000], 64, hop_length=16))
    x4 = T(signals.audio_like((2, 2, 2, 3000), seed=5))
    out['lead3_n256_h64'] = np32(ref.stft(x4, 256, hop_length=64))
    # Spectrogram with non-default power, and cfg-4 style magnitude
    out['spec_p07_n512'] = np32(ref.complex_norm(ref.stft(xs, 512, hop_length=256), 0.7))
    out['spec_p1_n4096'] = np32(ref.complex_norm(ref.stft(x, 4096, hop_length=1024), 1.0))
    # 44.1 kHz style mel chain on a short clip, HTK, + dB with non-default ref/amin
    xm = T(signals.audio_like((3, 1, 30000), seed=6))
    mel = ref.Melspectrogram(num_mels=128, sample_rate=44100, fft_length=2048, hop_length=512)
    out['mel_sr44100'] = np32(mel(xm))
    melh = ref.Melspectrogram(num_mels=40, sample_rate=16000, min_freq=20.0, max_freq=7600.0, htk=True,
                              fft_length=512, hop_length=160, win_length=400)
    out['mel_htk40_n512'] = np32(melh(xm))
    out['mel_htk40_n512_db'] = np32(ref.AmplitudeToDb(ref=2.0, amin=1e-5)(melh(xm)))
    np.savez_compressed(os.path.join(GOLD, 'g4_variants.npz'), **out)


def scan_thresholds(n_quantize=256, log=None):
    """Exhaustive monotonicity scan of the reference encoder over every float32 in [-1, 1]."""
    torch.set_num_threads(os.cpu_count())
    one = 0x3F800000
    chunk = 1 << 26
    res = {}
    for sign, name in ((1.0, 'pos'), (-1.0, 'neg')):
        thresholds, prev_last, viol = [], None, 0
        for lo in range(0, one + 1, chunk):
            hi = min(lo + chunk, one + 1)
            bits = torch.arange(lo, hi, dtype=torch.int32)
            x = bits.view(torch.float32)
            if sign < 0:
                x = -x
            codes = ref.mu_law_encoding(x, n_quantize)
            d = codes[1:] - codes[:-1]
            if prev_last is not None:
                d = torch.cat([(codes[:1] - prev_last), d])
                base = lo
            else:
                base = lo + 1
            bad = (d * sign < 0).sum().item() + (d.abs() > 1).sum().item()
            viol += bad
            idx = torch.nonzero(d).flatten()
            for i in idx.tolist():
                b = base + i                          # first bit pattern carrying the new code
                thresholds.append(b)
            prev_last = codes[-1:].clone()
        res[name] = (np.array(thresholds, dtype=np.int64), viol)
        msg = '%s half: %d transitions, %d monotonicity violations' % (name, len(thresholds), viol)
        print(msg)
        if log is not None:
            log.append(msg)
    return res


def g5(skip_scan):
    out = {}
    log = []
    if not skip_scan:
        t0 = time.time()
        res = scan_thresholds(256, log)
        log.append('exhaustive scan of 2*(0x3f800000+1) float32 values: %.1f s' % (time.time() - t0))
        pos_bits, pv = res['pos']
        neg_bits, nv = res['neg']
        assert pv == 0 and nv == 0
        # magnitude bit patterns at which the code changes, for x>=0 and x<=0
        out['thr256_pos_bits'] = pos_bits.astype(np.int32)
        out['thr256_neg_bits'] = neg_bits.astype(np.int32)
        out['code_at_zero_256'] = np.array(ref.mu_law_encoding(torch.zeros(1), 256).item())
        out['code_at_negzero_256'] = np.array(ref.mu_law_encoding(-torch.zeros(1), 256).item())
    else:
        old = np.load(os.path.join(GOLD, 'g5_mulaw.npz'))
        for k in ('thr256_pos_bits', 'thr256_neg_bits', 'code_at_zero_256', 'code_at_negzero_256'):
            out[k] = old[k]
        log = [str(s) for s in old['scan_log']]
    out['scan_log'] = np.array(log)
    out['lut256'] = np32(ref.mu_law_decoding(torch.arange(256), 256))
    out['lut65536_sample'] = np32(ref.mu_law_decoding(torch.arange(0, 65536, 16), 65536))
    x1 = T(signals.uniform((1000000,), seed=7, scale=4.0))        # includes |x|>1 (reference test does)
    out['enc256_scale4'] = np32(ref.mu_law_encoding(x1, 256)).astype(np.int16)
    x2 = T(signals.uniform((1000000,), seed=8, scale=1.0))
    out['enc256_unit'] = np32(ref.mu_law_encoding(x2, 256)).astype(np.int16)
    out['enc65536_unit'] = np32(ref.mu_law_encoding(x2[:200000], 65536)).astype(np.int32)
    pos, neg = out['thr256_pos_bits'].astype(np.uint32), out['thr256_neg_bits'].astype(np.uint32)
    mags = np.concatenate([pos, pos - 1, neg, neg - 1, [0, 0x3f800000]]).astype(np.uint32)
    edges = np.concatenate([mags.view(np.float32), -(mags.view(np.float32))])
    out['enc256_edges'] = np32(ref.mu_law_encoding(T(edges), 256)).astype(np.int16)
    out['enc16_unit'] = np32(ref.mu_law_encoding(x2[:200000], 16)).astype(np.int16)
    x3 = T(signals.uniform((200000,), seed=11, scale=1000.0))     # far outside [-1, 1]: closed-form path
    out['enc1024_scale1000'] = np32(ref.mu_law_encoding(x3, 1024)).astype(np.int32)
    out['enc7_scale1000'] = np32(ref.mu_law_encoding(x3, 7)).astype(np.int16)
    out['special_inputs'] = np.array([0.0, -0.0, 1.0, -1.0, 1e-45, -1e-45, 1e-39, -1e-39, 1.0000001, -1.0000001,
                                      1e10, -1e10, 1e30, -1e30, np.inf, -np.inf, np.nan], np.float32)
    out['enc256_special'] = np32(ref.mu_law_encoding(T(out['special_inputs']), 256)).astype(np.int64)
    out['enc65536_special'] = np32(ref.mu_law_encoding(T(out['special_inputs']), 65536)).astype(np.int64)
    codes = T((signals.uniform((4096,), seed=9) * 127.5 + 127.5).astype(np.int64).clip(0, 255))
    out['dec256_codes'] = np32(ref.mu_law_decoding(codes, 256))
    out['db_known_amp'] = np32(ref.amplitude_to_db(torch.tensor([1e-6, 1e-4, 0.1, 1.0, 10.0, 1e6]).sqrt()))
    xa = T(signals.audio_like((4, 5000), seed=10))
    out['a2db_ref2'] = np32(ref.amplitude_to_db(xa, ref=2.0, amin=1e-5))
    out['db2a_ref2'] = np32(ref.db_to_amplitude(xa * 40, ref=2.0))
    np.savez_compressed(os.path.join(GOLD, 'g5_mulaw.npz'), **out)


def g6():
    """angle / magphase (functional.py:187-201) on a small complex tensor incl. the axes and the origin."""
    z = signals.audio_like((3, 65, 11, 2), seed=31)
    z[0, 0, :4] = np.array([[0.0, 0.0], [1.0, 0.0], [-1.0, 0.0], [0.0, -2.0]], np.float32)
    zt = T(z)
    out = {'angle': np32(ref.angle(zt))}
    for p in (1.0, 2.0, 0.5):
        m, ph = ref.magphase(zt, power=p)
        out['mag_p%g' % p] = np32(m)
        out['phase_p%g' % p] = np32(ph)
    np.savez_compressed(os.path.join(GOLD, 'g6_magphase.npz'), **out)


def g7():
    """phase_vocoder (functional.py:204-274) on a small complex spectrogram, three rates."""
    num_freqs, hop = 65, 32
    z = signals.audio_like((2, 1, num_freqs, 40, 2), seed=41)
    adv = torch.linspace(0, math.pi * hop, num_freqs)[..., None]
    out = {}
    for rate in (1.3, 0.7, 2.0):
        out['pv_rate%g' % rate] = np32(ref.phase_vocoder(T(z), rate, adv))
    np.savez_compressed(os.path.join(GOLD, 'g7_phase_vocoder.npz'), **out)


def g8():
    """hpss (beta_hpss.py:35-127) on a small magnitude spectrogram: soft and hard masks, two kernel sizes."""
    sys.path.insert(0, os.environ.get('TAC_REFERENCE', '/root/reference'))
    from torchaudio_contrib import beta_hpss
    mag = np.abs(signals.audio_like((2, 2, 45, 60), seed=51)) + 0.01 * np.abs(signals.uniform((2, 2, 45, 60), seed=52))
    out = {}
    for k, power, hard in ((31, 2.0, False), (7, 1.0, False), (31, 2.0, True), (9, 0.5, False)):
        res = beta_hpss.hpss(T(mag.astype(np.float32)), k, power, hard)
        tag = 'k%d_p%g_%s' % (k, power, 'hard' if hard else 'soft')
        for name, r in zip(('harm', 'perc', 'mask_harm', 'mask_perc'), res):
            out[tag + '_' + name] = r.numpy().astype(np.float32 if not hard or name in ('harm', 'perc') else np.uint8)
    np.savez_compressed(os.path.join(GOLD, 'g8_hpss.npz'), **out)


def g9():
    """mu-law codes -> MuLawDecoding -> Melspectrogram -> AmplitudeToDb (layers.py:443-467, 307-381): the chain whose
    decode step the streaming kernel folds into its frame load."""
    x = T(signals.audio_like((2, 1, 20000), seed=61))
    codes = ref.mu_law_encoding(x, 256)
    wave = ref.mu_law_decoding(codes, 256)
    out = {'codes': codes.numpy().astype(np.uint8)}
    for n_fft, hop, mels in ((2048, 512, 128), (512, 128, 40)):
        model = torch.nn.Sequential(*ref.Melspectrogram(num_mels=mels, sample_rate=16000, fft_length=n_fft, hop_length=hop),
                                    ref.AmplitudeToDb())
        out['mel_db_n%d' % n_fft] = np32(model(wave))
    np.savez_compressed(os.path.join(GOLD, 'g9_mulaw_mel.npz'), **out)


def g10():
    """Melspectrogram (-> AmplitudeToDb) at fft_length 4096 (layers.py:307-381; BASELINE configs[3]'s geometry with the
    mel tail of configs[2]: 44.1 kHz, 128 bands over 2049 bins, bands up to 129 bins wide)."""
    x = T(signals.audio_like((2, 2, 30000), seed=71))
    mel = ref.Melspectrogram(num_mels=128, sample_rate=44100, fft_length=4096, hop_length=1024)
    out = {'mel': np32(mel(x)), 'mel_db': np32(torch.nn.Sequential(*mel, ref.AmplitudeToDb())(x))}
    mel80 = ref.Melspectrogram(num_mels=80, sample_rate=48000, fft_length=4096, hop_length=1024, htk=True, min_freq=50.0)
    out['mel80_htk'] = np32(mel80(x))
    np.savez_compressed(os.path.join(GOLD, 'g10_mel4096.npz'), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--skip-scan', action='store_true', help='reuse thresholds from the existing g5 file')
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    jobs = {'g1': g1, 'g2': g2, 'g3': g3, 'g4': g4, 'g5': lambda: g5(a.skip_scan), 'g6': g6, 'g7': g7, 'g8': g8, 'g9': g9, 'g10': g10}
    for name, fn in jobs.items():
        if a.only and name not in a.only.split(','):
            continue
        t0 = time.time()
        fn()
        print('%s done in %.1f s' % (name, time.time() - t0))
    print('torch', torch.__version__)


if __name__ == '__main__':
    main()
