#!/usr/bin/env python
"""GPU bring-up helper: run every kernel family once against the oracle and PRINT the errors
(no asserts) so one gpurun call shows everything that is wrong."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchaudio_contrib_amd as tac  # noqa: E402
from oracle import signals, torch_ref  # noqa: E402


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def section(name, fn):
    try:
        print('%-34s %s' % (name, fn()), flush=True)
    except Exception:
        print('%-34s EXCEPTION' % name)
        traceback.print_exc()
    torch.cuda.synchronize()


def stft_case(n, hop, L, rows=3, **kw):
    def f():
        x = signals.audio_like((rows, 2, L), seed=n)
        got = tac.stft(torch.from_numpy(x).cuda(), n, hop_length=hop, **kw).cpu().numpy()
        want = torch_ref.stft(torch.from_numpy(x), n, hop, **{k: v for k, v in kw.items()}).numpy()
        e = rel(got, want)
        # locate the worst frame/bin for debugging
        d = np.abs(got - want).reshape(-1, *got.shape[-3:])
        r, fb, t, c = np.unravel_index(d.argmax(), d.shape)
        return 'rel %.3e shape %s worst(row %d bin %d frame %d c %d)' % (e, got.shape, r, fb, t, c)
    return f


def main():
    print(torch.cuda.get_device_name(0), torch.version.hip)
    for n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        section('stft n=%d' % n, stft_case(n, n // 4, max(3 * n, 5000)))
    section('stft n=512 hop=100 odd', stft_case(512, 101, 7001))
    section('stft n=2048 nocenter', stft_case(2048, 512, 20000, center=False))

    def spec():
        x = signals.audio_like((2, 1, 30000), seed=2)
        got = tac.Spectrogram(2048, 512, power=2.).cuda()(torch.from_numpy(x).cuda()).cpu().numpy()
        want = torch_ref.spectrogram(torch.from_numpy(x), 2048, 512, power=2.0).numpy()
        return 'rel %.3e' % rel(got, want)
    section('spectrogram power=2', spec)

    def mel(n, hop, mels, sr):
        def f():
            x = signals.audio_like((3, 1, 40000), seed=5)
            m = tac.Melspectrogram(num_mels=mels, sample_rate=sr, fft_length=n, hop_length=hop).cuda()
            got = m(torch.from_numpy(x).cuda()).cpu().numpy()
            want = torch_ref.melspectrogram(torch.from_numpy(x), num_mels=mels, sample_rate=sr, n_fft=n, hop=hop).numpy()
            d = np.abs(got - want)
            r, c, b, t = np.unravel_index(d.argmax(), d.shape)
            return 'rel %.3e worst(row %d band %d frame %d) got %.4g want %.4g' % (rel(got, want), r, b, t, got[r, c, b, t], want[r, c, b, t])
        return f
    section('melspec fused 2048/512/128', mel(2048, 512, 128, 16000))
    section('melspec fused 512/128/40', mel(512, 128, 40, 16000))
    section('melspec fused 1024/256/80', mel(1024, 256, 80, 22050))

    def meldb():
        x = signals.audio_like((3, 1, 40000), seed=5)
        m = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512),
                                tac.AmplitudeToDb()).cuda()
        got = m(torch.from_numpy(x).cuda()).cpu().numpy()
        want = torch_ref.melspectrogram_db(torch.from_numpy(x), n_fft=2048, hop=512, num_mels=128, sample_rate=16000).numpy()
        return 'max abs dB err %.3e' % np.abs(got - want).max()
    section('melspec+dB fused', meldb)

    def afb():
        spec = signals.uniform((2, 257, 391), seed=13)
        fb = signals.uniform((257, 120), seed=14)
        got = tac.apply_filterbank(torch.from_numpy(spec).cuda(), torch.from_numpy(fb).cuda()).cpu().numpy()
        want = np.einsum('rft,fm->rmt', spec.astype(np.float64), fb.astype(np.float64))
        return 'rel %.3e' % rel(got, want)
    section('apply_filterbank dense', afb)

    def ew():
        z = signals.uniform((1025, 400, 2), seed=12, scale=4.0)
        e1 = np.abs(tac.complex_norm(torch.from_numpy(z).cuda(), 0.7).cpu().numpy() - (z.astype(np.float64) ** 2).sum(-1) ** 0.35).max()
        x = signals.audio_like((4, 5000), seed=10)
        e2 = np.abs(tac.amplitude_to_db(torch.from_numpy(x).cuda()).cpu().numpy() - torch_ref.amplitude_to_db(torch.from_numpy(x)).numpy()).max()
        return 'complex_norm abs %.3e  a2db abs %.3e' % (e1, e2)
    section('elementwise', ew)

    def mulaw():
        x = signals.uniform((1000000,), seed=8)
        got = tac.mu_law_encoding(torch.from_numpy(x).cuda(), 256).cpu()
        want = torch_ref.mu_law_encoding(torch.from_numpy(x), 256)
        nbad = int((got != want).sum())
        c = torch.arange(256)
        dec = tac.mu_law_decoding(c.cuda(), 256).cpu()
        dbad = int((dec != torch_ref.mu_law_decoding(c, 256)).sum())
        got_nothr = tac.mu_law_encoding(torch.from_numpy(x).cuda(), 255).cpu()
        nb2 = int((got_nothr != torch_ref.mu_law_encoding(torch.from_numpy(x), 255)).sum())
        return 'encode mismatches %d / 1e6, decode LUT mismatches %d, formula-path(nq=255) mismatches %d' % (nbad, dbad, nb2)
    section('mu-law', mulaw)


if __name__ == '__main__':
    main()
