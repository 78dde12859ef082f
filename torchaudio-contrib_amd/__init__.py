"""torchaudio_contrib_amd — MI355X (gfx950) engine behind the torchaudio-contrib Melspectrogram path.

Import surface mirrors ``torchaudio_contrib/__init__.py:1-2`` of the reference (everything from
``functional`` and ``layers`` at top level).  The directory is named ``torchaudio-contrib_amd``;
the importable name is ``torchaudio_contrib_amd`` (see the loader shim at the repo root).
"""
from . import _native
from . import _ops
from ._ops import set_strict, CompositeRouteWarning
from ._hip import invalidate
from ._lazy import realize, set_lazy_fusion, lazy_fusion_enabled, DeferredSpectral, DeferredWave, planned, PlannedChain
from . import functional
from . import layers
from .functional import *      # noqa: F401,F403
from .functional import (stft, complex_norm, create_mel_filter, apply_filterbank, angle, magphase,
                         phase_vocoder, amplitude_to_db, db_to_amplitude, mu_law_encoding,
                         mu_law_decoding, hpss)
from .layers import (STFT, ComplexNorm, ApplyFilterbank, Filterbank, MelFilterbank, TimeStretch,
                     Spectrogram, Melspectrogram, AmplitudeToDb, DbToAmplitude, MuLawEncoding,
                     MuLawDecoding, HPSS)
from . import distributed

__version__ = '0.1.0'


def set_fft_pipe(mode=None):
    """Where the fft_length-2048 fused chain runs a frame's 1024-point transform: ``'valu'`` (radix 16 . 16 . 4 through the LDS,
    the default and the faster form on MI355X), ``'mfma'`` (two chained 32 x 32 complex DFT products on the matrix pipe with fp16
    hi / lo operand pairs: ``csrc/melspec_mfma.hpp``; same results to ~2e-7 of a frame's largest bin) or ``None`` (the library's
    default: environment ``TAC_FFT_PIPE``).  Process-wide; returns the previous setting.  C ABI: ``tac_set_fft_pipe``."""
    names = {None: -1, 'valu': 0, 'mfma': 1}
    if mode not in names:
        raise ValueError("set_fft_pipe: mode must be None, 'valu' or 'mfma', not %r" % (mode,))
    prev = _native.lib().tac_set_fft_pipe(names[mode])
    return {-1: None, 0: 'valu', 1: 'mfma'}[prev]


def build_native(verbose=False):
    """(Re)build libtac_amd.so for gfx950."""
    return _native.build(verbose=verbose)
