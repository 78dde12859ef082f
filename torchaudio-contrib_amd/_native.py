"""ctypes binding of libtac_amd.so — the C ABI declared in include/tac_amd.h.

PyTorch is plumbing here (device memory, the current HIP stream); every compute call goes to
a hand-written gfx950 kernel through plain pointers.  There is deliberately NO fallback: if
the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import subprocess
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TAC_AMD_LIB', os.path.join(_HERE, 'libtac_amd.so'))   # override: A/B builds
CSRC = os.path.join(_HERE, 'csrc')

TAC_OK = 0
TAC_E_INVALID = -1
TAC_E_UNSUPPORTED = -2
TAC_E_SHORT_INPUT = -3
TAC_E_LAUNCH = -4

PAD_MODES = {'constant': 0, 'reflect': 1, 'replicate': 2, 'circular': 3}
SAMPLES_F32, SAMPLES_I16, SAMPLES_MULAW_U8, SAMPLES_MULAW_I64 = 0, 1, 2, 3

EXPORTS = (
    'tac_strerror', 'tac_last_hip_error', 'tac_abi_version', 'tac_num_frames', 'tac_num_bins',
    'tac_stft_f32', 'tac_spectrogram_f32', 'tac_melspec_f32', 'tac_melspec_supported', 'tac_filterbank_plan',
    'tac_melbank_pack', 'tac_melbank_pack_host', 'tac_melspec_sparse_f32',
    'tac_apply_filterbank_f32', 'tac_apply_filterbank_sparse_f32', 'tac_apply_filterbank_sparse_db_f32', 'tac_complex_norm_f32', 'tac_magphase_f32', 'tac_phase_vocoder_f32', 'tac_phase_vocoder_f64', 'tac_phase_vocoder_backward_f32', 'tac_amplitude_to_db_f32',
    'tac_db_to_amplitude_f32', 'tac_mulaw_encode_f32_i64', 'tac_mulaw_decode_i64_f32',
    'tac_mulaw_decode_f32_f32', 'tac_mulaw_encode_f64_i64', 'tac_mulaw_decode_f64',
    'tac_stft_backward_f32', 'tac_stft_norm_backward_f32', 'tac_spectrogram_backward_f32', 'tac_spectrogram_backward_ola_workspace', 'tac_spectrogram_backward_ola_f32', 'tac_melspectrogram_backward_ola_f32', 'tac_melspectrogram_backward_f32', 'tac_filterbank_adjoint_pack', 'tac_apply_filterbank_adjoint_f32', 'tac_overlap_add_f32', 'tac_complex_norm_backward_f32', 'tac_amplitude_to_db_backward_f32', 'tac_magphase_backward_f32', 'tac_db_to_amplitude_backward_f32', 'tac_hpss_f32', 'tac_hpss_backward_f32', 'tac_melspec_sparse_coded_f32', 'tac_pcm16_to_f32',
    'tac_fold_twosided_f32', 'tac_window_grad_partials', 'tac_window_grad_f32', 'tac_sum_slabs_f32',
    'tac_stft_f64', 'tac_spectrogram_f64', 'tac_apply_filterbank_f64', 'tac_magphase_f64', 'tac_amplitude_to_db_f64',
    'tac_db_to_amplitude_f64',
    'tac_last_route', 'tac_debug_clock_probe', 'tac_melbank_plan_pieces_host', 'tac_set_fft_pipe',
)
ABI_VERSION = 5          # tac_abi_version() of the library this binding was written against (csrc/host_common.hip)


class StftDesc(ctypes.Structure):
    """mirror of ``tac_stft_desc`` (include/tac_amd.h)."""
    _fields_ = [('rows', ctypes.c_int64), ('length', ctypes.c_int64), ('row_stride', ctypes.c_int64),
                ('n_fft', ctypes.c_int32), ('hop', ctypes.c_int32), ('win_length', ctypes.c_int32),
                ('center', ctypes.c_int32), ('pad_mode', ctypes.c_int32), ('normalized', ctypes.c_int32),
                ('onesided', ctypes.c_int32), ('reserved', ctypes.c_int32)]


class NativeLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into libtac_amd.so (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC, '-j', str(min(8, os.cpu_count() or 1))]
    proc = subprocess.run(cmd + ['../libtac_amd.so'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
    if proc.returncode != 0:
        raise NativeLibraryError('building libtac_amd.so failed:\n' + proc.stdout[-4000:])
    # the compiled PyTorch binding of the hot call (host C++ against the torch headers): an accelerator of the call, ctypes
    # remains the general binding — a failure here is reported, not fatal
    proc = subprocess.run(cmd + ['ext'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout[-4000:])
    global _lib, _ext, _ext_tried
    _lib = None
    _ext, _ext_tried = None, False
    return LIB_PATH


_lib = None
_lock = threading.Lock()

_P = ctypes.c_void_p
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64
_F = ctypes.c_float
_D = ctypes.c_double
_DESC = ctypes.POINTER(StftDesc)


def lib():
    """Load (once) and return the ctypes handle; fail loudly when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                'libtac_amd.so not found at %s — build it with `make -C %s` (or '
                '`python -c "import __graft_entry__ as g; g.build()"`).  There is no CPU fallback.'
                % (LIB_PATH, CSRC))
        h = ctypes.CDLL(LIB_PATH)
        h.tac_abi_version.restype = ctypes.c_int
        found = h.tac_abi_version()
        if found != ABI_VERSION:
            raise NativeLibraryError(
                '%s reports ABI version %d, this binding needs %d — rebuild it with `make -C %s`'
                % (LIB_PATH, found, ABI_VERSION, CSRC))
        h.tac_strerror.restype = ctypes.c_char_p
        h.tac_strerror.argtypes = [ctypes.c_int]
        h.tac_last_hip_error.restype = ctypes.c_int
        h.tac_abi_version.restype = ctypes.c_int
        h.tac_num_frames.restype = _I64
        h.tac_num_frames.argtypes = [_I64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        h.tac_num_bins.restype = ctypes.c_int
        h.tac_num_bins.argtypes = [ctypes.c_int, ctypes.c_int]
        h.tac_stft_f32.argtypes = [_P, _P, _DESC, _P, _P]
        h.tac_spectrogram_f32.argtypes = [_P, _P, _DESC, _F, ctypes.c_int, _F, _F, _P, _P]
        h.tac_melspec_f32.argtypes = [_P, _P, _DESC, _F, _P, _P, _I32, ctypes.c_int, _F, _F, _P, _P]
        h.tac_melspec_supported.argtypes = [_DESC, _F, _P, _I32]
        h.tac_melbank_pack.argtypes = [_P, _I32, _I32, _I32, _P, _I32, _P, _I32, _P, _P]
        h.tac_melspec_sparse_f32.argtypes = [_P, _P, _DESC, _F, _P, _P, _P, _I32, ctypes.c_int, _F, _F, _P, _P]
        h.tac_filterbank_plan.argtypes = [_P, _I32, _I32, _P, _P, _P]
        h.tac_apply_filterbank_f32.argtypes = [_P, _I64, _I32, _I64, _I64, _I64, _I64, _P, _P, _I32, _P, _P]
        h.tac_apply_filterbank_sparse_f32.argtypes = [_P, _I64, _I32, _I64, _I64, _I64, _P, _P, _P, _I32, _P, _P]
        h.tac_apply_filterbank_sparse_db_f32.argtypes = [_P, _I64, _I32, _I64, _I64, _I64, _P, _P, _P, _I32, ctypes.c_int, _F, _F, _P, _P]
        h.tac_complex_norm_f32.argtypes = [_P, _I64, _F, _P, _P]
        h.tac_magphase_f32.argtypes = [_P, _I64, _F, _P, _P, _P]
        h.tac_phase_vocoder_f32.argtypes = [_P, _I64, _I32, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _I64, _P, _P]
        h.tac_phase_vocoder_f64.argtypes = h.tac_phase_vocoder_f32.argtypes
        h.tac_phase_vocoder_backward_f32.argtypes = [_P, _I64, _I32, _I64, _I64, _I64, _I64, _P, _P, _P, _I64, _P, _P, _P]
        h.tac_amplitude_to_db_f32.argtypes = [_P, _I64, _F, _F, _P, _P]
        h.tac_stft_f64.argtypes = [_P, _P, _DESC, _P, _P]
        h.tac_spectrogram_f64.argtypes = [_P, _P, _DESC, _D, _I32, _D, _D, _P, _P]
        h.tac_apply_filterbank_f64.argtypes = [_P, _I64, _I32, _I64, _I64, _I64, _I64, _P, _I32, _I32, _D, _D, _P, _P]
        h.tac_magphase_f64.argtypes = [_P, _I64, _D, _P, _P, _P]
        h.tac_amplitude_to_db_f64.argtypes = [_P, _I64, _D, _D, _P, _P]
        h.tac_db_to_amplitude_f64.argtypes = [_P, _I64, _D, _P, _P]
        h.tac_db_to_amplitude_f32.argtypes = [_P, _I64, _F, _P, _P]
        h.tac_mulaw_encode_f32_i64.argtypes = [_P, _I64, _I32, _P, _I32, _I32, _I32, _P, _P]
        h.tac_mulaw_decode_i64_f32.argtypes = [_P, _I64, _I32, _P, _P, _P]
        h.tac_mulaw_decode_f32_f32.argtypes = [_P, _I64, _I32, _P, _P, _P]
        h.tac_stft_backward_f32.argtypes = [_P, _P, _DESC, _P, _P]
        h.tac_stft_norm_backward_f32.argtypes = [_P, _P, _F, _P, _DESC, _P, _P]
        h.tac_spectrogram_backward_f32.argtypes = [_P, _P, _DESC, _P, _F, _P, _P]
        h.tac_spectrogram_backward_ola_workspace.argtypes = [_DESC]
        h.tac_spectrogram_backward_ola_workspace.restype = _I64
        h.tac_spectrogram_backward_ola_f32.argtypes = [_P, _P, _DESC, _P, _F, _P, _I64, _P, _I64, _P]
        h.tac_melspectrogram_backward_ola_f32.argtypes = [_P, _P, _DESC, _P, _I32, _P, _I32, _F, _P, _I64, _P, _I64, _P]
        h.tac_melspectrogram_backward_f32.argtypes = [_P, _P, _DESC, _P, _I32, _P, _I32, _F, _P, _P]
        h.tac_filterbank_adjoint_pack.argtypes = [_P, _I32, _I32, _P, _P, _P]
        h.tac_apply_filterbank_adjoint_f32.argtypes = [_P, _I64, _I32, _P, _I32, _P, _P]
        h.tac_overlap_add_f32.argtypes = [_P, _DESC, _P, _I64, _P]
        h.tac_complex_norm_backward_f32.argtypes = [_P, _P, _I64, _F, _P, _P]
        h.tac_amplitude_to_db_backward_f32.argtypes = [_P, _P, _I64, _F, _P, _P]
        h.tac_magphase_backward_f32.argtypes = [_P, _P, _P, _I64, _F, _P, _P]
        h.tac_db_to_amplitude_backward_f32.argtypes = [_P, _P, _I64, _F, _P, _P]
        h.tac_melspec_sparse_coded_f32.argtypes = [_P, _I32, _P, _P, _DESC, _F, _P, _P, _P, _I32, ctypes.c_int, _F, _F, _P, _P]
        h.tac_pcm16_to_f32.argtypes = [_P, _I64, _P, _P]
        h.tac_fold_twosided_f32.argtypes = [_P, _I64, _I32, _I32, _P, _P]
        h.tac_window_grad_partials.argtypes = [_DESC]
        h.tac_window_grad_partials.restype = _I64
        h.tac_window_grad_f32.argtypes = [_P, _P, _DESC, _P, _I64, _P]
        h.tac_sum_slabs_f32.argtypes = [_P, _I64, _I64, _P, _P]
        h.tac_hpss_f32.argtypes = [_P, _I64, _I32, _I32, _I64, _I64, _I64, _I32, _I32, _F, ctypes.c_int, _P, _P, _P, _P, _P]
        h.tac_hpss_backward_f32.argtypes = [_P, _I64, _I32, _I32, _I64, _I64, _I64, _I32, _I32, _F, ctypes.c_int, _P, _P, _P, _P, _P, _P]
        h.tac_melbank_plan_pieces_host.argtypes = [_P, _I32, _I32, _P, _P, _P, _P, _P, _I32]
        h.tac_melbank_plan_pieces_host.restype = ctypes.c_int
        h.tac_last_route.restype = ctypes.c_char_p
        h.tac_last_route.argtypes = []
        h.tac_debug_clock_probe.restype = ctypes.c_int
        h.tac_debug_clock_probe.argtypes = [_P, _I32]
        h.tac_set_fft_pipe.restype = ctypes.c_int
        h.tac_set_fft_pipe.argtypes = [ctypes.c_int]
        h.tac_mulaw_encode_f64_i64.argtypes = [_P, _I64, _I32, _P, _P]
        h.tac_mulaw_decode_f64.argtypes = [_P, _I32, _I64, _I32, _P, _P]
        h.tac_mulaw_decode_f64.restype = ctypes.c_int
        for name in EXPORTS:
            fn = getattr(h, name)
            if name.endswith(('_f32', '_f64', '_i64', '_plan', '_supported', '_pack')):   # every launcher returns a TAC_* code
                fn.restype = ctypes.c_int
        _lib = h
    return _lib


EXT_PATH = os.path.join(_HERE, '_tac_ext.so')
_ext = None
_ext_tried = False


def ext():
    """The compiled PyTorch binding of the hot call (csrc/binding/tac_ext.cpp -> _tac_ext.so), or None when it is not built /
    disabled with TAC_AMD_EXT=0 / built against a different torch — ctypes then carries every call (the general binding)."""
    global _ext, _ext_tried
    if _ext_tried:
        return _ext
    with _lock:
        if _ext_tried:
            return _ext
        mod = None
        if os.environ.get('TAC_AMD_EXT', '1') != '0' and os.path.exists(EXT_PATH):
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location(__package__ + '._tac_ext', EXT_PATH)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if getattr(mod, 'ABI', None) != 1:
                    mod = None
            except Exception as exc:            # noqa: BLE001 — the extension is an accelerator of the call, not the product
                import warnings
                warnings.warn('torchaudio_contrib_amd: compiled binding %s not usable (%s: %s); using ctypes' %
                              (EXT_PATH, type(exc).__name__, exc))
                mod = None
        _ext = mod
        _ext_tried = True
    return _ext


def binding():
    """'compiled' when the fused chain launches through _tac_ext.so, else 'ctypes'."""
    return 'compiled' if ext() is not None else 'ctypes'


def check(rc, what):
    """Turn a TAC_E_* code into the Python exception the reference's torch ops would raise."""
    if rc == TAC_OK:
        return
    h = lib()
    msg = '%s: %s' % (what, h.tac_strerror(rc).decode())
    if rc == TAC_E_LAUNCH:
        msg += ' (hipError %d)' % h.tac_last_hip_error()
    if rc == TAC_E_UNSUPPORTED:
        raise NotImplementedError(msg)          # a RuntimeError subclass
    raise RuntimeError(msg)


def raw_stream(device):
    """hipStream_t of torch's current stream on `device` as an int (no Stream object is built)."""
    return torch._C._cuda_getCurrentRawStream(device.index)


def stream_ptr(device):
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(device.index))


class _NoGuard(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """Context manager making `device` the current HIP device for a launch — free when it already is."""
    return _NO_GUARD if torch._C._cuda_getDevice() == device.index else torch.cuda.device(device)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())
