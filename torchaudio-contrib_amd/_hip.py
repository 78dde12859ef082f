"""Launchers of the gfx950 kernels: the bodies of the ``tac_amd::*`` ops for float32 tensors on a HIP device.

Every function takes fully resolved, already validated arguments (``functional.py`` fills in defaults;
``_ops.py`` routes by device / dtype) and goes through the C ABI of ``include/tac_amd.h`` via ctypes — raw
device pointers, sizes and the caller's current HIP stream.  PyTorch only provides the output allocation and
the stream.  There is no fallback in here: a missing ``libtac_amd.so`` or a failing launch raises.
"""
import ctypes
import itertools
import math
import os
import threading
import weakref

import torch

from . import _native

_lock = threading.Lock()

#: number of kernels launched through the C ABI since import, per entry point (tests assert on it so that a
#: silently taken non-HIP route cannot pass as the HIP path)
launches = {}


def _count(name):
    launches[name] = launches.get(name, 0) + 1


# ----------------------------------------------------------------------------- constant-table caches
# Everything derived from a window / filterbank (packed weights, tile plans, transposes, adjoint tables, DFT matrices)
# is cached ON the tensor object and stamped with what identifies its contents cheaply: the version counter (every
# in-place torch op bumps it), the data pointer (``set_`` / ``.data = other``) and the invalidation epoch below.
# Writes that PyTorch itself does not record — ``t.data.mul_(2)`` (``.data`` has its own version counter), writes
# through an alias created before the cache, ``from_dlpack`` / raw-pointer writers — cannot be seen without reading the
# tensor back on every call; after such a write call ``invalidate(t)`` (or ``invalidate()`` for everything).
_epoch = 0
_CACHE_ATTRS = ('_tac_pack', '_tac_plan', '_tac_T', '_tac_adj', '_tac_dft', '_tac_dftT')


_unstamped = itertools.count()


def _stamp(t):
    """What identifies a tensor's contents cheaply.  Tensors created under ``torch.inference_mode`` carry no version counter
    (reading it raises): they get a stamp that never repeats, i.e. their derived tables are rebuilt on every call rather
    than risk serving a stale one after an in-place write nobody recorded."""
    if t.is_inference():
        return ('unversioned', next(_unstamped))
    return (t._version, t.data_ptr(), _epoch)


def invalidate(tensor=None):
    """Forget the tables derived from ``tensor`` (a window or filterbank), or — without an argument — from every
    tensor: they are rebuilt from the current contents on the next call.  Needed only after a write PyTorch's version
    counter does not see (``t.data.mul_(2)``, raw-pointer writes); ordinary in-place ops are detected by themselves.
    Pass the tensor object the layers were given (``invalidate(module.filterbank)``): only its tables are dropped.  Passing an
    alias that carries no tables itself (``module.filterbank.data``, a view) is allowed and invalidates everything."""
    global _epoch
    from . import _lazy
    if tensor is None:
        with _lock:
            _epoch += 1
            _geometry_routes_clear()
            _lazy._plans.clear()
            ext = _native.ext()
            if ext is not None:
                ext.set_epoch(_epoch)           # (plans a caller still holds stop matching)
        return
    carried = False
    for name in _CACHE_ATTRS:
        if hasattr(tensor, name):
            carried = True
            try:
                delattr(tensor, name)
            except Exception:
                pass
    with _lock:
        carried = carried or any(k[0] == id(tensor) for g in _geometry_cache.values() for k in g.routes) \
            or any(id(tensor) in k[:2] for k in _lazy._plans)
    if not carried:
        # nothing is cached ON this object: it is an alias of the tensor the tables hang on (`module.filterbank.data`, a view,
        # a second wrapper of the same storage) — which object that is cannot be told from here, so everything goes
        invalidate()
        return
    # only this tensor's entries go: the tables of every other window / filterbank stay valid (a global epoch bump would
    # have each of them rebuilt, with a host synchronisation, on its next use)
    with _lock:
        for g in _geometry_cache.values():
            for key in [k for k in g.routes if k[0] == id(tensor)]:
                del g.routes[key]
        for key in [k for k in _lazy._plans if id(tensor) in k[:2]]:         # bound-argument launchers built from it
            del _lazy._plans[key]


def _geometry_routes_clear():
    for g in _geometry_cache.values():
        g.routes.clear()


# ----------------------------------------------------------------------------- STFT geometry
class StftGeometry(object):
    """Validated geometry of one stft call on one input layout (the checks ``torch.stft`` performs, reference
    functional.py:99-107), cached per (shape, strides, parameters) so that a repeated call costs a dict lookup."""
    __slots__ = ('lead', 'length', 'rows', 'row_stride', 'flatten', 'n_fft', 'hop', 'win_length', 'center',
                 'pad_mode', 'normalized', 'onesided', 'n_frames', 'n_bins', 'fft_kernel', 'pow2_kernel', 'mixed_radix', 'desc',
                 'stft_shape', 'spec_shape', 'routes')


_geometry_cache = {}


def stft_frames(length, n_fft, hop, center):
    pad = n_fft // 2 if center else 0
    return 1 + (length + 2 * pad - n_fft) // hop


def check_stft_args(shape, n_fft, hop, win_length, center, pad_mode):
    """The argument checks of ``torch.stft`` + ``F.pad`` that do not depend on the backend (same exception types:
    the reference's tests expect ``RuntimeError`` for an input too short to reflect-pad,
    reference tests/test_functional.py:31)."""
    if len(shape) < 1 or any(int(s) == 0 for s in shape):
        raise RuntimeError('stft: expected a non-empty tensor of shape (*, channel, time)')
    length = int(shape[-1])
    if n_fft <= 0 or hop <= 0:
        raise RuntimeError('stft: expected 0 < n_fft and 0 < hop_length, got n_fft=%d hop_length=%d' % (n_fft, hop))
    if not 0 < win_length <= n_fft:
        raise RuntimeError('stft: expected 0 < win_length <= n_fft, got win_length=%d' % win_length)
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
        raise RuntimeError('stft: expected n_fft <= padded signal length %d, got n_fft=%d' % (length + 2 * pad, n_fft))


def fft_kernel_size(n_fft):
    """power-of-two sizes in [32, 4096] take the wave-level FFT kernels (``big_fft_size``: 8192 ... 32768 the four-step
    kernel; ``smooth_fft_size``: even lengths with a 7-smooth half the generic Stockham kernel); every other size up to 8192
    is evaluated as a windowed-DFT matrix product on the fp32 matrix cores (``_stft_dft``) — except ``mixed_radix_size``."""
    return (n_fft & (n_fft - 1)) == 0 and 32 <= n_fft <= 4096


def mixed_radix_size(n_fft):
    """fft_length 400 (25 ms at 16 kHz) has its own kernel (csrc/stft_n400.hip: 200 = 8 x 25) for the one-sided
    complex / |X| / |X|^2 (+dB) rows; its other forms take the DFT-matrix route."""
    return n_fft == 400


def big_fft_size(n_fft):
    """fft_length 8192 / 16384 / 32768: one frame per workgroup as a four-step transform over the wave-level 1024-point FFT
    (csrc/stft_big.hip, round 5) — forward stft / spectrogram rows of every form; gradients at 8192 take the generic
    Stockham adjoint of csrc/stft_smooth.hip, above that they are a composite route."""
    return n_fft in (8192, 16384, 32768)


def smooth_fft_size(n_fft):
    """even fft_length <= 8192 (not a power of two, not 400) whose half is 7-smooth — 480, 960, 1200, 1920, 882 ...: generic
    Stockham passes of radix 4 / 2 / 3 / 5 / 7 (csrc/stft_smooth.hip, round 5) for the forward stft / spectrogram rows AND their
    gradients (``stft_smooth_backward_kernel`` behind ``tac_stft_backward_f32``: the same passes on conjugated data; 8192 as well).  Mirrors ``stft_smooth_covers`` of the library."""
    if n_fft < 8 or n_fft % 2 or n_fft > 8192 or n_fft & (n_fft - 1) == 0 or n_fft == 400:
        return False
    m = n_fft // 2
    for r in (2, 3, 5, 7):
        while m % r == 0:
            m //= r
    return m == 1


def hip_covers_n_fft(n_fft):
    return fft_kernel_size(n_fft) or n_fft <= 8192 or big_fft_size(n_fft)


def hip_covers_backward(n_fft):
    return fft_kernel_size(n_fft) or n_fft <= 8192


def geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
    key = (tuple(wave.shape), tuple(wave.stride()), n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    g = _geometry_cache.get(key)
    if g is not None:
        return g
    check_stft_args(wave.shape, n_fft, hop, win_length, center, pad_mode)
    g = StftGeometry()
    length = int(wave.shape[-1])
    g.lead = tuple(int(s) for s in wave.shape[:-1])
    g.length = length
    g.rows = 1
    for s in g.lead:
        g.rows *= s
    # can the rows be addressed as base + r * row_stride without a copy?
    flat = wave.reshape(-1, length) if wave.dim() != 2 else wave
    viewable = flat.data_ptr() == wave.data_ptr() and flat.stride(1) == 1 and \
        (flat.shape[0] == 1 or flat.stride(0) >= length)
    g.flatten = not viewable
    g.row_stride = length if (g.flatten or flat.shape[0] == 1) else int(flat.stride(0))
    g.n_fft, g.hop, g.win_length = n_fft, hop, win_length
    g.center, g.pad_mode, g.normalized, g.onesided = bool(center), pad_mode, bool(normalized), bool(onesided)
    g.n_frames = stft_frames(length, n_fft, hop, center)
    g.n_bins = n_fft // 2 + 1 if onesided else n_fft
    g.fft_kernel = fft_kernel_size(n_fft) or big_fft_size(n_fft) or smooth_fft_size(n_fft)   # tac_stft_f32 / tac_spectrogram_f32 take it
    g.pow2_kernel = fft_kernel_size(n_fft)                  # ... and the fused chains / gradient kernels of the power-of-two sizes
    if not g.fft_kernel and n_fft > 8192:
        raise NotImplementedError('stft: fft_length %d is outside the HIP path (power of two in [32, 32768], or any '
                                  'length <= 8192 through the DFT-matrix kernel)' % n_fft)
    g.mixed_radix = mixed_radix_size(n_fft)
    g.desc = None if not (g.fft_kernel or g.mixed_radix) else _native.StftDesc(
        rows=g.rows, length=length, row_stride=g.row_stride, n_fft=n_fft, hop=hop, win_length=win_length,
        center=1 if center else 0, pad_mode=_native.PAD_MODES[pad_mode], normalized=1 if normalized else 0,
        onesided=1 if onesided else 0, reserved=0)
    g.stft_shape = g.lead + (g.n_frames, g.n_bins, 2)
    g.spec_shape = g.lead + (g.n_frames, g.n_bins)
    g.routes = {}
    with _lock:
        if len(_geometry_cache) > 512:
            _geometry_cache.clear()
        _geometry_cache[key] = g
    return g


def _rows_of(wave, g):
    """The tensor whose data pointer + g.row_stride addresses the rows (a copy only for layouts that cannot be)."""
    return wave.reshape(-1, g.length).contiguous() if g.flatten else wave


def _dft_matrix(window, n_fft, win_length, onesided, normalized):
    """(N, 2F) float32 device matrix [w[n] cos(2 pi k n / N), -w[n] sin(2 pi k n / N)] for the DFT-matrix path,
    built on the device in float64 (angles reduced exactly through (n*k) mod N in int64) and rounded once — a
    constant table like the FFT twiddles, cached on the window tensor per (version, geometry)."""
    cache = getattr(window, '_tac_dft', None)
    key = (_stamp(window), n_fft, win_length, bool(onesided), bool(normalized))
    if cache is not None and cache[0] == key:
        return cache[1]
    dev = window.device
    w = torch.zeros(n_fft, dtype=torch.float64, device=dev)
    off = (n_fft - win_length) // 2
    w[off:off + win_length] = window.detach().double()
    if normalized:
        w = w / math.sqrt(n_fft)
    n_bins = n_fft // 2 + 1 if onesided else n_fft
    n = torch.arange(n_fft, dtype=torch.int64, device=dev)[:, None]
    k = torch.arange(n_bins, dtype=torch.int64, device=dev)[None, :]
    ang = ((n * k) % n_fft).double() * (2.0 * math.pi / n_fft)
    mat = torch.stack([torch.cos(ang) * w[:, None], -torch.sin(ang) * w[:, None]], dim=-1)
    mat = mat.to(torch.float32).reshape(n_fft, 2 * n_bins).contiguous()
    try:
        window._tac_dft = (key, mat)
    except Exception:
        pass
    return mat


def _stft_dft(wave, window, g):
    """Any fft_length (non power of two, or 4096 < N <= 8192): the framed signal is never materialised — the
    filterbank GEMM kernel reads frame t, sample n at ``padded[row, t*hop + n]`` (stride_f = 1, stride_t = hop)
    and multiplies by the (N, 2F) windowed-DFT matrix on the fp32 matrix cores.  The padded copy is plain data
    movement done by torch; everything arithmetic is the HIP kernel."""
    x = wave.reshape(-1, g.length)
    if g.center:
        pad = g.n_fft // 2
        x = torch.nn.functional.pad(x.unsqueeze(1), (pad, pad), mode=g.pad_mode).squeeze(1)
    x = x.contiguous()
    mat = _dft_matrix(window, g.n_fft, g.win_length, g.onesided, g.normalized)
    out = torch.empty(g.stft_shape, dtype=torch.float32, device=x.device)
    with _native.on_device(x.device):
        rc = _native.lib().tac_apply_filterbank_f32(
            _native.ptr(x), x.shape[0], g.n_fft, g.n_frames, x.stride(0), 1, g.hop,
            _native.ptr(mat), None, 2 * g.n_bins, _native.ptr(out), _native.stream_ptr(x.device))
    _native.check(rc, 'tac_apply_filterbank_f32 (DFT matrix)')
    _count('tac_apply_filterbank_f32')
    return out.transpose(-3, -2)


def stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
    g = geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    if not (g.fft_kernel or (g.mixed_radix and g.onesided)):
        return _stft_dft(wave, window, g)
    src = _rows_of(wave, g)
    out = torch.empty(g.stft_shape, dtype=torch.float32, device=wave.device)
    with _native.on_device(wave.device):
        rc = _native.lib().tac_stft_f32(_native.ptr(src), _native.ptr(window), g.desc, _native.ptr(out),
                                        _native.stream_ptr(wave.device))
    _native.check(rc, 'tac_stft_f32')
    _count('tac_stft_f32')
    return out.transpose(-3, -2)


def spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref, amin):
    g = geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    if not (g.fft_kernel or (g.mixed_radix and g.onesided and power in (1.0, 2.0))):
        mag = complex_norm(_stft_dft(wave, window, g), power)
        return amplitude_to_db(mag, ref, amin) if db else mag
    src = _rows_of(wave, g)
    out = torch.empty(g.spec_shape, dtype=torch.float32, device=wave.device)
    with _native.on_device(wave.device):
        rc = _native.lib().tac_spectrogram_f32(
            _native.ptr(src), _native.ptr(window), g.desc, float(power), 1 if db else 0, float(ref), float(amin),
            _native.ptr(out), _native.stream_ptr(wave.device))
    _native.check(rc, 'tac_spectrogram_f32')
    _count('tac_spectrogram_f32')
    return out.transpose(-2, -1)


#: test hook: gradient outputs start as NaN, so that a position no kernel writes cannot pass a comparison by luck
POISON_OUTPUTS = False

# A/B knob for the two fused Melspectrogram kernels: 'auto' (band-sparse when the bank allows it, else MFMA),
# 'sparse', 'mfma'
MEL_PATH = os.environ.get('TAC_MEL_PATH', 'auto')


def _melbank_pack(fb, n_fft):
    """(wpack, desc, info) device/host buffers of the band-sparse contraction for this filterbank and fft size, or
    None when the bank is not band-sparse enough (then the MFMA kernels are used).  Built once per filterbank
    version (one host sync) and cached on the tensor object."""
    cache = getattr(fb, '_tac_pack', None)
    if cache is None or cache[0] != _stamp(fb):
        cache = (_stamp(fb), {})
        try:
            fb._tac_pack = cache
        except Exception:
            pass
    if n_fft in cache[1]:
        return cache[1][n_fft]
    n_freqs, n_mels = fb.shape
    wpack = torch.empty(24576, dtype=torch.float32, device=fb.device)
    desc = torch.empty(8192, dtype=torch.int32, device=fb.device)
    info = (ctypes.c_int32 * 8)()
    with _native.on_device(fb.device):
        rc = _native.lib().tac_melbank_pack(_native.ptr(fb), n_freqs, n_mels, n_fft, _native.ptr(wpack), 24576,
                                            _native.ptr(desc), 8192, ctypes.cast(info, ctypes.c_void_p),
                                            _native.stream_ptr(fb.device))
    if rc == _native.TAC_E_UNSUPPORTED:
        result = None
    else:
        _native.check(rc, 'tac_melbank_pack')
        result = (wpack, desc, info)
    cache[1][n_fft] = result
    return result


def _filterbank_plan(fb):
    """(device int32 plan, host ctypes copy): non-zero bin range per 16-band tile, computed by a device kernel.
    The plan rides on the filterbank tensor object itself (a module's constant buffer is scanned once — the only
    host sync on the path — and rescanned when modified in place); keying a cache on ``data_ptr`` would go stale
    when the allocator reuses an address."""
    hit = getattr(fb, '_tac_plan', None)
    if hit is not None and hit[0] == _stamp(fb) and hit[1].device == fb.device:
        return hit[1], hit[2]
    n_freqs, n_mels = fb.shape
    n_ints = 2 * ((n_mels + 15) // 16)
    plan = torch.empty(n_ints, dtype=torch.int32, device=fb.device)
    host = (ctypes.c_int32 * n_ints)()
    with _native.on_device(fb.device):
        rc = _native.lib().tac_filterbank_plan(_native.ptr(fb), n_freqs, n_mels, _native.ptr(plan),
                                               ctypes.cast(host, ctypes.c_void_p), _native.stream_ptr(fb.device))
    _native.check(rc, 'tac_filterbank_plan')
    try:
        fb._tac_plan = (_stamp(fb), plan, host)
    except Exception:       # exotic tensor subclasses without attribute storage: just recompute next time
        pass
    return plan, host


def _fused_mel_route(g, fb, power):
    """'sparse' / 'mfma' when one fused kernel covers this geometry + filterbank, else None (then the caller chains
    the spectrogram, filterbank and dB kernels)."""
    if not ((g.pow2_kernel or g.mixed_radix) and g.onesided and g.n_fft <= 4096 and fb.dim() == 2 and
            fb.shape[0] == g.n_bins and 0 < fb.shape[1] <= 512 and fb.is_contiguous()):
        return None
    if g.mixed_radix:       # fft_length 400: only the band-sparse form has a fused kernel
        ok = power in (1.0, 2.0) and MEL_PATH != 'mfma' and _melbank_pack(fb, g.n_fft) is not None
        return 'sparse' if ok else None
    if power in (1.0, 2.0) and MEL_PATH != 'mfma' and _melbank_pack(fb, g.n_fft) is not None:
        return 'sparse'
    if MEL_PATH == 'sparse' or g.n_fft > 2048:      # (fft_length 4096: only the band-sparse form has a fused kernel)
        return None
    _, host = _filterbank_plan(fb)
    rc = _native.lib().tac_melspec_supported(g.desc, float(power), ctypes.cast(host, ctypes.c_void_p), fb.shape[1])
    return 'mfma' if rc == _native.TAC_OK else None


class MelPlan(object):
    """Bound-argument launcher of the fused band-sparse kernel for ONE (waveform layout, window, filterbank, parameters):
    what ``melspectrogram`` below works out on every call — geometry, route, packed bank, ctypes conversions — done once.
    ``launch(wave)`` then costs an allocation and one foreign call.  Valid while the window and the filterbank keep their
    stamps (checked by the caller, ``_lazy.DeferredSpectral.realize``) and the device is current."""
    __slots__ = ('fn', 'win_ptr', 'desc', 'power', 'wpack', 'dsc', 'info', 'wpack_ptr', 'dsc_ptr', 'info_ptr', 'n_mels',
                 'db', 'ref', 'amin', 'shape', 'device', 'dev_index', 'window', 'fb', 'win_stamp', 'fb_stamp', 'layout', 'cplan',
                 'last_rc')

    def run(self, wave):
        """``matches`` + ``launch`` in one: the result, or None when the plan does not apply (any more).  With the compiled
        binding (csrc/binding/tac_ext.cpp) the checks, the allocation, the stream lookup and the foreign call are one C++ call."""
        c = self.cplan
        if c is not None:
            v = c.launch(wave)
            if v is not None:
                launches['tac_melspec_sparse_f32'] = launches.get('tac_melspec_sparse_f32', 0) + 1
            else:
                self.last_rc = c.last_rc
            return v
        return self.launch(wave) if self.matches(wave) else None

    def launch(self, wave):
        out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        rc = self.fn(wave.data_ptr(), self.win_ptr, self.desc, self.power, self.wpack_ptr, self.dsc_ptr, self.info_ptr,
                     self.n_mels, self.db, self.ref, self.amin, out.data_ptr(),
                     torch._C._cuda_getCurrentRawStream(self.dev_index))
        if rc != _native.TAC_OK:
            self.last_rc = rc
            return None                     # (the general path reports it)
        launches['tac_melspec_sparse_f32'] = launches.get('tac_melspec_sparse_f32', 0) + 1
        return out.transpose(-2, -1)

    def matches(self, wave):
        """same layout, same device current, tables still those of the tensors' current contents"""
        return (wave.shape, wave.stride(), wave.dtype) == self.layout and torch._C._cuda_getDevice() == self.dev_index \
            and _stamp(self.window) == self.win_stamp and _stamp(self.fb) == self.fb_stamp


def mel_plan(wave, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref, amin):
    """A ``MelPlan`` when this call is one launch of the fused band-sparse kernel on plain float32 tensors whose rows need no
    copy, else None (the general ``melspectrogram`` handles everything)."""
    if not (type(wave) is torch.Tensor and wave.dtype == torch.float32 and window.dtype == torch.float32
            and window.is_contiguous() and fb.dtype == torch.float32 and fb.dim() == 2 and wave.is_cuda
            and window.device == wave.device and fb.device == wave.device and not window.is_inference()
            and not fb.is_inference()):
        return None
    g = geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    if g.flatten or g.desc is None or fb.shape[0] != g.n_bins or _fused_mel_route(g, fb, power) != 'sparse':
        return None
    wpack, dsc, info = _melbank_pack(fb, g.n_fft)
    p = MelPlan()
    p.fn = _native.lib().tac_melspec_sparse_f32
    p.window, p.fb, p.win_stamp, p.fb_stamp = window, fb, _stamp(window), _stamp(fb)
    p.win_ptr, p.desc, p.power = window.data_ptr(), g.desc, float(power)
    p.wpack, p.dsc, p.info = wpack, dsc, info                          # (kept alive with the plan)
    p.wpack_ptr, p.dsc_ptr, p.info_ptr = wpack.data_ptr(), dsc.data_ptr(), ctypes.cast(info, ctypes.c_void_p)
    p.n_mels, p.db, p.ref, p.amin = int(fb.shape[1]), 1 if db else 0, float(ref), float(amin)
    p.shape = g.lead + (g.n_frames, int(fb.shape[1]))
    p.device, p.dev_index = wave.device, wave.device.index
    p.layout = (wave.shape, wave.stride(), wave.dtype)
    p.cplan = None
    p.last_rc = 0
    ext = _native.ext()
    if ext is not None:
        try:
            p.cplan = ext.MelPlan(ctypes.cast(p.fn, ctypes.c_void_p).value, wave, window, fb, wpack, dsc, bytes(g.desc),
                                  [int(v) for v in info], p.power, p.n_mels, bool(db), p.ref, p.amin, [int(n) for n in p.shape],
                                  _epoch)
        except Exception:                       # noqa: BLE001 — the ctypes launcher above covers the call
            p.cplan = None
    return p


def melspectrogram(wave, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref,
                   amin):
    g = geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    if fb.dim() != 2 or fb.shape[0] != g.n_bins:
        raise RuntimeError('apply_filterbank: size mismatch, spectrogram has %d bins, filterbank %s'
                           % (g.n_bins, tuple(fb.shape)))
    # the route is cached per (filterbank object, version, power); ``id`` alone could be reused by a NEW tensor once
    # the old one is collected, so the entry carries a weak reference that must still point at this very object
    rkey = (id(fb), _stamp(fb), power, MEL_PATH)
    hit = g.routes.get(rkey)
    if hit is not None and hit[0]() is fb:
        route = hit[1]
    else:
        if len(g.routes) > 16:
            g.routes.clear()
        route = _fused_mel_route(g, fb, power)
        g.routes[rkey] = (weakref.ref(fb), route)
    if route is None:       # (fft_length 4096, |X|^p with p outside {1, 2}, banks the fused kernels reject)
        spec = spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power,
                           False, 1.0, 1e-7)
        return apply_filterbank(spec, fb, db=(ref, amin) if db else None)    # the dB epilogue rides on the filterbank kernel
    n_mels = fb.shape[1]
    src = _rows_of(wave, g)
    out = torch.empty(g.lead + (g.n_frames, n_mels), dtype=torch.float32, device=wave.device)
    if route == 'sparse':          # band-sparse contraction (the faster form for triangular banks)
        wpack, desc, info = _melbank_pack(fb, g.n_fft)
        with _native.on_device(wave.device):
            rc = _native.lib().tac_melspec_sparse_f32(
                _native.ptr(src), _native.ptr(window), g.desc, float(power), _native.ptr(wpack), _native.ptr(desc),
                ctypes.cast(info, ctypes.c_void_p), n_mels, 1 if db else 0, float(ref), float(amin),
                _native.ptr(out), _native.stream_ptr(wave.device))
        if rc == _native.TAC_E_UNSUPPORTED:         # a geometry the fused kernel of this size declines: the two-kernel chain
            spec = spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power,
                               False, 1.0, 1e-7)
            return apply_filterbank(spec, fb, db=(ref, amin) if db else None)
        _native.check(rc, 'tac_melspec_sparse_f32')
        _count('tac_melspec_sparse_f32')
        return out.transpose(-2, -1)
    _, plan_host = _filterbank_plan(fb)
    with _native.on_device(wave.device):
        rc = _native.lib().tac_melspec_f32(
            _native.ptr(src), _native.ptr(window), g.desc, float(power), _native.ptr(fb),
            ctypes.cast(plan_host, ctypes.c_void_p), n_mels, 1 if db else 0, float(ref), float(amin),
            _native.ptr(out), _native.stream_ptr(wave.device))
    _native.check(rc, 'tac_melspec_f32')
    _count('tac_melspec_f32')
    return out.transpose(-2, -1)


# ----------------------------------------------------------------------------- filterbank
def apply_filterbank(spec, fb, allow_sparse=True, db=None):
    """``db = (ref, amin)``: followed by ``amplitude_to_db`` — in the same launch when the band-sparse streaming kernel
    takes the call, as a second kernel behind the MFMA GEMM."""
    fb = fb if fb.is_contiguous() else fb.contiguous()
    n_freqs, n_frames = spec.shape[-2], spec.shape[-1]
    lead = tuple(spec.shape[:-2])
    n_mels = fb.shape[1]
    out = torch.empty(lead + (n_frames, n_mels), dtype=torch.float32, device=spec.device)
    if out.numel():
        rows = spec.reshape(-1, n_freqs, n_frames)
        # frame-major spectrogram (what the kernels here produce) + band-sparse bank: stream it through the fused
        # kernel's contraction; anything else goes through the fp32 MFMA GEMM
        pack = _melbank_pack(fb, 0) if (allow_sparse and rows.stride(1) == 1 and MEL_PATH != 'mfma') else None
        if pack is not None:
            wpack, desc, info = pack
            with _native.on_device(spec.device):
                if db is None:
                    name = 'tac_apply_filterbank_sparse_f32'
                    rc = _native.lib().tac_apply_filterbank_sparse_f32(
                        _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0) if rows.shape[0] > 1 else 0,
                        rows.stride(2), _native.ptr(wpack), _native.ptr(desc), ctypes.cast(info, ctypes.c_void_p), n_mels,
                        _native.ptr(out), _native.stream_ptr(spec.device))
                else:
                    name = 'tac_apply_filterbank_sparse_db_f32'
                    rc = _native.lib().tac_apply_filterbank_sparse_db_f32(
                        _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0) if rows.shape[0] > 1 else 0,
                        rows.stride(2), _native.ptr(wpack), _native.ptr(desc), ctypes.cast(info, ctypes.c_void_p), n_mels,
                        1, float(db[0]), float(db[1]), _native.ptr(out), _native.stream_ptr(spec.device))
            if rc != _native.TAC_E_UNSUPPORTED:
                _native.check(rc, name)
                _count(name)
                return out.transpose(-2, -1)
        plan, _ = _filterbank_plan(fb)
        with _native.on_device(spec.device):
            rc = _native.lib().tac_apply_filterbank_f32(
                _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0), rows.stride(1),
                rows.stride(2), _native.ptr(fb), _native.ptr(plan), n_mels, _native.ptr(out),
                _native.stream_ptr(spec.device))
        _native.check(rc, 'tac_apply_filterbank_f32')
        _count('tac_apply_filterbank_f32')
    out = out.transpose(-2, -1)
    return out if db is None else amplitude_to_db(out, db[0], db[1])


# ----------------------------------------------------------------------------- complex pairs
def is_dense(x):
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


def _pairs(z):
    """Dense view of a ``(*, 2)`` tensor whose storage order the elementwise kernels can walk pair by pair."""
    if z.stride(-1) != 1 or not is_dense(z) or any(s % 2 for s, n in zip(z.stride()[:-1], z.shape[:-1]) if n > 1):
        z = z.contiguous()
    return z


def _pair_output(z):
    return torch.empty_strided(z.shape[:-1], tuple(s // 2 for s in z.stride()[:-1]), dtype=torch.float32,
                               device=z.device)


def complex_norm(z, power):
    z = _pairs(z)
    out = _pair_output(z)
    n = out.numel()
    if n:
        with _native.on_device(z.device):
            rc = _native.lib().tac_complex_norm_f32(_native.ptr(z), n, float(power), _native.ptr(out),
                                                    _native.stream_ptr(z.device))
        _native.check(rc, 'tac_complex_norm_f32')
        _count('tac_complex_norm_f32')
    return out


def angle(z):
    z = _pairs(z)
    phase = _pair_output(z)
    if phase.numel():
        with _native.on_device(z.device):
            rc = _native.lib().tac_magphase_f32(_native.ptr(z), phase.numel(), 1.0, None, _native.ptr(phase),
                                                _native.stream_ptr(z.device))
        _native.check(rc, 'tac_magphase_f32')
        _count('tac_magphase_f32')
    return phase


def magphase(z, power):
    z = _pairs(z)
    mag, phase = _pair_output(z), _pair_output(z)
    if phase.numel():
        with _native.on_device(z.device):
            rc = _native.lib().tac_magphase_f32(_native.ptr(z), phase.numel(), float(power), _native.ptr(mag),
                                                _native.ptr(phase), _native.stream_ptr(z.device))
        _native.check(rc, 'tac_magphase_f32')
        _count('tac_magphase_f32')
    return mag, phase


_PV_GRID_CACHE = {}


def _phase_vocoder_grid(n_frames, rate, device, dtype):
    """Source-frame indices and interpolation weights of every output frame, evaluated exactly as the reference's CPU
    path does (functional.py:233-247: ``torch.arange(0, T, rate)`` in the default dtype, ``% 1``, ``.long()``); which
    frames get paired depends on that rounding, so it is computed with the same host ops and cached."""
    key = (int(n_frames), float(rate), str(device), dtype, torch.get_default_dtype())
    hit = _PV_GRID_CACHE.get(key)
    if hit is None:
        steps = torch.arange(0, n_frames, rate)
        if len(_PV_GRID_CACHE) > 64:
            _PV_GRID_CACHE.clear()
        hit = (steps.long().to(torch.int32).to(device), (steps + 1).long().to(torch.int32).to(device),
               torch.remainder(steps, torch.tensor(1., dtype=steps.dtype)).to(dtype).to(device))
        _PV_GRID_CACHE[key] = hit
    return hit


def phase_vocoder_out_frames(n_frames, rate):
    return int(torch.arange(0, n_frames, rate).numel())


def phase_vocoder(spec, rate, phase_advance):
    """float32 or float64 (the reference's own test runs this op in float64, tests/test_functional.py:69-116)."""
    dtype = spec.dtype
    n_freqs, n_frames = spec.shape[-3], spec.shape[-2]
    pa = phase_advance.reshape(-1).to(dtype).contiguous()
    lead = tuple(spec.shape[:-3])
    idx0, idx1, alpha = _phase_vocoder_grid(n_frames, rate, spec.device, dtype)
    n_out = idx0.numel()
    if spec.stride(-1) != 1:
        spec = spec.contiguous()
    rows = spec.reshape((-1,) + tuple(spec.shape[-3:]))          # a view whenever the leading dims collapse
    if any(st % 2 for st in rows.stride()[:-1]) or rows.data_ptr() % (2 * rows.element_size()):
        rows = rows.contiguous()                                 # (re, im) pairs are fetched as one aligned access
    out = torch.empty(lead + (n_out, n_freqs, 2), dtype=dtype, device=spec.device)
    if out.numel() and n_frames:
        name = 'tac_phase_vocoder_f64' if dtype == torch.float64 else 'tac_phase_vocoder_f32'
        with _native.on_device(spec.device):
            rc = getattr(_native.lib(), name)(
                _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0) if rows.shape[0] > 1 else 0,
                rows.stride(1), rows.stride(2), _native.ptr(pa), _native.ptr(idx0), _native.ptr(idx1),
                _native.ptr(alpha), n_out, _native.ptr(out), _native.stream_ptr(spec.device))
        _native.check(rc, name)
        _count(name)
    return out.transpose(-3, -2)


def phase_vocoder_backward(spec, rate, grad_out):
    """Gradient of the float32 ``phase_vocoder`` with respect to ``spec`` (*, F, T, 2): ``grad_out`` (*, F, n_out, 2) in any layout;
    the result is frame-major like the forward's output."""
    n_freqs, n_frames = spec.shape[-3], spec.shape[-2]
    lead = tuple(spec.shape[:-3])
    idx0, idx1, alpha = _phase_vocoder_grid(n_frames, rate, spec.device, torch.float32)
    n_out = idx0.numel()
    if spec.stride(-1) != 1:
        spec = spec.contiguous()
    rows = spec.reshape((-1,) + tuple(spec.shape[-3:]))
    if any(st % 2 for st in rows.stride()[:-1]) or rows.data_ptr() % (2 * rows.element_size()):
        rows = rows.contiguous()
    go = torch.empty(lead + (n_out, n_freqs, 2), dtype=torch.float32, device=spec.device)      # frame-major, dense
    go.transpose(-3, -2).copy_(grad_out)
    gs = torch.zeros(lead + (n_frames, n_freqs, 2), dtype=torch.float32, device=spec.device)    # the kernel accumulates into it
    if go.numel() and n_frames:
        with _native.on_device(spec.device):
            rc = _native.lib().tac_phase_vocoder_backward_f32(
                _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0) if rows.shape[0] > 1 else 0,
                rows.stride(1), rows.stride(2), _native.ptr(idx0), _native.ptr(idx1), _native.ptr(alpha), n_out,
                _native.ptr(go), _native.ptr(gs), _native.stream_ptr(spec.device))
        _native.check(rc, 'tac_phase_vocoder_backward_f32')
        _count('tac_phase_vocoder_backward_f32')
    return gs.transpose(-3, -2)


# ----------------------------------------------------------------------------- elementwise
def _unary(x, name, launch):
    x = x if is_dense(x) else x.contiguous()
    out = torch.empty_like(x)
    if x.numel():
        with _native.on_device(x.device):
            rc = launch(_native.lib(), _native.ptr(x), x.numel(), _native.ptr(out), _native.stream_ptr(x.device))
        _native.check(rc, name)
        _count(name)
    return out


def amplitude_to_db(x, ref, amin):
    return _unary(x, 'tac_amplitude_to_db_f32',
                  lambda h, p, n, o, s: h.tac_amplitude_to_db_f32(p, n, float(ref), float(amin), o, s))


def db_to_amplitude(x, ref):
    return _unary(x, 'tac_db_to_amplitude_f32', lambda h, p, n, o, s: h.tac_db_to_amplitude_f32(p, n, float(ref), o, s))


_mulaw_consts = {}


def _mulaw_tables(device):
    key = str(device)
    hit = _mulaw_consts.get(key)
    if hit is None:
        from . import _mulaw_tables as tab
        thr = torch.tensor(list(tab.THR256_POS) + list(tab.THR256_NEG), dtype=torch.int32, device=device)
        lut = torch.tensor(list(tab.LUT256_BITS), dtype=torch.int64).to(torch.int32).view(torch.float32)
        hit = (thr, len(tab.THR256_POS), len(tab.THR256_NEG), tab.ZERO_CODE_256, lut.to(device))
        with _lock:
            _mulaw_consts[key] = hit
    return hit


def mu_law_encoding(x, n_quantize):
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    if x.numel():
        if n_quantize == 256:
            thr, n_pos, n_neg, zero, _ = _mulaw_tables(x.device)
            thr_ptr = _native.ptr(thr)
        else:
            thr_ptr, n_pos, n_neg, zero = None, 0, 0, 0
        with _native.on_device(x.device):
            rc = _native.lib().tac_mulaw_encode_f32_i64(_native.ptr(x), x.numel(), n_quantize, thr_ptr, n_pos, n_neg,
                                                        zero, _native.ptr(out), _native.stream_ptr(x.device))
        _native.check(rc, 'tac_mulaw_encode_f32_i64')
        _count('tac_mulaw_encode_f32_i64')
    return out


def mu_law_decoding_int(codes, n_quantize):
    """int64 codes -> float32 (reference functional.py:348-354 with the default dtype)."""
    codes = codes.to(torch.int64).contiguous()
    out = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    if codes.numel():
        lut_ptr = _native.ptr(_mulaw_tables(codes.device)[4]) if n_quantize == 256 else None
        with _native.on_device(codes.device):
            rc = _native.lib().tac_mulaw_decode_i64_f32(_native.ptr(codes), codes.numel(), n_quantize, lut_ptr,
                                                        _native.ptr(out), _native.stream_ptr(codes.device))
        _native.check(rc, 'tac_mulaw_decode_i64_f32')
        _count('tac_mulaw_decode_i64_f32')
    return out


def mu_law_decoding_float(codes, n_quantize):
    """float32 codes -> float32: integral codes in [0, 256) with n_quantize == 256 come from the reference's own
    table (bit-exact, what reference tests/test_functional.py:182-193 bit-compares), everything else from the
    closed form."""
    lut = _mulaw_tables(codes.device)[4] if n_quantize == 256 else None
    return _unary(codes, 'tac_mulaw_decode_f32_f32',
                  lambda h, p, n, o, s: h.tac_mulaw_decode_f32_f32(p, n, n_quantize,
                                                                   None if lut is None else _native.ptr(lut), o, s))


def mu_law_encoding_f64(x, n_quantize):
    """float64 waveform -> int64 codes, the formula in double (reference functional.py:329-335 on double input)."""
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    if x.numel():
        with _native.on_device(x.device):
            rc = _native.lib().tac_mulaw_encode_f64_i64(_native.ptr(x), x.numel(), n_quantize, _native.ptr(out),
                                                        _native.stream_ptr(x.device))
        _native.check(rc, 'tac_mulaw_encode_f64_i64')
        _count('tac_mulaw_encode_f64_i64')
    return out


def mu_law_decoding_f64(codes, n_quantize):
    """int64 or float64 codes -> float64 (reference functional.py:349-354 evaluated in double)."""
    codes = codes if codes.is_contiguous() else codes.contiguous()
    out = torch.empty(codes.shape, dtype=torch.float64, device=codes.device)
    if codes.numel():
        with _native.on_device(codes.device):
            rc = _native.lib().tac_mulaw_decode_f64(_native.ptr(codes), 1 if codes.dtype == torch.int64 else 0, codes.numel(),
                                                    n_quantize, _native.ptr(out), _native.stream_ptr(codes.device))
        _native.check(rc, 'tac_mulaw_decode_f64')
        _count('tac_mulaw_decode_f64')
    return out


# ----------------------------------------------------------------------------- gradients
def transposed_bank(fb):
    """(M, F) contiguous transpose of a filterbank, cached on the tensor per version (the adjoint of the filterbank
    stage is the same GEMM kernel with this matrix)."""
    hit = getattr(fb, '_tac_T', None)
    if hit is not None and hit[0] == _stamp(fb):
        return hit[1]
    t = fb.detach().t().contiguous()
    try:
        fb._tac_T = (_stamp(fb), t)
    except Exception:
        pass
    return t


def _adjoint_table(fb):
    """Per-bin {w0, w1, band0, band1} table of a bank with at most two non-zero weights per bin (every triangular mel
    bank), built on the device once per filterbank version (one host sync) and cached on the tensor; None otherwise."""
    hit = getattr(fb, '_tac_adj', None)
    if hit is not None and hit[0] == _stamp(fb):
        return hit[1]
    n_freqs, n_mels = fb.shape
    table = None
    if n_mels <= 512 and 16 * n_freqs + 4 * 4 * n_mels <= 64 * 1024:
        src = fb if fb.is_contiguous() else fb.contiguous()         # (the cache stays on the caller's tensor object)
        table = torch.empty(4 * n_freqs + 4, dtype=torch.float32, device=fb.device)
        nnz = ctypes.c_int32(0)
        with _native.on_device(fb.device):
            rc = _native.lib().tac_filterbank_adjoint_pack(_native.ptr(src), n_freqs, n_mels, _native.ptr(table),
                                                           ctypes.cast(ctypes.pointer(nnz), ctypes.c_void_p),
                                                           _native.stream_ptr(fb.device))
        _native.check(rc, 'tac_filterbank_adjoint_pack')
        if nnz.value > 2:
            table = None
    try:
        fb._tac_adj = (_stamp(fb), table)
    except Exception:
        pass
    return table


def apply_filterbank_backward(grad_out, fb):
    """(*, M, T) gradient -> (*, F, T): two multiply-adds per output through the per-bin table of a bank with at most
    two non-zero weights per bin (tac_apply_filterbank_adjoint_f32); any other bank: the forward MFMA GEMM with the
    transposed bank."""
    table = _adjoint_table(fb) if (grad_out.dim() >= 2 and fb.dim() == 2 and MEL_PATH != 'mfma') else None
    if table is not None:
        n_freqs, n_mels = fb.shape
        gm = grad_out.transpose(-2, -1)                               # physical frame-major (*, T, M)
        gm = gm if gm.is_contiguous() else gm.contiguous()
        if gm.dtype != torch.float32:
            gm = gm.float()
        out = torch.empty(tuple(gm.shape[:-1]) + (n_freqs,), dtype=torch.float32, device=gm.device)
        with _native.on_device(gm.device):
            rc = _native.lib().tac_apply_filterbank_adjoint_f32(_native.ptr(gm), gm.numel() // n_mels, n_mels,
                                                                _native.ptr(table), n_freqs, _native.ptr(out),
                                                                _native.stream_ptr(gm.device))
        if rc != _native.TAC_E_UNSUPPORTED:
            _native.check(rc, 'tac_apply_filterbank_adjoint_f32')
            _count('tac_apply_filterbank_adjoint_f32')
            return out.transpose(-2, -1)
    return apply_filterbank(grad_out, transposed_bank(fb), allow_sparse=False)


def melspectrogram_backward_fused(grad_mel, wave, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, power):
    """Gradient of the waveform from the gradient of the (linear) mel values ``(*, M, T)`` in ONE kernel: the filterbank
    adjoint is formed per frame inside the backward kernel (tac_melspectrogram_backward_ola_f32), so the gradient of the
    power spectrogram — 4·F bytes per frame written by the adjoint kernel and read back by the backward kernel — never
    exists.  None when the form does not cover the case (fft_length other than 512 / 1024 / 2048, too many bands, a hop the
    register-ring kernels are not instantiated for at 512 / 1024, a bank with
    more than two non-zero weights per bin): the caller then runs the two kernels."""
    if n_fft not in (400, 512, 1024, 2048) or fb.dim() != 2 or fb.shape[1] > (256 if n_fft == 2048 else 128) or MEL_PATH == 'mfma':
        return None
    if n_fft == 400:        # overlap-add inside the kernel when the hop allows it, else frame gradients + the gather kernel
        out = _melspectrogram_backward_ola(grad_mel, wave, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, power)
        if out is not None:
            return out
        return _melspectrogram_backward_fused_n400(grad_mel, wave, window, fb, hop, win_length, center, pad_mode, normalized, power)
    return _melspectrogram_backward_ola(grad_mel, wave, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, power)


def _melspectrogram_backward_ola(grad_mel, wave, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, power):
    table = _adjoint_table(fb)
    if table is None:
        return None
    g = geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, True)
    if g.desc is None:
        return None
    need = _native.lib().tac_spectrogram_backward_ola_workspace(g.desc)
    if need < 0:
        return None
    n_freqs, n_mels = fb.shape
    gm = grad_mel.transpose(-2, -1)                                   # physical frame-major (*, T, M)
    gm = gm if gm.is_contiguous() else gm.contiguous()
    if gm.dtype != torch.float32:
        gm = gm.float()
    out = torch.empty(tuple(wave.shape), dtype=torch.float32, device=wave.device)
    if POISON_OUTPUTS:
        out.fill_(float('nan'))
    work = torch.empty(max(int(need), 4) // 4, dtype=torch.float32, device=wave.device)
    with _native.on_device(wave.device):
        rc = _native.lib().tac_melspectrogram_backward_ola_f32(
            _native.ptr(_rows_of(wave, g)), _native.ptr(window), g.desc, _native.ptr(gm), n_mels, _native.ptr(table),
            n_freqs, float(power), _native.ptr(work), int(need), _native.ptr(out), g.length,
            _native.stream_ptr(wave.device))
    if rc == _native.TAC_E_UNSUPPORTED:
        return None
    _native.check(rc, 'tac_melspectrogram_backward_ola_f32')
    _count('tac_melspectrogram_backward_ola_f32')
    return out


def _melspectrogram_backward_fused_n400(grad_mel, wave, window, fb, hop, win_length, center, pad_mode, normalized, power):
    """fft_length 400: the mixed-radix backward kernel forms the filterbank adjoint itself (tac_melspectrogram_backward_f32),
    frame gradients cross memory once, gather overlap-add follows (tac_overlap_add_f32)."""
    table = _adjoint_table(fb)
    if table is None:
        return None
    g = geometry(wave, 400, hop, win_length, center, pad_mode, normalized, True)
    if g.desc is None:
        return None
    n_freqs, n_mels = fb.shape
    gm = grad_mel.transpose(-2, -1)                                   # physical frame-major (*, T, M)
    gm = gm if gm.is_contiguous() else gm.contiguous()
    if gm.dtype != torch.float32:
        gm = gm.float()
    frames = torch.empty((g.rows, g.n_frames, 400), dtype=torch.float32, device=wave.device)
    out = torch.empty(tuple(wave.shape), dtype=torch.float32, device=wave.device)
    desc = _native.StftDesc(rows=g.rows, length=g.length, row_stride=g.length, n_fft=400, hop=hop, win_length=win_length,
                            center=1 if center else 0, pad_mode=_native.PAD_MODES[pad_mode],
                            normalized=1 if normalized else 0, onesided=1, reserved=0)
    with _native.on_device(wave.device):
        rc = _native.lib().tac_melspectrogram_backward_f32(
            _native.ptr(_rows_of(wave, g)), _native.ptr(window), g.desc, _native.ptr(gm), n_mels, _native.ptr(table), n_freqs,
            float(power), _native.ptr(frames), _native.stream_ptr(wave.device))
        if rc == _native.TAC_E_UNSUPPORTED:
            return None
        _native.check(rc, 'tac_melspectrogram_backward_f32')
        _count('tac_melspectrogram_backward_f32')
        rc = _native.lib().tac_overlap_add_f32(_native.ptr(frames), desc, _native.ptr(out), g.length,
                                               _native.stream_ptr(wave.device))
        _native.check(rc, 'tac_overlap_add_f32')
        _count('tac_overlap_add_f32')
    return out


def stft_backward(grad_spec, wave, window, n_fft, hop, win_length, center, pad_mode, normalized, grad_norm=None,
                  power=2.0):
    """grad of the one-sided stft output ``(*, F, T, 2)`` w.r.t. the waveform: one inverse real FFT per frame
    (tac_stft_backward_f32) followed by the gather form of overlap-add (tac_overlap_add_f32).
    With ``grad_norm`` (the gradient of ``complex_norm(spec, power)``, shape ``(*, F, T)``) ``grad_spec`` is the
    spectrum itself and the norm's adjoint is folded into the load (tac_stft_norm_backward_f32) — or ``None``: the
    kernel then transforms the frames of ``wave`` again itself and no spectrum exists in memory
    (tac_spectrogram_backward_f32)."""
    g = geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, True)
    gs = None
    if grad_spec is not None:
        gs = grad_spec.transpose(-3, -2)                              # physical frame-major (*, T, F, 2)
        gs = gs if gs.is_contiguous() else gs.contiguous()
    gn = None
    if grad_norm is not None:
        gn = grad_norm.transpose(-2, -1)                              # (*, T, F)
        gn = gn if gn.is_contiguous() else gn.contiguous()
        if gn.dtype != torch.float32:
            gn = gn.float()
    out = torch.empty(tuple(wave.shape), dtype=torch.float32, device=wave.device)
    if POISON_OUTPUTS:
        out.fill_(float('nan'))
    if gs is None and g.desc is not None:
        # fft_length 256 … 2048 with hop a multiple of fft_length / 16, fft_length 400 with hop a multiple of 4: overlap-add
        # inside the kernel, no frame gradients in memory
        need = _native.lib().tac_spectrogram_backward_ola_workspace(g.desc)
        if need >= 0:
            work = torch.empty(max(int(need), 4) // 4, dtype=torch.float32, device=wave.device)
            with _native.on_device(wave.device):
                rc = _native.lib().tac_spectrogram_backward_ola_f32(
                    _native.ptr(_rows_of(wave, g)), _native.ptr(window), g.desc, _native.ptr(gn), float(power),
                    _native.ptr(work), int(need), _native.ptr(out), g.length, _native.stream_ptr(wave.device))
            if rc != _native.TAC_E_UNSUPPORTED:
                _native.check(rc, 'tac_spectrogram_backward_ola_f32')
                _count('tac_spectrogram_backward_ola_f32')
                return out
    frames = torch.empty((g.rows, g.n_frames, n_fft), dtype=torch.float32, device=wave.device)
    desc = _native.StftDesc(rows=g.rows, length=g.length, row_stride=g.length, n_fft=n_fft, hop=hop,
                            win_length=win_length, center=1 if center else 0, pad_mode=_native.PAD_MODES[pad_mode],
                            normalized=1 if normalized else 0, onesided=1, reserved=0)
    with _native.on_device(wave.device):
        if gn is None:
            rc = _native.lib().tac_stft_backward_f32(_native.ptr(gs), _native.ptr(window), desc, _native.ptr(frames),
                                                     _native.stream_ptr(wave.device))
            _native.check(rc, 'tac_stft_backward_f32')
            _count('tac_stft_backward_f32')
        elif gs is None:
            rc = _native.lib().tac_spectrogram_backward_f32(_native.ptr(_rows_of(wave, g)), _native.ptr(window), g.desc,
                                                            _native.ptr(gn), float(power), _native.ptr(frames),
                                                            _native.stream_ptr(wave.device))
            _native.check(rc, 'tac_spectrogram_backward_f32')
            _count('tac_spectrogram_backward_f32')
        else:
            rc = _native.lib().tac_stft_norm_backward_f32(_native.ptr(gs), _native.ptr(gn), float(power), _native.ptr(window),
                                                          desc, _native.ptr(frames), _native.stream_ptr(wave.device))
            _native.check(rc, 'tac_stft_norm_backward_f32')
            _count('tac_stft_norm_backward_f32')
        rc = _native.lib().tac_overlap_add_f32(_native.ptr(frames), desc, _native.ptr(out), g.length,
                                               _native.stream_ptr(wave.device))
        _native.check(rc, 'tac_overlap_add_f32')
        _count('tac_overlap_add_f32')
    return out


# ---- the general gradient routes: every fft_length the forward kernels cover, two-sided outputs, and the gradients of
# the window and the filterbank (the reference differentiates through every argument, functional.py:99-107, 183-184).
# Built from the unfused pieces — inverse FFT per frame or the DFT-matrix GEMM with the transposed matrix, gather
# overlap-add, a frames x samples reduction for the window, the MFMA GEMM for the filterbank — so frame gradients do go
# through memory here; the fused backward kernels above remain the route of the common case (gradient of the waveform
# only, one-sided power-of-two sizes).
def fold_twosided(grad, n_fft, width):
    """physical frame-major gradient of a two-sided output (*, T, n_fft[, 2]) -> the one-sided bins (*, T, F[, 2])."""
    n_bins = n_fft // 2 + 1
    lead = tuple(grad.shape[:-2]) if width == 2 else tuple(grad.shape[:-1])
    out = torch.empty(lead + ((n_bins, 2) if width == 2 else (n_bins,)), dtype=torch.float32, device=grad.device)
    frames = out.numel() // (n_bins * width)
    with _native.on_device(grad.device):
        rc = _native.lib().tac_fold_twosided_f32(_native.ptr(grad), frames, n_fft, width, _native.ptr(out),
                                                 _native.stream_ptr(grad.device))
    _native.check(rc, 'tac_fold_twosided_f32')
    _count('tac_fold_twosided_f32')
    return out


def sum_slabs(x):
    """(S, ...) -> (...): the slabs added up in order (deterministic)."""
    out = torch.empty(tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
    with _native.on_device(x.device):
        rc = _native.lib().tac_sum_slabs_f32(_native.ptr(x), x.shape[0], out.numel(), _native.ptr(out),
                                             _native.stream_ptr(x.device))
    _native.check(rc, 'tac_sum_slabs_f32')
    _count('tac_sum_slabs_f32')
    return out


_ones_cache = {}


def _ones_window(device, n):
    key = (str(device), n)
    w = _ones_cache.get(key)
    if w is None:
        with torch.inference_mode(False):
            w = torch.ones(n, dtype=torch.float32, device=device)
        _ones_cache[key] = w
    return w


def _dft_matrix_t(window, n_fft, win_length, normalized):
    """(2F, N) transpose of the one-sided windowed-DFT matrix: the adjoint of ``_stft_dft`` is the same GEMM with it."""
    cache = getattr(window, '_tac_dftT', None)
    key = (_stamp(window), n_fft, win_length, bool(normalized))
    if cache is not None and cache[0] == key:
        return cache[1]
    mat = _dft_matrix(window, n_fft, win_length, True, normalized).t().contiguous()
    try:
        window._tac_dftT = (key, mat)
    except Exception:
        pass
    return mat


def _desc(g, row_stride=None, onesided=None):
    return _native.StftDesc(rows=g.rows, length=g.length, row_stride=g.length if row_stride is None else row_stride,
                            n_fft=g.n_fft, hop=g.hop, win_length=g.win_length, center=1 if g.center else 0,
                            pad_mode=_native.PAD_MODES[g.pad_mode], normalized=1 if g.normalized else 0,
                            onesided=(1 if g.onesided else 0) if onesided is None else onesided, reserved=0)


def _frame_gradients(gs, window, g):
    """one-sided gradient spectrum (rows, T, F, 2) -> frame gradients (rows, T, n_fft), window and scale applied."""
    frames = torch.empty((g.rows, g.n_frames, g.n_fft), dtype=torch.float32, device=gs.device)
    with _native.on_device(gs.device):
        if fft_kernel_size(g.n_fft) or g.mixed_radix or smooth_fft_size(g.n_fft) or g.n_fft == 8192:
            rc = _native.lib().tac_stft_backward_f32(_native.ptr(gs), _native.ptr(window), _desc(g, onesided=1),
                                                     _native.ptr(frames), _native.stream_ptr(gs.device))
            _native.check(rc, 'tac_stft_backward_f32')
            _count('tac_stft_backward_f32')
        else:
            n_cols = 2 * (g.n_fft // 2 + 1)
            mat_t = _dft_matrix_t(window, g.n_fft, g.win_length, g.normalized)
            rc = _native.lib().tac_apply_filterbank_f32(
                _native.ptr(gs), g.rows, n_cols, g.n_frames, g.n_frames * n_cols, 1, n_cols, _native.ptr(mat_t), None,
                g.n_fft, _native.ptr(frames), _native.stream_ptr(gs.device))
            _native.check(rc, 'tac_apply_filterbank_f32 (DFT matrix, adjoint)')
            _count('tac_apply_filterbank_f32')
    return frames


def stft_backward_general(grad_spec, wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided,
                          need_wave=True, need_window=False):
    """(grad_wave, grad_window) from the gradient of the complex stft output ``(*, n_bins, T, 2)`` — any fft_length up
    to 8192, one- or two-sided."""
    g = geometry(wave, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    gs = grad_spec.transpose(-3, -2)                                   # physical frame-major (*, T, n_bins, 2)
    gs = gs if (gs.is_contiguous() and gs.dtype == torch.float32) else gs.contiguous().float()
    if not g.onesided:
        gs = fold_twosided(gs, n_fft, 2)
    window = window if window.is_contiguous() else window.contiguous()
    dev = wave.device
    grad_wave = grad_window = None
    if need_wave:
        frames = _frame_gradients(gs, window, g)
        grad_wave = torch.empty(tuple(wave.shape), dtype=torch.float32, device=dev)
        with _native.on_device(dev):
            rc = _native.lib().tac_overlap_add_f32(_native.ptr(frames), _desc(g), _native.ptr(grad_wave), g.length,
                                                   _native.stream_ptr(dev))
        _native.check(rc, 'tac_overlap_add_f32')
        _count('tac_overlap_add_f32')
        del frames
    if need_window:
        unwindowed = _frame_gradients(gs, _ones_window(dev, win_length), g)
        src = _rows_of(wave, g)
        desc = _desc(g, row_stride=g.row_stride)
        n_part = int(_native.lib().tac_window_grad_partials(desc))
        if n_part < 0:
            _native.check(n_part, 'tac_window_grad_partials')
        partial = torch.empty((n_part, n_fft), dtype=torch.float32, device=dev)
        with _native.on_device(dev):
            rc = _native.lib().tac_window_grad_f32(_native.ptr(unwindowed), _native.ptr(src), desc, _native.ptr(partial),
                                                   n_part, _native.stream_ptr(dev))
        _native.check(rc, 'tac_window_grad_f32')
        _count('tac_window_grad_f32')
        off = (n_fft - win_length) // 2
        grad_window = sum_slabs(partial)[off:off + win_length]
    return grad_wave, grad_window


def filterbank_grad(spec, grad_out):
    """d/d filterbank of ``apply_filterbank(spec, fb)``: grad_fb[f][m] = sum over (*, t) of spec[.., f, t] *
    grad_out[.., m, t] — one fp32 MFMA GEMM whose contraction runs over every frame of the batch."""
    sp = spec.transpose(-2, -1)                                        # (*, T, F)
    sp = sp if (sp.is_contiguous() and sp.dtype == torch.float32) else sp.contiguous().float()
    gm = grad_out.transpose(-2, -1)                                    # (*, T, M)
    gm = gm if (gm.is_contiguous() and gm.dtype == torch.float32) else gm.contiguous().float()
    n_freqs, n_mels = sp.shape[-1], gm.shape[-1]
    total = sp.numel() // n_freqs
    if total >= 2 ** 31:
        raise NotImplementedError('filterbank gradient: more than 2^31 frames in one call')
    out = torch.empty((n_freqs, n_mels), dtype=torch.float32, device=sp.device)
    if total == 0:
        return out.zero_()
    with _native.on_device(sp.device):
        rc = _native.lib().tac_apply_filterbank_f32(_native.ptr(sp), 1, total, n_freqs, 0, n_freqs, 1, _native.ptr(gm), None,
                                                    n_mels, _native.ptr(out), _native.stream_ptr(sp.device))
    _native.check(rc, 'tac_apply_filterbank_f32 (filterbank gradient)')
    _count('tac_apply_filterbank_f32')
    return out


def stft_backward_supported(n_fft, onesided):
    """one-sided power-of-two sizes and fft_length 400: an inverse-FFT kernel per frame (csrc/backward.hip,
    csrc/stft_n400.hip); everything else takes ``stft_backward_general``"""
    return bool(onesided) and (fft_kernel_size(n_fft) or mixed_radix_size(n_fft))


def backward_recomputes_spectrum(n_fft):
    """Sizes whose spectrogram backward kernel transforms the frames again itself (16 elements per lane, and the
    mixed-radix fft_length 400; at 4096 the second transform does not fit the registers and the spectrum is recomputed
    into memory by the stft kernel)."""
    return (fft_kernel_size(n_fft) and n_fft <= 2048) or mixed_radix_size(n_fft)


def complex_norm_backward(z, grad_out, power):
    """(*, F, T, 2) pairs and (*, F, T) gradients in, (*, F, T, 2) out — all walked in z's dense storage order."""
    z = _pairs(z)
    want = tuple(s // 2 for s in z.stride()[:-1])
    if grad_out.dtype == torch.float32 and grad_out.shape == z.shape[:-1] and grad_out.stride() == want:
        go = grad_out                                                 # already in z's storage order: no copy
    else:
        go = torch.empty_strided(z.shape[:-1], want, dtype=torch.float32, device=z.device)
        go.copy_(grad_out)
    gz = torch.empty_strided(z.shape, z.stride(), dtype=torch.float32, device=z.device)
    n = go.numel()
    if n:
        with _native.on_device(z.device):
            rc = _native.lib().tac_complex_norm_backward_f32(_native.ptr(z), _native.ptr(go), n, float(power),
                                                             _native.ptr(gz), _native.stream_ptr(z.device))
        _native.check(rc, 'tac_complex_norm_backward_f32')
        _count('tac_complex_norm_backward_f32')
    return gz


def magphase_backward(z, grad_mag, grad_phase, power):
    """Gradient of ``magphase`` / ``angle``: (*, F, T, 2) pairs and one or two (*, F, T) gradients in (either may be None), (*, F, T, 2)
    out — all walked in z's dense storage order."""
    z = _pairs(z)
    want = tuple(s // 2 for s in z.stride()[:-1])

    def ordered(g):
        if g is None or (g.dtype == torch.float32 and g.shape == z.shape[:-1] and g.stride() == want):
            return g                                                  # already in z's storage order: no copy
        go = torch.empty_strided(z.shape[:-1], want, dtype=torch.float32, device=z.device)
        go.copy_(g)
        return go
    gm, gp = ordered(grad_mag), ordered(grad_phase)
    gz = torch.empty_strided(z.shape, z.stride(), dtype=torch.float32, device=z.device)
    n = z.numel() // 2
    if n:
        with _native.on_device(z.device):
            rc = _native.lib().tac_magphase_backward_f32(_native.ptr(z), None if gm is None else _native.ptr(gm),
                                                         None if gp is None else _native.ptr(gp), n, float(power),
                                                         _native.ptr(gz), _native.stream_ptr(z.device))
        _native.check(rc, 'tac_magphase_backward_f32')
        _count('tac_magphase_backward_f32')
    return gz


def db_to_amplitude_backward(x, grad_out, ref):
    x = x if is_dense(x) else x.contiguous()
    if grad_out.dtype == torch.float32 and grad_out.shape == x.shape and grad_out.stride() == x.stride():
        go = grad_out                                                 # already in x's storage order: no copy
    else:
        go = torch.empty_like(x)
        go.copy_(grad_out)
    gx = torch.empty_like(x)
    if x.numel():
        with _native.on_device(x.device):
            rc = _native.lib().tac_db_to_amplitude_backward_f32(_native.ptr(x), _native.ptr(go), x.numel(), float(ref),
                                                                _native.ptr(gx), _native.stream_ptr(x.device))
        _native.check(rc, 'tac_db_to_amplitude_backward_f32')
        _count('tac_db_to_amplitude_backward_f32')
    return gx


def amplitude_to_db_backward(x, grad_out, amin):
    x = x if is_dense(x) else x.contiguous()
    if grad_out.dtype == torch.float32 and grad_out.shape == x.shape and grad_out.stride() == x.stride():
        go = grad_out                                                 # already in x's storage order: no copy
    else:
        go = torch.empty_like(x)
        go.copy_(grad_out)
    gx = torch.empty_like(x)
    if x.numel():
        with _native.on_device(x.device):
            rc = _native.lib().tac_amplitude_to_db_backward_f32(_native.ptr(x), _native.ptr(go), x.numel(), float(amin),
                                                                _native.ptr(gx), _native.stream_ptr(x.device))
        _native.check(rc, 'tac_amplitude_to_db_backward_f32')
        _count('tac_amplitude_to_db_backward_f32')
    return gx


# ----------------------------------------------------------------------------- hpss
def hpss_supported(kernel_f, kernel_t):
    return kernel_f % 2 == 1 and kernel_t % 2 == 1 and 1 <= kernel_f <= 63 and 1 <= kernel_t <= 63


def hpss(mag, kernel_f, kernel_t, power, hard, masks_only=False):
    """(harm, perc, mask_harm, mask_perc), float32, in the (dense) layout of ``mag``  (*, F, T); with ``masks_only`` the two
    masks alone (the kernels then skip the two masked-spectrogram stores: 40 % of the traffic)."""
    mag = mag if is_dense(mag) else mag.contiguous()
    n_freqs, n_frames = mag.shape[-2], mag.shape[-1]
    rows = mag.reshape(-1, n_freqs, n_frames)
    if rows.data_ptr() != mag.data_ptr():                          # leading dims do not collapse in this layout
        mag = mag.contiguous()
        rows = mag.reshape(-1, n_freqs, n_frames)
    outs = [torch.empty_like(mag) for _ in range(2 if masks_only else 4)]
    if mag.numel():
        null = ctypes.c_void_p(0)
        with _native.on_device(mag.device):
            rc = _native.lib().tac_hpss_f32(_native.ptr(rows), rows.shape[0], n_freqs, n_frames,
                                            rows.stride(0) if rows.shape[0] > 1 else 0, rows.stride(1), rows.stride(2),
                                            kernel_f, kernel_t, float(power), 1 if hard else 0,
                                            null if masks_only else _native.ptr(outs[0]),
                                            null if masks_only else _native.ptr(outs[1]), _native.ptr(outs[-2]),
                                            _native.ptr(outs[-1]), _native.stream_ptr(mag.device))
        _native.check(rc, 'tac_hpss_f32')
        _count('tac_hpss_f32')
    return tuple(outs)


def hpss_backward(mag, kernel_f, kernel_t, power, hard, grads):
    """d hpss / d mag: ``grads`` = the gradients of (harm, perc, mask_harm, mask_perc), each a tensor of mag's shape or None; the result
    is in mag's (dense) layout.  The gradient of a median goes to the element it selected (float atomics: the last bits vary)."""
    mag = mag if is_dense(mag) else mag.contiguous()
    n_freqs, n_frames = mag.shape[-2], mag.shape[-1]
    rows = mag.reshape(-1, n_freqs, n_frames)
    if rows.data_ptr() != mag.data_ptr():
        mag = mag.contiguous()
        rows = mag.reshape(-1, n_freqs, n_frames)

    def like_mag(g):
        if g is None:
            return None
        if g.dtype == torch.float32 and g.shape == mag.shape and g.stride() == mag.stride():
            return g
        out = torch.empty_like(mag)
        out.copy_(g)
        return out
    gs = [like_mag(g) for g in grads]
    gmag = torch.zeros_like(mag)                                   # the kernel accumulates into it
    if mag.numel():
        null = ctypes.c_void_p(0)
        with _native.on_device(mag.device):
            rc = _native.lib().tac_hpss_backward_f32(
                _native.ptr(rows), rows.shape[0], n_freqs, n_frames, rows.stride(0) if rows.shape[0] > 1 else 0, rows.stride(1),
                rows.stride(2), kernel_f, kernel_t, float(power), 1 if hard else 0,
                *[null if g is None else _native.ptr(g) for g in gs], _native.ptr(gmag), _native.stream_ptr(mag.device))
        _native.check(rc, 'tac_hpss_backward_f32')
        _count('tac_hpss_backward_f32')
    return gmag


# ----------------------------------------------------------------------------- coded waveforms (int16 PCM, mu-law codes)
def pcm16_to_f32(x):
    """int16 PCM -> float32 in [-1, 1): x * 2^-15 (for the kernels without a coded frame load)."""
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if x.numel():
        with _native.on_device(x.device):
            rc = _native.lib().tac_pcm16_to_f32(_native.ptr(x), x.numel(), _native.ptr(out), _native.stream_ptr(x.device))
        _native.check(rc, 'tac_pcm16_to_f32')
        _count('tac_pcm16_to_f32')
    return out


_SAMPLE_FORMATS = {torch.int16: _native.SAMPLES_I16, torch.uint8: _native.SAMPLES_MULAW_U8,
                   torch.int64: _native.SAMPLES_MULAW_I64}


def melspectrogram_coded(samples, window, fb, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db,
                         ref, amin):
    """The fused chain reading int16 PCM (value = sample * 2^-15) or 8-bit mu-law codes stored as uint8 / int64
    (value = the reference's 256-entry decode table) straight from the stored samples, converted in registers inside
    the frame load.  Returns None when the single-kernel route does not cover the configuration — the caller then
    converts first and takes the float32 path."""
    g = geometry(samples, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    if not ((g.pow2_kernel or g.mixed_radix) and g.onesided and g.n_fft in (256, 400, 512, 1024, 2048) and fb.dim() == 2
            and fb.shape[0] == g.n_bins
            and fb.is_contiguous() and power in (1.0, 2.0) and MEL_PATH != 'mfma'):
        return None
    pack = _melbank_pack(fb, g.n_fft)
    if pack is None:
        return None
    wpack, desc, info = pack
    fmt = _SAMPLE_FORMATS[samples.dtype]
    lut = _mulaw_tables(samples.device)[4] if fmt != _native.SAMPLES_I16 else None
    src = _rows_of(samples, g)
    out = torch.empty(g.lead + (g.n_frames, fb.shape[1]), dtype=torch.float32, device=samples.device)
    with _native.on_device(samples.device):
        rc = _native.lib().tac_melspec_sparse_coded_f32(
            _native.ptr(src), fmt, None if lut is None else _native.ptr(lut), _native.ptr(window), g.desc, float(power),
            _native.ptr(wpack), _native.ptr(desc), ctypes.cast(info, ctypes.c_void_p), fb.shape[1], 1 if db else 0,
            float(ref), float(amin), _native.ptr(out), _native.stream_ptr(samples.device))
    if rc == _native.TAC_E_UNSUPPORTED:
        return None
    _native.check(rc, 'tac_melspec_sparse_coded_f32')
    _count('tac_melspec_sparse_coded_f32')
    return out.transpose(-2, -1)
