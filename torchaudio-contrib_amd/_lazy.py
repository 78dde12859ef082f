"""Deferred evaluation so that a plain ``nn.Sequential`` of the reference's layers runs as ONE kernel.

The reference composes Melspectrogram as ``Sequential(STFT, ComplexNorm(2), ApplyFilterbank)`` and users append
``AmplitudeToDb()`` themselves (reference ``tests/test_layers.py:69`` unpacks the factory result with ``*``), so the
fusion boundary cannot be a container we own.  Instead ``STFT.forward`` returns a ``DeferredSpectral`` — a
``torch.Tensor`` wrapper subclass with the right shape / strides / dtype / device but no storage — and
``ComplexNorm`` / ``ApplyFilterbank`` extend the recipe when handed one.  ``AmplitudeToDb`` is terminal (nothing can
fuse behind it): it launches the single fused ``tac_amd::melspectrogram`` op at once and returns an ordinary
tensor.  A recipe that is still pending when it reaches anything else (any torch op, ``.cpu()``, printing,
``realize()``) is materialised through ``__torch_dispatch__``.

Safety of the deferral (the reference is eager; these make the difference unobservable or loud):
  * the recipe records the version counters and data pointers of the waveform, window and filterbank at
    ``forward`` time; materialising after any of them was modified in place raises instead of returning features
    of the wrong batch;
  * the kernel is enqueued on the stream that was current at ``forward`` time; if another stream is current when
    the value is needed, that stream is made to wait for it;
  * CPU tensors, float64 and ``torch.compile`` tracing are never deferred — the layers call the ops eagerly there;
  * a waveform that requires grad IS deferred (so that the reference idiom trains through the fused kernel and its
    fused backward): the recipe is then materialised by the registered ``tac_amd::*`` op, whose autograd entry links the
    result to the waveform.  ``AmplitudeToDb`` / ``realize()`` do that at module level; any other use of such a pending
    result is caught by ``__torch_function__`` — i.e. ABOVE autograd — and replaced by the materialised tensor, so the
    graph is the one eager evaluation would have built.  (``__torch_dispatch__`` runs below autograd: a value
    materialised there would be cut off from the waveform, so reaching it with a gradient-carrying recipe raises.)
"""
import sys

import torch
from torch.utils._pytree import tree_map

_enabled = True
_plans = {}             # (window id, filterbank id, stft args, power, dB) -> _hip.MelPlan of the fused chain, or a negative
                        # entry (_NO_PLAN, window, filterbank, layout) for a call no plan covers; mutated under _hip._lock
_NO_PLAN = object()


def _hip_lock():
    from . import _hip
    return _hip._lock


def set_lazy_fusion(flag):
    """Enable/disable deferred fusion of layer chains (on by default).  When off every layer launches
    its own kernel eagerly, exactly mirroring the reference's op-by-op evaluation."""
    global _enabled
    _enabled = bool(flag)


def lazy_fusion_enabled():
    return _enabled


_DEFERRABLE = (torch.float32, torch.float16, torch.bfloat16)
_DEFERRABLE_WAVE = _DEFERRABLE + (torch.int16,)                  # int16 = PCM, converted inside the frame load
_CODES = (torch.uint8, torch.int64)                              # mu-law codes a deferred MuLawDecoding can carry


def can_defer(wave, window):
    """Deferral applies to plain tensors on a HIP device that take the gfx950 kernels and carry no autograd state."""
    if isinstance(wave, DeferredWave):
        return wave.pending() and window.device == wave.device and window.dtype in _DEFERRABLE \
            and not (torch.is_grad_enabled() and window.requires_grad)
    if not _enabled or type(wave) is not torch.Tensor or not wave.is_cuda or wave.dtype not in _DEFERRABLE_WAVE:
        return False
    if torch.compiler.is_compiling() or torch._C._len_torch_dispatch_stack():
        return False
    if torch.is_grad_enabled() and window.requires_grad:
        return False
    if torch.is_grad_enabled() and wave.requires_grad and not wave.dtype.is_floating_point:
        return False
    return window.device == wave.device and window.dtype in _DEFERRABLE


def can_defer_codes(codes):
    """mu-law codes whose decoding may wait for the STFT behind it (plain integer tensors on a HIP device)."""
    return _enabled and type(codes) is torch.Tensor and codes.is_cuda and codes.dtype in _CODES \
        and not torch.compiler.is_compiling() and not torch._C._len_torch_dispatch_stack()


class DeferredWave(torch.Tensor):
    """Result of ``MuLawDecoding`` that has not been decoded yet: when an ``STFT`` layer takes it, the codes are decoded
    inside that kernel's frame load (through the reference's 256-entry table) and the waveform never exists in memory;
    anything else materialises it with the ordinary decode kernel.  Same safety rules as ``DeferredSpectral``."""

    @staticmethod
    def __new__(cls, codes, n_quantize):
        r = torch.Tensor._make_wrapper_subclass(cls, codes.shape, strides=codes.stride(), dtype=torch.float32,
                                                device=codes.device, requires_grad=False)
        r._codes = codes
        r._nq = int(n_quantize)
        r._stamp = (codes._version, codes.data_ptr())
        r._value = None
        return r

    def pending(self):
        return self._value is None

    def realize(self):
        if self._value is None:
            if (self._codes._version, self._codes.data_ptr()) != self._stamp:
                raise RuntimeError('torchaudio_contrib_amd: the mu-law codes handed to MuLawDecoding were modified in '
                                   'place before the deferred waveform was used')
            from ._ops import call
            self._value = call('mu_law_decoding', self._codes, self._nq, torch.float32)
            self._codes = None
        return self._value

    def __repr__(self):
        return 'DeferredWave(shape=%s, pending=%s)' % (tuple(self.shape), self.pending())

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def unwrap(a):
            return a.realize() if isinstance(a, (DeferredWave, DeferredSpectral)) else a
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))


class _Source(object):
    """The STFT call a recipe starts from, plus what is needed to detect that its inputs changed meanwhile."""
    __slots__ = ('wave', 'window', 'args', 'stream', 'stamps', 'lead', 'n_frames', 'n_bins', 'decode', 'grad', 'spec_layout')

    def __init__(self, wave, window, args):
        self.spec_layout = None                 # (shape, strides) of the |X|^p stage when STFT.forward supplied a template
        self.decode = None                      # n_quantize when `wave` holds mu-law codes (from a DeferredWave)
        # gradient flows through this recipe: it must be materialised above autograd (see the module docstring)
        self.grad = bool(torch.is_grad_enabled() and isinstance(wave, torch.Tensor) and wave.requires_grad)
        if isinstance(wave, DeferredWave):
            self.decode, wave = wave._nq, wave._codes
        self.wave, self.window, self.args = wave, window, args
        self.stream = torch._C._cuda_getCurrentRawStream(wave.device.index)      # hipStream_t of the forward call
        self.stamps = [(wave, wave._version, wave.data_ptr(), 'waveform'),
                       (window, window._version, window.data_ptr(), 'window')]

    def watch(self, tensor, what):
        self.stamps.append((tensor, tensor._version, tensor.data_ptr(), what))

    def check_unchanged(self):
        for tensor, version, address, what in self.stamps:
            if tensor._version != version or tensor.data_ptr() != address:
                raise RuntimeError(
                    'torchaudio_contrib_amd: the %s handed to STFT was modified in place before its deferred '
                    'spectrogram was used (the fused kernel had not been launched yet).  Use the result — or call '
                    'torchaudio_contrib_amd.realize() on it — before overwriting the input, finish the chain with '
                    'AmplitudeToDb (which launches immediately), or disable deferral with set_lazy_fusion(False).'
                    % what)


class DeferredSpectral(torch.Tensor):
    """Result of an STFT-rooted layer chain that has not been launched yet."""

    @staticmethod
    def __new__(cls, src, stage, shape, strides, power=None, filterbank=None):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, strides=strides, dtype=torch.float32,
                                                device=src.wave.device, requires_grad=src.grad)
        r._src = src
        r._tracks_grad = src.grad
        r._stage = stage            # 'stft' | 'spec' | 'mel'
        r._power = power
        r._fb = filterbank
        r._value = None
        return r

    # -- chain construction -----------------------------------------------------
    @classmethod
    def from_stft(cls, wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
        src = _Source(wave, window, (n_fft, hop, win_length, center, pad_mode, normalized, onesided))
        wave = src.wave
        src.lead = tuple(wave.shape[:-1])
        src.n_frames = 1 + (wave.shape[-1] + (2 * (n_fft // 2) if center else 0) - n_fft) // hop
        src.n_bins = n_fft // 2 + 1 if onesided else n_fft
        shape = src.lead + (src.n_bins, src.n_frames, 2)
        return cls(src, 'stft', shape, _transposed_strides(src.lead, (src.n_frames, src.n_bins, 2), -3, -2))

    @staticmethod
    def template(wave_shape, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
        """What ``from_stft`` derives from the input's SHAPE and the STFT arguments (``STFT.forward`` remembers it per shape)."""
        lead = tuple(wave_shape[:-1])
        n_frames = 1 + (wave_shape[-1] + (2 * (n_fft // 2) if center else 0) - n_fft) // hop
        n_bins = n_fft // 2 + 1 if onesided else n_fft
        return ((n_fft, hop, win_length, center, pad_mode, normalized, onesided), lead, n_frames, n_bins,
                lead + (n_bins, n_frames, 2), _transposed_strides(lead, (n_frames, n_bins, 2), -3, -2),
                lead + (n_bins, n_frames), _transposed_strides(lead, (n_frames, n_bins), -2, -1))

    @classmethod
    def from_template(cls, wave, window, tmpl):
        src = _Source(wave, window, tmpl[0])
        src.lead, src.n_frames, src.n_bins = tmpl[1], tmpl[2], tmpl[3]
        src.spec_layout = (tmpl[6], tmpl[7])
        return cls(src, 'stft', tmpl[4], tmpl[5])

    def pending(self):
        return self._value is None

    def with_norm(self, power):
        s = self._src
        lay = s.spec_layout
        if lay is not None:
            return DeferredSpectral(s, 'spec', lay[0], lay[1], power=power)
        return DeferredSpectral(s, 'spec', s.lead + (s.n_bins, s.n_frames),
                                _transposed_strides(s.lead, (s.n_frames, s.n_bins), -2, -1), power=power)

    def with_filterbank(self, fb):
        s = self._src
        s.watch(fb, 'filterbank')
        if torch.is_grad_enabled() and fb.requires_grad:
            s.grad = True
        return DeferredSpectral(s, 'mel', s.lead + (fb.shape[1], s.n_frames),
                                _transposed_strides(s.lead, (s.n_frames, fb.shape[1]), -2, -1),
                                power=self._power, filterbank=fb)

    # -- materialisation --------------------------------------------------------
    def _launch(self, db):
        from ._ops import call
        s = self._src
        if db is not None and self._tracks_grad and torch.is_grad_enabled() and self._stage in ('spec', 'mel'):
            # training: the dB gradient needs the linear values.  Fused into one op they would have to be recomputed in
            # backward (a second launch of the fused kernel, 0.15 ms at cfg-2); as two ops the dB op saves its input
            # (one tiny extra kernel forward, n_mels x frames floats kept) and backward recomputes nothing.
            return call('amplitude_to_db', self._launch(None), float(db[0]), float(db[1]))
        ref, amin = db if db is not None else (1.0, 1e-7)
        wave = s.wave
        if s.decode is not None:
            if self._stage == 'mel':            # codes decoded inside the fused kernel's frame load
                return call('melspectrogram_mulaw', wave, s.window, self._fb, s.decode, *s.args, float(self._power),
                            db is not None, float(ref), float(amin))
            wave = call('mu_law_decoding', wave, s.decode, torch.float32)
        if self._stage == 'stft':
            return call('stft', wave, s.window, *s.args)
        if self._stage == 'spec':
            return call('spectrogram', wave, s.window, *s.args, float(self._power), db is not None, float(ref),
                        float(amin))
        return call('melspectrogram', wave, s.window, self._fb, *s.args, float(self._power), db is not None,
                    float(ref), float(amin))

    def _launch_planned(self, db):
        """The fused mel chain through a cached bound-argument launcher (``_hip.MelPlan``: geometry, route, packed bank and
        argument conversions done once per window / filterbank / layout); anything the plan does not cover, and the first
        call, take ``_launch``."""
        s = self._src
        fb = self._fb
        key = (id(s.window), id(fb), s.args, self._power, db)
        plan = _plans.get(key)
        if type(plan) is tuple:                 # negative entry
            if plan[1] is s.window and plan[2] is fb and plan[3] == (s.wave.shape, s.wave.stride(), s.wave.dtype):
                return self._launch(db)         # ... for exactly this call: the general path, no new plan
        elif plan is not None and plan.window is s.window and plan.fb is fb:
            v = plan.run(s.wave)                # layout / device / stamps checked, output allocated, kernel launched
            if v is not None:
                return v
            if not plan.matches(s.wave):        # a different layout or new contents: the general path, then a new plan below
                return self._replan(key, db)
            from . import _native
            if plan.last_rc in (_native.TAC_E_UNSUPPORTED, _native.TAC_E_INVALID):
                with _hip_lock():               # refused for what the call IS: remember that instead of rebuilding the plan on every call
                    _plans[key] = (_NO_PLAN, s.window, fb, (s.wave.shape, s.wave.stride(), s.wave.dtype))
            # (a HIP runtime error — TAC_E_LAUNCH — may be transient: the plan stays, the general path reports this call's error)
            return self._launch(db)
        return self._replan(key, db)

    def _replan(self, key, db):
        s = self._src
        fb = self._fb
        v = self._launch(db)
        try:                                    # (a missing library / unsupported geometry was reported by _launch already)
            from . import _hip
            ref, amin = db if db is not None else (1.0, 1e-7)
            plan = _hip.mel_plan(s.wave, s.window, fb, *s.args, float(self._power), db is not None, float(ref), float(amin))
        except Exception:                       # noqa: BLE001 — the plan is an optimisation only
            plan = None
        with _hip_lock():                       # (invalidate() walks and prunes this dict under the same lock)
            if len(_plans) > 64:
                _plans.clear()
            if plan is not None:
                _plans[key] = plan
            else:
                _plans[key] = (_NO_PLAN, s.window, fb, (s.wave.shape, s.wave.stride(), s.wave.dtype))
        return v

    def realize(self, db=None):
        """Launch the recorded chain (with an ``amplitude_to_db(ref, amin)`` epilogue when ``db`` is given) and return
        an ordinary tensor; a chain without dB epilogue remembers its value."""
        if self._value is not None:
            if db is None:
                return self._value
            from ._ops import call
            return call('amplitude_to_db', self._value, float(db[0]), float(db[1]))
        s = self._src
        s.check_unchanged()
        device = s.wave.device
        if torch._C._cuda_getCurrentRawStream(device.index) == s.stream:
            v = self._launch_planned(db) if (self._stage == 'mel' and not self._tracks_grad and s.decode is None) \
                else self._launch(db)
        else:                                   # enqueue where forward() was called, then order the consumer behind it
            now = torch.cuda.current_stream(device)
            then = torch.cuda.ExternalStream(s.stream, device=device) if s.stream else torch.cuda.default_stream(device)
            with torch.cuda.stream(then):
                v = self._launch(db)
                done = torch.cuda.Event()
                done.record(then)
            now.wait_event(done)
            v.record_stream(now)
        if db is None:
            self._value = v
            self._src = None
            self._fb = None
        return v

    def __repr__(self):
        return 'DeferredSpectral(stage=%s, shape=%s, pending=%s)' % (self._stage, tuple(self.shape), self.pending())

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if not torch.is_grad_enabled() or _is_metadata_query(func) or not _carries_grad(args, kwargs):
            # what a subclass without __torch_function__ gets: the function runs as is (no re-wrapping of its results)
            # and pending recipes are materialised by __torch_dispatch__
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        # a gradient flows through the recipe: hand the caller's function the materialised tensor(s) here, above
        # autograd, so that it records the graph eager evaluation would have recorded
        def unwrap(a):
            return a.realize() if isinstance(a, DeferredSpectral) else a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def unwrap(a):
            if isinstance(a, DeferredSpectral):
                if a._tracks_grad and torch.is_grad_enabled():
                    raise RuntimeError(
                        'torchaudio_contrib_amd: a deferred spectrogram whose waveform requires grad reached %s below '
                        'autograd; materialise it first with torchaudio_contrib_amd.realize(x) (or disable deferral '
                        'with set_lazy_fusion(False))' % (func,))
                return a.realize()
            return a
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))


#: attribute getters and methods that only read what the wrapper itself carries (shape, strides, dtype, device, autograd
#: flags): they never need the value
_METADATA_PROPERTIES = frozenset((
    'shape', 'device', 'dtype', 'layout', 'ndim', 'requires_grad', 'grad', 'grad_fn', 'is_leaf', 'is_cuda', 'is_cpu',
    'is_meta', 'is_sparse', 'is_sparse_csr', 'is_quantized', 'is_mkldnn', 'is_nested', 'is_xpu', 'is_mps', 'is_xla',
    'is_vulkan', 'is_ort', 'is_ipu', 'is_mtia', 'is_maia', 'names', 'itemsize', 'nbytes', 'output_nr', '_version', '_base',
    'name', 'retains_grad', '_grad', '_grad_fn'))
_METADATA_METHODS = frozenset((
    'size', 'dim', 'ndimension', 'stride', 'numel', 'nelement', 'is_contiguous', 'storage_offset', 'element_size',
    'is_floating_point', 'is_complex', 'is_signed', 'get_device', 'is_pinned', 'is_inference', 'is_shared', '__len__',
    '__repr__', '__hash__', 'has_names', '_is_view', 'is_conj', 'is_neg', 'sym_size', 'sym_stride', 'sym_numel',
    'sym_storage_offset', 'dim_order'))


def _is_metadata_query(func):
    name = getattr(func, '__name__', '')
    if name == '__get__':
        return getattr(getattr(func, '__self__', None), '__name__', '') in _METADATA_PROPERTIES
    if name == '__set__':
        return True
    return name in _METADATA_METHODS


def _carries_grad(args, kwargs):
    found = []

    def look(a):
        if isinstance(a, DeferredSpectral) and a._tracks_grad:
            found.append(a)
        return a
    tree_map(look, args)
    if not found and kwargs:
        tree_map(look, kwargs)
    return bool(found)


_stride_cache = {}


def _transposed_strides(lead, tail, a, b):
    """Strides of ``empty(lead + tail).transpose(a, b)`` (memoised: a pipeline asks for the same few every call)."""
    key = (lead, tail, a, b)
    hit = _stride_cache.get(key)
    if hit is None:
        if len(_stride_cache) > 256:
            _stride_cache.clear()
        hit = _stride_cache[key] = _compute_transposed_strides(lead, tail, a, b)
    return hit


def _compute_transposed_strides(lead, tail, a, b):
    shape = tuple(lead) + tuple(tail)
    strides = [0] * len(shape)
    acc = 1
    for i in range(len(shape) - 1, -1, -1):
        strides[i] = acc
        acc *= max(shape[i], 1)
    n = len(shape)
    a, b = a % n, b % n
    strides[a], strides[b] = strides[b], strides[a]
    return tuple(strides)


_SEQ_FORWARD_CODE = torch.nn.Sequential.forward.__code__
_MODULE_PY = torch.nn.Module.__call__.__code__.co_filename           # torch/nn/modules/module.py: __call__ / _call_impl frames


_ends_chain_resolved = False


def ends_chain(module):
    """True when ``module`` — a deferring layer, asked from inside its ``forward`` — produces the END of a layer chain: it was
    called by an ``nn.Sequential`` as that container's LAST child, through any nesting of such containers, and the outermost of
    them by something that is not a container.  Nothing can fuse behind such a result, so the layer launches the chain and hands
    back an ordinary ``torch.Tensor`` — ``type(nn.Sequential(*Melspectrogram(...))(x)) is torch.Tensor``, as with the reference.
    A direct call of the layer, a container that goes on after it (``..., AmplitudeToDb()``), or a factory container (which
    realises by itself) keep the result deferred.  One walk over the caller frames: ~0.2 us per layer in the compiled binding
    (``chain_end`` of csrc/binding/tac_ext.cpp), ~1.5 us here (``sys._getframe``, ``f_locals``) without it."""
    global ends_chain, _ends_chain_resolved
    if not _ends_chain_resolved:                # first call: hand over to the compiled walk when the binding offers it
        _ends_chain_resolved = True
        try:
            from . import _native
            ext = _native.ext()
            if ext is not None and getattr(ext, 'chain_end_supported', False):
                ext.chain_end_init(_SEQ_FORWARD_CODE, _MODULE_PY)
                ends_chain = ext.ends_chain     # (callers read _lazy.ends_chain at call time)
                return ext.chain_end(module, 1) == 1
        except Exception:                       # noqa: BLE001 — the Python walk below answers
            pass
    f = sys._getframe(2)
    child = module
    while True:
        while f is not None and f.f_code.co_filename == _MODULE_PY:
            f = f.f_back
        if f is None or f.f_code is not _SEQ_FORWARD_CODE:
            return child is not module          # called by user code: the end of the chain iff we came out of a container
        cont = f.f_locals.get('self')
        mods = cont.__dict__.get('_modules') if cont is not None else None      # (plain dict reads: nn.Module.__getattr__ is slow)
        if not mods or getattr(type(cont), '_tac_realizes', False) or next(reversed(mods.values())) is not child:
            return False
        child = cont
        f = f.f_back


class PlannedChain(object):
    """``planned(model, example)``: the layer chain ``model`` bound to ONE launch for inputs laid out like ``example``.

    The reference's idiom pays five ``nn.Module`` calls and three deferred intermediates per forward (~16 us of host time on
    MI355X: ``bench.py`` ``stages.small_batch``); for callers who run many small batches this object makes the same call — layout
    and content-stamp checks, output allocation, current-stream lookup, one launch of the fused kernel — in one step (~5 us).
    Bound at construction: the STFT arguments, the window and filterbank TENSORS (their contents are re-checked on every call) and
    the dB parameters; after changing a module attribute build a new one.  Anything the fused kernel does not cover (another
    layout or device, a waveform that requires grad, a chain it does not recognise) goes through ``model`` itself."""

    def __init__(self, model, example):
        from . import layers as L, functional as F, _hip
        self.model = model
        self.plan = None
        kids = []

        def flat(m):
            if isinstance(m, torch.nn.Sequential):
                for c in m:
                    flat(c)
            else:
                kids.append(m)
        flat(model)
        db = None
        if kids and type(kids[-1]) is L.AmplitudeToDb:
            db = (float(kids[-1].ref), float(kids[-1].amin))
            kids = kids[:-1]
        if len(kids) == 3 and type(kids[0]) is L.STFT and type(kids[1]) is L.ComplexNorm and type(kids[2]) is L.ApplyFilterbank \
                and lazy_fusion_enabled() and type(example) is torch.Tensor and example.is_cuda:
            st, fb = kids[0], kids[2].filterbank
            n_fft, hop, win_length, window = F.resolve_stft_args(example, st.fft_length, st.hop_length, st.win_length, st.window)
            _hip.check_stft_args(example.shape, n_fft, hop, win_length, st.center, st.pad_mode)
            if window is st.window:
                ref, amin = db if db is not None else (1.0, 1e-7)
                self.plan = _hip.mel_plan(example, window, fb, n_fft, hop, win_length, bool(st.center), st.pad_mode,
                                          bool(st.normalized), bool(st.onesided), float(kids[1].power), db is not None, ref, amin)

    def fused(self):
        """whether calls take the bound one-launch path (else they are ``model(x)``)"""
        return self.plan is not None

    def __call__(self, x):
        p = self.plan
        if p is not None and not (x.requires_grad and torch.is_grad_enabled()):
            v = p.run(x)
            if v is not None:
                return v
        return realize(self.model(x))


def planned(model, example):
    """A callable equivalent to ``model`` with the whole chain bound to one launch (``PlannedChain``)."""
    return PlannedChain(model, example)


def realize(x):
    """Materialise a deferred layer-chain result (no-op for ordinary tensors).  The kernel is enqueued on the HIP
    stream that was current when the chain's STFT was called; like any HIP op it completes asynchronously."""
    if isinstance(x, (DeferredSpectral, DeferredWave)):
        return x.realize()
    return x
