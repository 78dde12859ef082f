"""``torch.library`` registration of the path: every functional of the reference's API is one PyTorch custom op in
the ``tac_amd`` namespace, so the dispatcher, ``torch.compile`` / FakeTensor tracing, autograd and the profiler
see them by name.

    op                      CUDA (= HIP on ROCm) kernel                 CPU kernel            Meta (fake)
    tac_amd::stft           gfx950 kernels through the C ABI (_hip.py)  stock torch ops       shapes + strides
    tac_amd::spectrogram        "      (STFT + |.|^p [+ dB] fused)      (_composite.py)
    tac_amd::melspectrogram     "      (the whole chain in ONE kernel)
    tac_amd::apply_filterbank, complex_norm, angle, magphase, phase_vocoder, amplitude_to_db, db_to_amplitude,
    tac_amd::mu_law_encoding, mu_law_decoding                           likewise

Routing on a HIP device: float32 (and float16/bfloat16, widened as ``torch.stft`` widens half input) always runs
the hand-written kernels and raises if ``libtac_amd.so`` is missing.  float64 runs the float64 kernels of the STFT chain
(``_hip64.py`` -> csrc/chain_f64.hip) and of the phase vocoder.  What those do not cover (float64 mu-law / HPSS) and fft
sizes outside the kernels' range are evaluated by torch's own GPU operators
(``_composite.py``) with a one-time ``CompositeRouteWarning``; ``set_strict(True)`` turns that route into an error
(the GPU parity tests run strict, so nothing they check can have come from anywhere but the HIP kernels).

Autograd: stft / spectrogram / melspectrogram / apply_filterbank / complex_norm / amplitude_to_db have hand-written
gradient kernels for the signal path (csrc/backward.hip: inverse real FFT per frame + gather overlap-add, the
filterbank GEMM with the transposed bank, elementwise adjoints; the fused ops recompute their spectrum instead of
saving it).  Gradients w.r.t. the window / filterbank, float64, CPU tensors and the remaining ops differentiate by
re-evaluating the op with torch operators under ``enable_grad``.
"""
import warnings

import torch
from torch.library import Library

from . import _composite as C
from . import _hip as H
from . import _hip64 as H64

NS = 'tac_amd'
_lib = Library(NS, 'DEF')

_strict = False
_warned = set()
#: how many op calls took the stock-torch route, per (op, reason) — introspection for tests / users
composite_calls = {}


class CompositeRouteWarning(UserWarning):
    """An op was evaluated by stock torch operators instead of the gfx950 kernels (float64 input, unsupported size)."""


_strict_backward = None        # None: follow _strict


def set_strict(flag, backward=None):
    """With strict on, a tensor on a HIP device is never handed to the stock-torch route: the call raises instead.

    That covers backward passes as well: the gradient of an op without a gradient kernel (``phase_vocoder``, ``angle``,
    ``magphase``, ``hpss``, ``db_to_amplitude``), every float64 gradient and every double backward re-evaluate the op with
    stock torch operators, so under ``set_strict(True)`` they RAISE (since round 3; earlier rounds let a backward pass through
    when its forward had run on the kernels).  ``backward=False`` keeps strictness for forward calls only — those backward
    passes then run, announced by ``CompositeRouteWarning`` and counted in ``composite_calls``; ``backward=None`` (default)
    follows ``flag``."""
    global _strict, _strict_backward
    _strict = bool(flag)
    _strict_backward = None if backward is None else bool(backward)


def strict():
    return _strict


def strict_backward():
    return _strict if _strict_backward is None else _strict_backward


def _composite_route(op, reason):
    composite_calls[(op, reason)] = composite_calls.get((op, reason), 0) + 1
    if strict_backward() if reason.startswith('backward: ') else _strict:
        raise RuntimeError('tac_amd::%s: %s is outside the gfx950 kernels and strict mode forbids the stock-torch '
                           'route' % (op, reason))
    if (op, reason) not in _warned:
        _warned.add((op, reason))
        warnings.warn('tac_amd::%s: %s — evaluated by stock torch operators on the device, not by the gfx950 '
                      'kernels' % (op, reason), CompositeRouteWarning, stacklevel=3)


_WIDEN = (torch.float16, torch.bfloat16)


def _f32(t):
    return t.float() if t.dtype in _WIDEN else t


def _hip_dtype(*tensors):
    """None when the gfx950 kernels take these tensors (float32, or half widened), else the reason they do not."""
    for t in tensors:
        if t.dtype == torch.float32 or t.dtype in _WIDEN:
            continue
        return 'dtype %s' % str(t.dtype).replace('torch.', '')
    return None


def _same_device(op, *tensors):
    dev = tensors[0].device
    for t in tensors[1:]:
        if t.device != dev:
            raise RuntimeError('tac_amd::%s: tensors are on different devices (%s and %s)' % (op, dev, t.device))


# ============================================================================= fake (meta) helpers
def _swapped(lead, tail, dtype, device, a, b):
    """empty(lead + tail).transpose(a, b): the frame-major physical layout every STFT-family kernel writes, returned
    as the logical (.., freq, time[, 2]) view — the same strides the reference's own torch ops produce."""
    return torch.empty(tuple(lead) + tuple(tail), dtype=dtype, device=device).transpose(a, b)


def _out_dtype(t):
    return torch.float32 if (t.dtype in _WIDEN or t.dtype == torch.int16) else t.dtype


def _n_frames(length, n_fft, hop, center):
    return 1 + (length + (2 * (n_fft // 2) if center else 0) - n_fft) // hop


# ============================================================================= autograd
def _plain_hip_f32(*tensors):
    return all(t is not None and type(t) is torch.Tensor and t.is_cuda and t.dtype == torch.float32 for t in tensors)


def _register_autograd(op, fn, n_tensors, hip_backward=None):
    """Backward of ``tac_amd::<op>``.  ``hip_backward(tensors, rest, needs, grads)`` — the hand-written gradient
    kernels (csrc/backward.hip) — is used when it applies (float32 on a HIP device); it returns None otherwise and
    the op is then differentiated by re-evaluating it with differentiable torch operators (``_composite``) on the
    saved inputs.  That second route is the only one on CPU tensors; on a HIP device it is announced like every other
    use of stock torch operators there (``CompositeRouteWarning``, an error under ``set_strict(True)``, counted in
    ``composite_calls``).

    Double backward (``create_graph=True``; the reference, being stock torch operators, is twice differentiable): the
    gradient kernels produce values without a graph, so a backward pass that is itself recorded re-evaluates the op
    with torch operators on the SAVED tensors — still attached to the caller's graph — and differentiates that with
    ``create_graph=True``."""

    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(*inputs[:n_tensors])
        ctx.rest = tuple(inputs[n_tensors:])

    def backward(ctx, *grads):
        needs = ctx.needs_input_grad[:n_tensors]
        saved = ctx.saved_tensors
        second_order = torch.is_grad_enabled()
        why = 'double backward (create_graph=True)' if second_order else None
        if why is None and hip_backward is not None:
            if _plain_hip_f32(*saved) and all(g is None or _plain_hip_f32(g) for g in grads):
                with torch.no_grad():
                    res = hip_backward(saved, ctx.rest, needs, grads)
                if res is not None:
                    return tuple(res) + (None,) * len(ctx.rest)
                why = 'this gradient has no gfx950 kernel'
            else:
                why = _hip_dtype(*[t for t in tuple(saved) + tuple(grads) if t is not None]) or 'tensor subclass'
        elif why is None:
            why = 'the op has no gradient kernel'
        if any(t.is_cuda for t in saved):
            _composite_route(op, 'backward: ' + why)
        with torch.enable_grad():
            if second_order:        # keep the graph: the saved tensors are the caller's own
                ins = list(saved)
                wanted = [t for t, need in zip(ins, needs) if need and t.requires_grad]
            else:
                ins = [t.detach().requires_grad_(True) if (need and t.is_floating_point()) else t.detach()
                       for t, need in zip(saved, needs)]
                wanted = [t for t in ins if t.requires_grad]
            outs = fn(*ins, *ctx.rest)
            outs = outs if isinstance(outs, tuple) else (outs,)
            pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
            got = torch.autograd.grad([o for o, _ in pairs], wanted, [g for _, g in pairs], allow_unused=True,
                                      create_graph=second_order) if pairs and wanted else ()
        by_id = {id(t): g for t, g in zip(wanted, got)}
        result = [by_id.get(id(t)) for t in ins]
        return tuple(result) + (None,) * len(ctx.rest)

    torch.library.register_autograd('%s::%s' % (NS, op), backward, setup_context=setup_context, lib=_lib)


def _signal_path_only(needs):
    """Gradient wanted for the first tensor (waveform / spectrogram) and for none of the constant tables."""
    return bool(needs[0]) and not any(needs[1:])


def _fast_backward(needs, n_fft, onesided):
    """the fused backward kernels: gradient of the waveform only, one-sided power-of-two fft_length"""
    return _signal_path_only(needs) and H.stft_backward_supported(n_fft, onesided)


def _stft_hip_backward(saved, rest, needs, grads):
    wave, window = saved
    n_fft, hop, win_length, center, pad_mode, normalized, onesided = rest[:7]
    if grads[0] is None or not H.hip_covers_backward(n_fft):
        return None
    window = window.contiguous()
    if _fast_backward(needs, n_fft, onesided):
        return [H.stft_backward(grads[0], wave, window, n_fft, hop, win_length, center, pad_mode, normalized), None]
    return list(H.stft_backward_general(grads[0], wave, window, n_fft, hop, win_length, center, pad_mode, normalized,
                                        onesided, bool(needs[0]), bool(needs[1])))


def _spectrogram_general_backward(wave, window, geo, power, g, need_wave, need_window):
    """spectrum recomputed by the forward kernel, the norm's adjoint, then the general stft adjoint"""
    n_fft, hop, win_length, center, pad_mode, normalized, onesided = geo
    z = H.stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    gz = H.complex_norm_backward(z, g, power)
    return H.stft_backward_general(gz, wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided,
                                   need_wave, need_window)


def _spectrogram_hip_backward(saved, rest, needs, grads):
    wave, window = saved
    n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref, amin = rest
    if grads[0] is None or not H.hip_covers_backward(n_fft):
        return None
    window = window.contiguous()
    g = grads[0]
    if db:      # the |z|^power values the dB gradient needs: one fused forward launch (nothing was saved)
        g = H.amplitude_to_db_backward(H.spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized,
                                                     onesided, power, False, 1.0, 1e-7), g, amin)
    if not _fast_backward(needs, n_fft, onesided):
        return list(_spectrogram_general_backward(wave, window, rest[:7], power, g, bool(needs[0]), bool(needs[1])))
    # the backward kernel transforms the frames again itself and folds the norm's adjoint into the inverse FFT's load:
    # neither the spectrum nor a gradient spectrum exists in memory (fft_length 4096: the spectrum is recomputed first)
    z = None if H.backward_recomputes_spectrum(n_fft) else \
        H.stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    return [H.stft_backward(z, wave, window, n_fft, hop, win_length, center, pad_mode, normalized, grad_norm=g, power=power),
            None]


def _melspectrogram_hip_backward(saved, rest, needs, grads):
    wave, window, bank = saved
    n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref, amin = rest
    if grads[0] is None or not H.hip_covers_backward(n_fft):
        return None
    window = window.contiguous()
    g = grads[0]
    if db:      # the mel values the dB gradient needs come from the fused forward kernel (one launch)
        mel = H.melspectrogram(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power,
                               False, 1.0, 1e-7)
        g = H.amplitude_to_db_backward(mel, g, amin)
    grad_bank = None
    if needs[2]:
        grad_bank = H.filterbank_grad(H.spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized,
                                                    onesided, power, False, 1.0, 1e-7), g)
    if not (needs[0] or needs[1]):
        return [None, None, grad_bank]
    if _fast_backward(needs[:2], n_fft, onesided):
        # fft_length 2048: filterbank adjoint, frame re-transform, norm adjoint, inverse FFT and overlap-add in one kernel
        gw = H.melspectrogram_backward_fused(g, wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, power)
        if gw is not None:
            return [gw, None, grad_bank]
    gp = H.apply_filterbank_backward(g, bank)
    if not _fast_backward(needs[:2], n_fft, onesided):
        gw, gwin = _spectrogram_general_backward(wave, window, rest[:7], power, gp, bool(needs[0]), bool(needs[1]))
        return [gw, gwin, grad_bank]
    z = None if H.backward_recomputes_spectrum(n_fft) else \
        H.stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    return [H.stft_backward(z, wave, window, n_fft, hop, win_length, center, pad_mode, normalized, grad_norm=gp, power=power),
            None, grad_bank]


def _apply_filterbank_hip_backward(saved, rest, needs, grads):
    if grads[0] is None:
        return None
    spec, bank = saved
    return [H.apply_filterbank_backward(grads[0], bank) if needs[0] else None,
            H.filterbank_grad(spec, grads[0]) if needs[1] else None]


def _complex_norm_hip_backward(saved, rest, needs, grads):
    if grads[0] is None or saved[0].shape[-1] != 2:
        return None
    return [H.complex_norm_backward(saved[0], grads[0], rest[0])]


def _amplitude_to_db_hip_backward(saved, rest, needs, grads):
    if grads[0] is None:
        return None
    return [H.amplitude_to_db_backward(saved[0], grads[0], rest[1])]


def _angle_hip_backward(saved, rest, needs, grads):
    if grads[0] is None or saved[0].shape[-1] != 2:
        return None
    return [H.magphase_backward(saved[0], None, grads[0], 1.0)]


def _magphase_hip_backward(saved, rest, needs, grads):
    if (grads[0] is None and grads[1] is None) or saved[0].shape[-1] != 2:
        return None
    return [H.magphase_backward(saved[0], grads[0], grads[1], rest[0])]


def _phase_vocoder_hip_backward(saved, rest, needs, grads):
    spec, phase_advance = saved
    if grads[0] is None or spec.shape[-1] != 2 or spec.dim() < 3:
        return None
    return [H.phase_vocoder_backward(spec, rest[0], grads[0]) if needs[0] else None,
            torch.zeros_like(phase_advance) if needs[1] else None]       # (the wrap and the advance cancel: zero, as autograd finds)


def _hpss_hip_backward(saved, rest, needs, grads):
    kernel_f, kernel_t, power, hard = rest
    if all(g is None for g in grads) or not H.hpss_supported(kernel_f, kernel_t):
        return None
    gs = list(grads) if len(grads) == 4 else [None, None] + list(grads)      # (hpss_masks: the two masks only)
    if hard:
        gs[2] = gs[3] = None                                                  # boolean masks carry no gradient
    return [H.hpss_backward(saved[0], kernel_f, kernel_t, power, hard, gs)]


def _db_to_amplitude_hip_backward(saved, rest, needs, grads):
    if grads[0] is None or not rest[0] > 0.0:
        return None
    return [H.db_to_amplitude_backward(saved[0], grads[0], rest[0])]


_HIP_BACKWARD = {'stft': _stft_hip_backward, 'spectrogram': _spectrogram_hip_backward,
                 'melspectrogram': _melspectrogram_hip_backward, 'apply_filterbank': _apply_filterbank_hip_backward,
                 'complex_norm': _complex_norm_hip_backward, 'amplitude_to_db': _amplitude_to_db_hip_backward,
                 'angle': _angle_hip_backward, 'magphase': _magphase_hip_backward, 'db_to_amplitude': _db_to_amplitude_hip_backward,
                 'phase_vocoder': _phase_vocoder_hip_backward, 'hpss': _hpss_hip_backward, 'hpss_masks': _hpss_hip_backward}


#: the CUDA-key kernels by op name: `call` below invokes them directly when the dispatcher has nothing to add
cuda_kernels = {}


def call(op, *args):
    """Invoke ``tac_amd::<op>``.  In plain eager mode on HIP tensors without autograd state, tracing, dispatch modes or
    an active profiler, the dispatcher would do nothing but box the arguments twice (~11 us per call through the
    Python-kernel path) before reaching the CUDA-key kernel — so that kernel is called directly; every other situation
    (CPU tensors, autograd, torch.compile / FakeTensor, functorch, profiling) goes through the registered op."""
    if torch.compiler.is_compiling():
        return getattr(ops, op)(*args)
    grad = torch.is_grad_enabled()
    direct = not (torch._C._len_torch_dispatch_stack() or torch.autograd._profiler_enabled()
                  or torch._C._functorch.peek_interpreter_stack() is not None)
    if direct:
        for a in args:
            if isinstance(a, torch.Tensor):
                if type(a) is not torch.Tensor or not a.is_cuda or (grad and a.requires_grad):
                    direct = False
                    break
    if direct:
        return cuda_kernels[op](*args)
    return getattr(ops, op)(*args)


def _register(op, schema, cuda, cpu, fake, n_tensors, differentiable=True):
    cuda_kernels[op] = cuda
    _lib.define(op + schema)
    _lib.impl(op, cuda, 'CUDA')
    _lib.impl(op, cpu, 'CPU')
    torch.library.register_fake('%s::%s' % (NS, op), fake, lib=_lib)
    if differentiable:
        _register_autograd(op, cpu, n_tensors, _HIP_BACKWARD.get(op))


# ============================================================================= stft
_STFT_ARGS = 'int n_fft, int hop, int win_length, bool center, str pad_mode, bool normalized, bool onesided'


def _pcm(wave):
    """int16 waveforms are PCM: sample * 2^-15 (converted by a HIP kernel where the frame load cannot do it)."""
    return H.pcm16_to_f32(wave) if wave.dtype == torch.int16 else _f32(wave)


def _stft_route(op, wave, n_fft, *others):
    reason = _hip_dtype(*others) if wave.dtype == torch.int16 else _hip_dtype(wave, *others)
    if reason is None and not H.hip_covers_n_fft(n_fft):
        reason = 'fft_length %d' % n_fft
    if reason is not None:
        _composite_route(op, reason)
    return reason


def _stft_cuda(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
    _same_device('stft', wave, window)
    if H64.all_f64(wave, window) and H64.covers(n_fft):
        return H64.stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    if _stft_route('stft', wave, n_fft, window) is not None:
        return C.stft(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided)
    return H.stft(_pcm(wave), _f32(window).contiguous(), n_fft, hop, win_length, center, pad_mode, normalized,
                  onesided)


def _stft_fake(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided):
    n_bins = n_fft // 2 + 1 if onesided else n_fft
    frames = _n_frames(wave.shape[-1], n_fft, hop, center)
    return _swapped(wave.shape[:-1], (frames, n_bins, 2), _out_dtype(wave), wave.device, -3, -2)


_register('stft', '(Tensor wave, Tensor window, %s) -> Tensor' % _STFT_ARGS, _stft_cuda, C.stft, _stft_fake, 2)


# ============================================================================= spectrogram (stft + |.|^p [+ dB])
def _spectrogram_cuda(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref,
                      amin):
    _same_device('spectrogram', wave, window)
    if H64.all_f64(wave, window) and H64.covers(n_fft):
        return H64.spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref,
                               amin)
    if _stft_route('spectrogram', wave, n_fft, window) is not None:
        return C.spectrogram(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db,
                             ref, amin)
    return H.spectrogram(_pcm(wave), _f32(window).contiguous(), n_fft, hop, win_length, center, pad_mode, normalized,
                         onesided, power, db, ref, amin)


def _spectrogram_fake(wave, window, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db, ref,
                      amin):
    n_bins = n_fft // 2 + 1 if onesided else n_fft
    frames = _n_frames(wave.shape[-1], n_fft, hop, center)
    return _swapped(wave.shape[:-1], (frames, n_bins), _out_dtype(wave), wave.device, -2, -1)


_register('spectrogram', '(Tensor wave, Tensor window, %s, float power, bool db, float ref, float amin) -> Tensor'
          % _STFT_ARGS, _spectrogram_cuda, C.spectrogram, _spectrogram_fake, 2)


# ============================================================================= melspectrogram (the fused chain)
def _melspectrogram_cuda(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power,
                         db, ref, amin):
    _same_device('melspectrogram', wave, window, bank)
    if H64.all_f64(wave, window, bank) and H64.covers(n_fft) and bank.dim() == 2 and bank.shape[0] == (
            n_fft // 2 + 1 if onesided else n_fft):
        return H64.melspectrogram(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power,
                                  db, ref, amin)
    if _stft_route('melspectrogram', wave, n_fft, window, bank) is not None:
        return C.melspectrogram(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided,
                                power, db, ref, amin)
    window, bank = _f32(window).contiguous(), _f32(bank)
    if wave.dtype == torch.int16:                                        # PCM converted inside the frame load when one kernel covers it
        fused = H.melspectrogram_coded(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided,
                                       power, db, ref, amin)
        if fused is not None:
            return fused
    return H.melspectrogram(_pcm(wave), window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided,
                            power, db, ref, amin)


def _melspectrogram_fake(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power,
                         db, ref, amin):
    frames = _n_frames(wave.shape[-1], n_fft, hop, center)
    return _swapped(wave.shape[:-1], (frames, bank.shape[1]), _out_dtype(wave), wave.device, -2, -1)


cuda_kernels['melspectrogram'] = _melspectrogram_cuda
_lib.define('melspectrogram(Tensor wave, Tensor window, Tensor filterbank, %s, float power, bool db, float ref, '
            'float amin) -> Tensor' % _STFT_ARGS)
_lib.impl('melspectrogram', _melspectrogram_cuda, 'CUDA')
_lib.impl('melspectrogram', C.melspectrogram, 'CPU')
torch.library.register_fake(NS + '::melspectrogram', _melspectrogram_fake, lib=_lib)
_register_autograd('melspectrogram', C.melspectrogram, 3, _melspectrogram_hip_backward)


# ============================================================================= mu-law codes -> melspectrogram
def _melspectrogram_mulaw_cuda(codes, window, bank, n_quantize, n_fft, hop, win_length, center, pad_mode, normalized,
                               onesided, power, db, ref, amin):
    _same_device('melspectrogram_mulaw', codes, window, bank)
    reason = _hip_dtype(window, bank)
    if reason is None and not H.hip_covers_n_fft(n_fft):
        reason = 'fft_length %d' % n_fft
    if reason is not None:
        _composite_route('melspectrogram_mulaw', reason)
        return C.melspectrogram_mulaw(codes, window, bank, n_quantize, n_fft, hop, win_length, center, pad_mode, normalized,
                                      onesided, power, db, ref, amin)
    window, bank = _f32(window).contiguous(), _f32(bank)
    if n_quantize == 256 and codes.dtype in (torch.uint8, torch.int64):   # decoded inside the frame load
        fused = H.melspectrogram_coded(codes, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided,
                                       power, db, ref, amin)
        if fused is not None:
            return fused
    wave = H.mu_law_decoding_int(codes, n_quantize) if not codes.is_floating_point() else H.mu_law_decoding_float(_f32(codes), n_quantize)
    return H.melspectrogram(wave, window, bank, n_fft, hop, win_length, center, pad_mode, normalized, onesided, power, db,
                            ref, amin)


def _melspectrogram_mulaw_fake(codes, window, bank, n_quantize, n_fft, hop, win_length, center, pad_mode, normalized,
                               onesided, power, db, ref, amin):
    frames = _n_frames(codes.shape[-1], n_fft, hop, center)
    return _swapped(codes.shape[:-1], (frames, bank.shape[1]), torch.float32, codes.device, -2, -1)


_register('melspectrogram_mulaw', '(Tensor codes, Tensor window, Tensor filterbank, int n_quantize, %s, float power, bool db, '
          'float ref, float amin) -> Tensor' % _STFT_ARGS, _melspectrogram_mulaw_cuda, C.melspectrogram_mulaw,
          _melspectrogram_mulaw_fake, 3, differentiable=False)


# ============================================================================= apply_filterbank
def _apply_filterbank_cuda(spec, bank):
    _same_device('apply_filterbank', spec, bank)
    if H64.all_f64(spec, bank) and spec.dim() >= 2 and bank.dim() == 2 and spec.shape[-2] == bank.shape[0]:
        return H64.apply_filterbank(spec, bank)
    if _hip_dtype(spec, bank) is not None:
        _composite_route('apply_filterbank', _hip_dtype(spec, bank))
        return C.apply_filterbank(spec, bank)
    out = H.apply_filterbank(_f32(spec), _f32(bank))
    return out if spec.dtype == out.dtype else out.to(spec.dtype)


def _apply_filterbank_fake(spec, bank):
    return _swapped(spec.shape[:-2], (spec.shape[-1], bank.shape[1]), spec.dtype, spec.device, -2, -1)


_register('apply_filterbank', '(Tensor spec, Tensor filterbank) -> Tensor', _apply_filterbank_cuda,
          C.apply_filterbank, _apply_filterbank_fake, 2)


# ============================================================================= complex pairs
def _pairwise_cuda(op, hip_fn, composite_fn):
    def run(z, *args):
        if z.dtype == torch.float64 and z.dim() >= 1 and z.shape[-1] == 2:
            return getattr(H64, op)(z, *args)
        reason = _hip_dtype(z)
        if reason is not None:
            _composite_route(op, reason)
            return composite_fn(z, *args)
        out = hip_fn(_f32(z), *args)
        if z.dtype in _WIDEN:
            out = tuple(o.to(z.dtype) for o in out) if isinstance(out, tuple) else out.to(z.dtype)
        return out
    return run


def _pair_meta(z):
    """Layout of a per-pair result: the kernels keep the (dense) storage order of their input, halved."""
    if z.stride(-1) == 1 and H.is_dense(z) and not any(s % 2 for s, n in zip(z.stride()[:-1], z.shape[:-1]) if n > 1):
        return torch.empty_strided(z.shape[:-1], tuple(s // 2 for s in z.stride()[:-1]), dtype=z.dtype, device=z.device)
    return torch.empty(z.shape[:-1], dtype=z.dtype, device=z.device)


def _like_meta(x, dtype=None):
    return torch.empty_like(x, dtype=dtype) if H.is_dense(x) else torch.empty(x.shape, dtype=dtype or x.dtype,
                                                                              device=x.device)


def _complex_norm_fake(z, power):
    return _pair_meta(z)


_register('complex_norm', '(Tensor z, float power) -> Tensor', _pairwise_cuda('complex_norm', H.complex_norm,
                                                                               C.complex_norm),
          C.complex_norm, _complex_norm_fake, 1)
_register('angle', '(Tensor z) -> Tensor', _pairwise_cuda('angle', H.angle, C.angle), C.angle, _pair_meta, 1)
_register('magphase', '(Tensor z, float power) -> (Tensor, Tensor)', _pairwise_cuda('magphase', H.magphase, C.magphase),
          C.magphase, lambda z, power: (_pair_meta(z), _pair_meta(z)), 1)


# ============================================================================= phase vocoder
def _phase_vocoder_cuda(spec, phase_advance, rate):
    _same_device('phase_vocoder', spec, phase_advance)
    if spec.dtype == torch.float64 and phase_advance.is_floating_point():
        return H.phase_vocoder(spec, rate, phase_advance)          # the float64 kernel (reference's own test dtype)
    reason = _hip_dtype(spec, phase_advance)
    if reason is not None:
        _composite_route('phase_vocoder', reason)
        return C.phase_vocoder(spec, rate, phase_advance)
    out = H.phase_vocoder(_f32(spec), rate, _f32(phase_advance))
    return out if spec.dtype == out.dtype else out.to(spec.dtype)


def _phase_vocoder_cpu(spec, phase_advance, rate):
    return C.phase_vocoder(spec, rate, phase_advance)


def _phase_vocoder_fake(spec, phase_advance, rate):
    n_out = H.phase_vocoder_out_frames(spec.shape[-2], rate)
    return _swapped(spec.shape[:-3], (n_out, spec.shape[-3], 2), spec.dtype, spec.device, -3, -2)


_register('phase_vocoder', '(Tensor spec, Tensor phase_advance, float rate) -> Tensor', _phase_vocoder_cuda,
          _phase_vocoder_cpu, _phase_vocoder_fake, 2)


# ============================================================================= dB
def _unary_cuda(op, hip_fn, composite_fn):
    def run(x, *args):
        if x.dtype == torch.float64:
            return getattr(H64, op)(x, *args)
        reason = _hip_dtype(x)
        if reason is not None:
            _composite_route(op, reason)
            return composite_fn(x, *args)
        out = hip_fn(_f32(x), *args)
        return out if x.dtype == out.dtype else out.to(x.dtype)
    return run


_register('amplitude_to_db', '(Tensor x, float ref, float amin) -> Tensor',
          _unary_cuda('amplitude_to_db', H.amplitude_to_db, C.amplitude_to_db), C.amplitude_to_db,
          lambda x, ref, amin: _like_meta(x), 1)
_register('db_to_amplitude', '(Tensor x, float ref) -> Tensor',
          _unary_cuda('db_to_amplitude', H.db_to_amplitude, C.db_to_amplitude), C.db_to_amplitude,
          lambda x, ref: _like_meta(x), 1)


# ============================================================================= mu-law
def _mu_law_encoding_cuda(x, n_quantize):
    if not x.is_floating_point():
        x = x.to(torch.float)                              # reference functional.py:329-330
    if x.dtype == torch.float64:
        return H.mu_law_encoding_f64(x, n_quantize)        # the formula in double, like the reference's CPU path on double input
    reason = _hip_dtype(x)
    if reason is not None:
        _composite_route('mu_law_encoding', reason)
        return C.mu_law_encoding(x, n_quantize)
    return H.mu_law_encoding(_f32(x), n_quantize)


_register('mu_law_encoding', '(Tensor x, int n_quantize) -> Tensor', _mu_law_encoding_cuda, C.mu_law_encoding,
          lambda x, n_quantize: torch.empty(x.shape, dtype=torch.int64, device=x.device), 1, differentiable=False)


def _mu_law_decoding_cuda(codes, n_quantize, dtype):
    if not codes.is_floating_point():
        if dtype == torch.float32:
            return H.mu_law_decoding_int(codes, n_quantize)
        if dtype in _WIDEN:
            return H.mu_law_decoding_int(codes, n_quantize).to(dtype)
        if dtype == torch.float64:
            return H.mu_law_decoding_f64(codes.to(torch.int64), n_quantize)
        _composite_route('mu_law_decoding', 'dtype %s' % str(dtype).replace('torch.', ''))
        return C.mu_law_decoding(codes, n_quantize, dtype)
    if codes.dtype == torch.float64:
        return H.mu_law_decoding_f64(codes, n_quantize)
    reason = _hip_dtype(codes)
    if reason is not None:
        _composite_route('mu_law_decoding', reason)
        return C.mu_law_decoding(codes, n_quantize, dtype)
    out = H.mu_law_decoding_float(_f32(codes), n_quantize)
    return out if codes.dtype == out.dtype else out.to(codes.dtype)


def _mu_law_decoding_fake(codes, n_quantize, dtype):
    if codes.is_floating_point():
        return _like_meta(codes)
    return torch.empty(codes.shape, dtype=dtype, device=codes.device)


_register('mu_law_decoding', '(Tensor codes, int n_quantize, ScalarType dtype) -> Tensor', _mu_law_decoding_cuda,
          C.mu_law_decoding, _mu_law_decoding_fake, 1)

# ============================================================================= hpss
def _hpss_cuda(mag, kernel_f, kernel_t, power, hard):
    reason = _hip_dtype(mag)
    if reason is None and not H.hpss_supported(kernel_f, kernel_t):
        reason = 'kernel_size (%d, %d)' % (kernel_f, kernel_t)
    if reason is not None:
        _composite_route('hpss', reason)
        return C.hpss(mag, kernel_f, kernel_t, power, hard)
    outs = H.hpss(_f32(mag), kernel_f, kernel_t, power, hard)
    return outs if mag.dtype == torch.float32 else tuple(o.to(mag.dtype) for o in outs)


_register('hpss', '(Tensor mag, int kernel_f, int kernel_t, float power, bool hard) -> (Tensor, Tensor, Tensor, Tensor)',
          _hpss_cuda, C.hpss, lambda mag, kernel_f, kernel_t, power, hard: tuple(_like_meta(mag) for _ in range(4)), 1)

def _hpss_masks_cuda(mag, kernel_f, kernel_t, power, hard):
    reason = _hip_dtype(mag)
    if reason is None and not H.hpss_supported(kernel_f, kernel_t):
        reason = 'kernel_size (%d, %d)' % (kernel_f, kernel_t)
    if reason is not None:
        _composite_route('hpss', reason)
        return C.hpss(mag, kernel_f, kernel_t, power, hard)[2:]
    outs = H.hpss(_f32(mag), kernel_f, kernel_t, power, hard, masks_only=True)
    return outs if mag.dtype == torch.float32 else tuple(o.to(mag.dtype) for o in outs)


_register('hpss_masks', '(Tensor mag, int kernel_f, int kernel_t, float power, bool hard) -> (Tensor, Tensor)',
          _hpss_masks_cuda, lambda mag, kernel_f, kernel_t, power, hard: C.hpss(mag, kernel_f, kernel_t, power, hard)[2:],
          lambda mag, kernel_f, kernel_t, power, hard: tuple(_like_meta(mag) for _ in range(2)), 1)

ops = getattr(torch.ops, NS)
