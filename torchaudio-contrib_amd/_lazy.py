"""Deferred evaluation so that a plain ``nn.Sequential`` of the reference's layers runs as ONE kernel.

The reference composes Melspectrogram as ``Sequential(STFT, ComplexNorm(2), ApplyFilterbank)`` and
users append ``AmplitudeToDb()`` themselves (``tests/test_layers.py:69`` unpacks the factory result
with ``*``), so the fusion boundary cannot be a container we own.  Instead ``STFT.forward`` returns a
``DeferredSpectral`` — a ``torch.Tensor`` wrapper subclass with the right shape / strides / dtype /
device but no storage, holding the validated STFT plan.  ``ComplexNorm``, ``ApplyFilterbank`` and
``AmplitudeToDb`` extend the recipe when handed one; anything else (any torch op, ``.cpu()``,
printing, ``realize()``) materialises it through ``__torch_dispatch__`` by launching the single fused
HIP kernel that covers the recorded chain.  Nothing here computes on the CPU.
"""
import torch
from torch.utils._pytree import tree_map

_enabled = True


def set_lazy_fusion(flag):
    """Enable/disable deferred fusion of layer chains (on by default).  When off every layer launches
    its own kernel eagerly, exactly mirroring the reference's op-by-op evaluation."""
    global _enabled
    _enabled = bool(flag)


def lazy_fusion_enabled():
    return _enabled


class DeferredSpectral(torch.Tensor):
    """Result of an STFT-rooted layer chain that has not been launched yet."""

    @staticmethod
    def __new__(cls, plan, stage, shape, strides, power=None, filterbank=None, db=None):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, strides=strides, dtype=torch.float32,
                                                device=plan.wave.device, requires_grad=False)
        r._plan = plan
        r._stage = stage            # 'stft' | 'spec' | 'mel'
        r._power = power
        r._fb = filterbank
        r._db = db                  # None or (ref, amin)
        r._value = None
        return r

    # -- chain construction -----------------------------------------------------
    @classmethod
    def from_plan(cls, plan):
        shape = plan.lead + (plan.n_bins, plan.n_frames, 2)
        return cls(plan, 'stft', shape, _transposed_strides(plan.lead, (plan.n_frames, plan.n_bins, 2), -3, -2))

    def pending(self):
        return self._value is None

    def with_norm(self, power):
        p = self._plan
        return DeferredSpectral(p, 'spec', p.lead + (p.n_bins, p.n_frames),
                                _transposed_strides(p.lead, (p.n_frames, p.n_bins), -2, -1), power=power)

    def with_filterbank(self, fb):
        p = self._plan
        return DeferredSpectral(p, 'mel', p.lead + (fb.shape[1], p.n_frames),
                                _transposed_strides(p.lead, (p.n_frames, fb.shape[1]), -2, -1),
                                power=self._power, filterbank=fb)

    def with_db(self, ref, amin):
        return DeferredSpectral(self._plan, self._stage, tuple(self.shape), tuple(self.stride()),
                                power=self._power, filterbank=self._fb, db=(ref, amin))

    # -- materialisation --------------------------------------------------------
    def realize(self):
        if self._value is None:
            p = self._plan
            if self._stage == 'stft':
                v = p.run_stft()
            elif self._stage == 'spec':
                v = p.run_spectrogram(self._power, self._db)
            else:
                v = p.run_melspec(self._power, self._fb, self._db)
            self._value = v
            self._plan = None
            self._fb = None
        return self._value

    def __repr__(self):
        return 'DeferredSpectral(stage=%s, shape=%s, pending=%s)' % (self._stage, tuple(self.shape), self.pending())

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        def unwrap(a):
            return a.realize() if isinstance(a, DeferredSpectral) else a
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))


def _transposed_strides(lead, tail, a, b):
    """Strides of ``empty(lead + tail).transpose(a, b)``."""
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


def realize(x):
    """Materialise a deferred layer-chain result (no-op for ordinary tensors).  The kernel is enqueued
    on the current HIP stream; like any CUDA/HIP op it completes asynchronously."""
    if isinstance(x, DeferredSpectral):
        return x.realize()
    return x
