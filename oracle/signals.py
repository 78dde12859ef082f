"""Portable, counter-based synthetic waveforms (TEST INFRASTRUCTURE).

``uniform(shape, seed)`` is a pure function of (flat index, seed): splitmix64 on uint64,
top 24 bits → an exact float32 in [-1, 1).  No libm, no RNG stream state, so the golden
fixtures only need to store OUTPUTS — inputs regenerate bit-identically anywhere.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(z):
    with np.errstate(over='ignore'):
        z = (z + _GOLD)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform(shape, seed=0, scale=1.0):
    """float32 array in [-scale, scale); ``scale`` should be a power of two to stay exact."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
    bits = _splitmix64(ctr) >> np.uint64(40)                  # 24 random bits
    val = bits.astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    return (val * np.float32(scale)).reshape(shape)


def audio_like(shape, seed=0):
    """Uniform noise with a per-row power-of-two gain (2^0 … 2^-7) so rows exercise
    different dynamic ranges (and the dB clamp) while staying exactly reproducible."""
    x = uniform(shape, seed)
    rows = x.reshape(-1, shape[-1])
    gains = (2.0 ** -(np.arange(rows.shape[0]) % 8)).astype(np.float32)
    return (rows * gains[:, None]).reshape(shape)
