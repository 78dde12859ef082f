"""not-gpu: host logic — C-ABI surface, geometry, module contracts, error behaviour, sharding."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def tac():
    import torchaudio_contrib_amd as t
    if not os.path.exists(t._native.LIB_PATH):
        t.build_native()
    return t


def test_library_exports_every_declared_symbol(tac):
    header = open(os.path.join(ROOT, 'include', 'tac_amd.h')).read()
    declared = sorted(set(re.findall(r'\b(tac_[a-z0-9_]+)\s*\(', header)))
    assert len(declared) >= 16
    h = tac._native.lib()
    for name in declared:
        assert hasattr(h, name), name
    assert sorted(tac._native.EXPORTS) == declared
    assert h.tac_abi_version() == tac._native.ABI_VERSION == 5
    assert h.tac_strerror(-3).decode().startswith('input too short')
    # diagnostics (11): no launch yet on this thread, an empty probe is refused, NULL clears
    assert isinstance(h.tac_last_route(), bytes)
    assert h.tac_debug_clock_probe(ctypes.c_void_p(8), 0) == tac._native.TAC_E_INVALID
    assert h.tac_debug_clock_probe(None, 0) == tac._native.TAC_OK
    # route selection (12): the previous mode comes back, anything else than -1 / 0 / 1 is refused and changes nothing
    assert tac.set_fft_pipe('mfma') in (None, 'valu', 'mfma')
    assert h.tac_set_fft_pipe(7) == tac._native.TAC_E_INVALID
    assert tac.set_fft_pipe('valu') == 'mfma' and tac.set_fft_pipe(None) == 'valu' and tac.set_fft_pipe(None) is None
    with pytest.raises(ValueError):
        tac.set_fft_pipe('tensor')


def test_stale_library_is_refused(tac, monkeypatch):
    """A libtac_amd.so built from older sources fails the ABI check when it is loaded, not at the first missing symbol."""
    monkeypatch.setattr(tac._native, '_lib', None)
    monkeypatch.setattr(tac._native, 'ABI_VERSION', 99)
    with pytest.raises(tac._native.NativeLibraryError, match='ABI version 5'):
        tac._native.lib()


def test_geometry_helpers(tac):
    h = tac._native.lib()
    # T = (L + 2*pad - N + hop)//hop, tests/test_functional.py:14-15
    for L, n, hop in [(100000, 512, 256), (160000, 2048, 512), (16000, 512, 256), (2880000, 4096, 1024)]:
        assert h.tac_num_frames(L, n, hop, 1) == (L + 2 * (n // 2) - n + hop) // hop
    assert h.tac_num_frames(160000, 2048, 512, 1) == 313
    assert h.tac_num_frames(1000, 2048, 512, 0) == 0
    assert h.tac_num_bins(2048, 1) == 1025 and h.tac_num_bins(2048, 0) == 2048
    assert ctypes.sizeof(tac._native.StftDesc) == 56


def test_hip_route_fails_loudly_without_the_library(tac, monkeypatch, tmp_path):
    """The HIP launchers have no fallback: with libtac_amd.so absent every one of them raises (they all go through
    _native.lib()); and no launcher body mentions the stock-torch backend."""
    monkeypatch.setattr(tac._native, 'LIB_PATH', str(tmp_path / 'missing.so'))
    monkeypatch.setattr(tac._native, '_lib', None)
    with pytest.raises(tac._native.NativeLibraryError, match='no CPU fallback'):
        tac._native.lib()
    src = open(os.path.join(ROOT, 'torchaudio-contrib_amd', '_hip.py')).read()
    assert '_composite' not in src and 'torch.stft(' not in src and 'torch.fft' not in src and 'matmul' not in src
    assert src.count('_native.lib()') >= 15


def test_strict_mode_and_route_bookkeeping(tac):
    """set_strict(True) turns the stock-torch route into an error; CPU tensors are not subject to it (they never
    were candidates for the kernels)."""
    from torchaudio_contrib_amd import _ops
    assert _ops._hip_dtype(torch.zeros(2)) is None and _ops._hip_dtype(torch.zeros(2, dtype=torch.float16)) is None
    assert _ops._hip_dtype(torch.zeros(2, dtype=torch.float64)) == 'dtype float64'
    tac.set_strict(True)
    try:
        with pytest.raises(RuntimeError, match='strict mode'):
            _ops._composite_route('stft', 'dtype float64')
        assert tac.amplitude_to_db(torch.ones(3, dtype=torch.float64)).dtype == torch.float64   # CPU: unaffected
        # backward passes are strict too by default ('backward: ...' reasons come from the autograd entry); a caller can keep
        # strictness for forward calls only
        with pytest.raises(RuntimeError, match='strict mode'):
            _ops._composite_route('angle', 'backward: the op has no gradient kernel')
        tac.set_strict(True, backward=False)
        assert _ops.strict() and not _ops.strict_backward()
        _ops._warned.add(('angle', 'backward: the op has no gradient kernel'))
        _ops._composite_route('angle', 'backward: the op has no gradient kernel')          # runs (announced, counted)
        with pytest.raises(RuntimeError, match='strict mode'):
            _ops._composite_route('stft', 'dtype float64')
    finally:
        tac.set_strict(False)
    assert not _ops.strict() and not _ops.strict_backward()
    with pytest.warns(tac.CompositeRouteWarning):
        _ops._warned.discard(('stft', 'test reason'))
        _ops._composite_route('stft', 'test reason')
    assert _ops.composite_calls[('stft', 'test reason')] == 1


def test_ops_are_registered_with_torch_library(tac):
    """north_star: 'via PyTorch-ROCm custom ops' — every functional is a dispatcher op with CUDA, CPU, Meta and
    Autograd entries."""
    names = ['stft', 'spectrogram', 'melspectrogram', 'apply_filterbank', 'complex_norm', 'angle', 'magphase',
             'phase_vocoder', 'amplitude_to_db', 'db_to_amplitude', 'mu_law_encoding', 'mu_law_decoding']
    for n in names:
        op = getattr(torch.ops.tac_amd, n).default
        for key in ('CUDA', 'CPU', 'Meta'):
            assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), key), (n, key)
    # FakeTensor propagation gives the shapes AND the strided layout the kernels return
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(3, 2, 16000)
        z = tac.stft(x, 512, 128)
        assert tuple(z.shape) == (3, 2, 257, 126, 2) and z.stride() == (2 * 126 * 514, 126 * 514, 2, 514, 1)
        mel = tac.Melspectrogram(num_mels=40, sample_rate=16000, fft_length=512, hop_length=128)(x)
        assert tuple(mel.shape) == (3, 2, 40, 126) and mel.stride()[-2:] == (1, 40)
        assert tac.mu_law_encoding(x).dtype == torch.int64
        assert tuple(tac.phase_vocoder(z, 1.3, torch.empty(257, 1)).shape) == (3, 2, 257, 97, 2)


def test_torch_compile_traces_the_pipeline(tac):
    """The factory pipeline is one graph node under torch.compile (fullgraph: no break on the custom op)."""
    mel = tac.Melspectrogram(num_mels=32, sample_rate=16000, fft_length=256, hop_length=64)
    full = torch.nn.Sequential(*mel, tac.AmplitudeToDb())
    x = torch.randn(2, 1, 4000)
    seen = []

    def backend(gm, example_inputs):
        seen.append([str(n.target) for n in gm.graph.nodes if n.op == 'call_function'])
        return gm.forward
    got = torch.compile(mel, backend=backend, fullgraph=True)(x)
    assert [t.replace('.default', '') for t in seen[-1]] == ['tac_amd.melspectrogram'], seen
    assert torch.equal(got, mel(x))
    got = torch.compile(full, backend=backend, fullgraph=True)(x)
    assert any('tac_amd.amplitude_to_db' in t for t in seen[-1]) and all('tac_amd' in t for t in seen[-1]), seen
    assert torch.equal(got, full(x))


def test_layer_contracts(tac):
    layer = tac.STFT(fft_length=512, hop_length=256)
    assert torch.is_tensor(layer.window) and not layer.window.requires_grad
    assert layer.window.size(0) <= layer.fft_length
    assert torch.equal(layer.window, torch.hann_window(512))
    assert tac.STFT(512, win_length=400).window.shape == (400,)
    assert repr(layer) == ('STFT(fft_length=512, hop_length=256, win_length=None)'
                           '(center=True, pad_mode=reflect, normalized=False, onesided=True)')
    assert repr(tac.ComplexNorm(2.0)) == 'ComplexNorm(power=2.0)'
    assert repr(tac.AmplitudeToDb()) == 'AmplitudeToDb(ref=1.0, amin=1e-07)'
    assert repr(tac.DbToAmplitude(2.0)) == 'DbToAmplitude(ref=2.0)'
    assert repr(tac.MuLawEncoding()) == 'MuLawEncoding(n_quantize=256)'
    assert repr(tac.MuLawDecoding(16)) == 'MuLawDecoding(n_quantize=16)'
    assert repr(tac.TimeStretch(256, 257, 0.7)) == 'TimeStretch(fixed_rate=0.7)'
    assert repr(tac.MelFilterbank(sample_rate=16000)) == \
        'MelFilterbank(num_freqs=1025, snum_mels=128, min_freq=0.0, max_freq=8000), htk=False'
    ts = tac.TimeStretch(hop_length=256, num_freqs=1025)
    assert torch.is_tensor(ts.phase_advance) and ts.phase_advance.shape == (1025, 1)
    with pytest.raises(ValueError):
        ts(torch.zeros(1, 1025, 4, 2))


def test_factories_and_state_dict(tac):
    mel = tac.Melspectrogram(num_mels=128, sample_rate=16000, fft_length=2048, hop_length=512)
    assert isinstance(mel, torch.nn.Sequential)
    assert [type(c).__name__ for c in mel] == ['STFT', 'ComplexNorm', 'ApplyFilterbank']
    assert [n for n, _ in mel.named_buffers()] == ['0.window', '2.filterbank']
    assert mel[1].power == 2.0 and mel[2].filterbank.shape == (1025, 128)
    assert list(mel.parameters()) == [] and len(mel.state_dict()) == 0
    full = torch.nn.Sequential(*mel, tac.AmplitudeToDb())
    assert len(full) == 4 and len(full.state_dict()) == 0
    full.load_state_dict({})
    # num_freqs argument is ignored (layers.py:330-331 of the reference)
    assert tac.Melspectrogram(num_freqs=7, fft_length=512)[2].filterbank.shape == (257, 128)
    spec = tac.Spectrogram(512, hop_length=256)
    assert [type(c).__name__ for c in spec] == ['STFT', 'ComplexNorm'] and spec[1].power == 1.0
    with pytest.raises(TypeError):
        tac.Melspectrogram(num_mels=64)                      # missing fft_length
    with pytest.raises(ValueError):
        tac.MelFilterbank()                                  # neither max_freq nor sample_rate
    with pytest.raises(AssertionError):
        tac.AmplitudeToDb(ref=1e-8, amin=1e-7)

    class Flat(tac.Filterbank):                              # pluggable provider class
        def __init__(self, num_freqs, num_mels, **kw):
            self.shape = (num_freqs, num_mels)

        def get_filterbank(self):
            return torch.ones(self.shape)
    assert tac.Melspectrogram(num_mels=5, mel_filterbank=Flat, fft_length=64)[2].filterbank.sum() == 33 * 5
    with pytest.raises(NotImplementedError):
        tac.Filterbank().get_filterbank()


def test_import_surface(tac):
    for name in ['stft', 'complex_norm', 'create_mel_filter', 'apply_filterbank', 'angle', 'magphase',
                 'phase_vocoder', 'amplitude_to_db', 'db_to_amplitude', 'mu_law_encoding', 'mu_law_decoding',
                 'STFT', 'ComplexNorm', 'ApplyFilterbank', 'Filterbank', 'MelFilterbank', 'TimeStretch',
                 'Spectrogram', 'Melspectrogram', 'AmplitudeToDb', 'DbToAmplitude', 'MuLawEncoding',
                 'MuLawDecoding']:
        assert hasattr(tac, name), name
    import inspect
    sig = inspect.signature(tac.stft)
    assert list(sig.parameters) == ['waveforms', 'fft_length', 'hop_length', 'win_length', 'window', 'center',
                                    'pad_mode', 'normalized', 'onesided']
    assert sig.parameters['pad_mode'].default == 'reflect' and sig.parameters['onesided'].default is True
    assert list(inspect.signature(tac.Melspectrogram).parameters)[:7] == [
        'num_mels', 'sample_rate', 'min_freq', 'max_freq', 'num_freqs', 'htk', 'mel_filterbank']
    assert inspect.signature(tac.mu_law_decoding).parameters['dtype'].default == torch.get_default_dtype()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'torchaudio-contrib_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('the oracle', '').replace('CPU oracle', ''), f
                assert '/root/reference' not in src, f


def test_only_the_checkers_import_the_oracle():
    """oracle/ is test infrastructure: outside tests/ only smoke() and bench.py's cpu_baseline leg may import it."""
    import re
    allowed = {'bench.py', '__graft_entry__.py'}
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in ('.git', 'gpurun_out', 'gpurun_variants', '__pycache__', 'tests', 'oracle')]
        for f in files:
            if not f.endswith('.py'):
                continue
            rel = os.path.relpath(os.path.join(dirpath, f), ROOT)
            src = open(os.path.join(dirpath, f)).read()
            if re.search(r'^\s*(from\s+oracle\b|import\s+oracle\b)', src, re.M):
                assert rel in allowed, rel


def test_shard_bounds(tac):
    from torchaudio_contrib_amd.distributed import shard_bounds
    for n, w in [(2048, 8), (10, 3), (3, 8), (256, 1)]:
        cuts = [shard_bounds(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [e - b for b, e in cuts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd.distributed import shard_batch, all_gather_batch, ShardedPipeline
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group('gloo', rank=rank, world_size=world)
whole = torch.arange(5 * 2 * 3 * 4, dtype=torch.float32).reshape(5, 2, 3, 4)     # uneven: 3 + 2 rows
local = shard_batch(whole)
assert local.shape[0] == (3 if rank == 0 else 2)
# strided (.., M, T) view like the layers return: physical (.., T, M)
phys = local.transpose(-2, -1).contiguous()
view = phys.transpose(-2, -1)
out = all_gather_batch(view, total_rows=5)
assert out.shape == whole.shape and torch.equal(out, whole), rank
out2 = all_gather_batch(local.contiguous())                   # row count discovered by all_reduce
assert torch.equal(out2, whole)
even = torch.arange(4 * 6, dtype=torch.float32).reshape(4, 6)
assert torch.equal(all_gather_batch(shard_batch(even).clone(), total_rows=4), even)   # all_gather_into_tensor path
pipe = ShardedPipeline(torch.nn.Identity(), gather=True)
assert torch.equal(pipe(whole), whole)
assert torch.equal(ShardedPipeline(torch.nn.Identity(), gather=False)(whole), local)
# fewer rows than ranks: rank 1 owns nothing but still enters the collective (no hang), result is the single row
one = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(1, 2, 3, 4)
assert shard_batch(one).shape[0] == (1 if rank == 0 else 0)
assert torch.equal(ShardedPipeline(torch.nn.Identity(), gather=True)(one), one)
mel = tac.Melspectrogram(num_mels=8, sample_rate=8000, fft_length=64, hop_length=16)
wave = torch.arange(1 * 1 * 400, dtype=torch.float32).reshape(1, 1, 400).sin()
assert torch.equal(ShardedPipeline(mel, gather=True)(wave), mel(wave))
# a 2-D transposed local (M, T) view: dim 0 is not the physical dim 0, so it must not take the transposed fast path
mt = torch.arange(6 * 5, dtype=torch.float32).reshape(5, 6).t()[2 * rank:2 * rank + 2]       # rows of a (6, 5) view
assert torch.equal(all_gather_batch(mt, total_rows=4), torch.arange(30, dtype=torch.float32).reshape(5, 6).t()[:4])
# the one-shot direct exchange (every rank sends its shard to each peer in one batched group of sends / receives): same results
assert torch.equal(all_gather_batch(view, total_rows=5, method='p2p'), whole)                    # uneven, strided view
assert torch.equal(all_gather_batch(shard_batch(even).clone(), total_rows=4, method='p2p'), even)
assert torch.equal(all_gather_batch(mt, total_rows=4, method='p2p'), torch.arange(30, dtype=torch.float32).reshape(5, 6).t()[:4])
os.environ['TAC_ALLGATHER'] = 'p2p'
assert torch.equal(ShardedPipeline(torch.nn.Identity(), gather=True)(one), one)                  # a rank with zero rows
assert torch.equal(ShardedPipeline(mel, gather=True)(wave), mel(wave))
os.environ['TAC_ALLGATHER'] = 'ring'
try:
    all_gather_batch(view, total_rows=5)
    raise SystemExit('an unknown TAC_ALLGATHER must raise')
except ValueError:
    pass
os.environ['TAC_ALLGATHER'] = 'rccl'
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_gloo_world2_shard_and_allgather(tmp_path):
    """N>1 control flow (shard -> local pipeline -> single all-gather) on CPU with gloo, world_size 2."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % ROOT)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '2', port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ok' % r in o


def test_gloo_world3_p2p_allgather(tmp_path):
    """The direct exchange with three ranks (two peers per rank, staggered send order) and an uneven 7-row batch."""
    script = tmp_path / 'worker3.py'
    script.write_text(r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from torchaudio_contrib_amd.distributed import shard_batch, all_gather_batch
rank, world = int(sys.argv[1]), 3
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], RANK=str(rank), WORLD_SIZE='3')
dist.init_process_group('gloo', rank=rank, world_size=world)
whole = torch.arange(7 * 2 * 3 * 4, dtype=torch.float32).reshape(7, 2, 3, 4)
view = shard_batch(whole).transpose(-2, -1).contiguous().transpose(-2, -1)
for method in ('p2p', 'rccl'):
    assert torch.equal(all_gather_batch(view, total_rows=7, method=method), whole), (rank, method)
even = torch.arange(6 * 5, dtype=torch.float32).reshape(6, 5)
assert torch.equal(all_gather_batch(shard_batch(even).clone(), method='p2p'), even)
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
''' % ROOT)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ok' % r in o


_OVERLAP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd.distributed import shard_batch, all_gather_batch, ShardedPipeline, ChunkedAllGather
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group('gloo', rank=rank, world_size=world)


class FrameMajor(torch.nn.Module):          # what the layers return: a (.., M, T) view of a frame-major buffer
    def forward(self, x):
        return (x * 2.0).transpose(-2, -1).contiguous().transpose(-2, -1)


for rows in (1, 2, 5, 7, 12):               # fewer rows than ranks, uneven shards, even shards
    whole = torch.arange(rows * 2 * 3 * 4, dtype=torch.float32).reshape(rows, 2, 3, 4)
    for method in ('rccl', 'p2p'):
        for chunks in (1, 2, 3, 5, 64):     # one piece, uneven pieces, more pieces than rows
            got = ShardedPipeline(FrameMajor(), overlap=True, chunks=chunks, method=method)(whole)
            assert got.shape == whole.shape and torch.equal(got, whole * 2.0), (rank, rows, method, chunks)
            assert got.stride() == FrameMajor()(whole).stride(), 'the gathered batch keeps the frame-major layout'
# pieces must come in order, and all of them
g = ChunkedAllGather(4 * world, 2)
try:
    g.add(1, torch.zeros(2, 3))
    raise SystemExit('out-of-order piece accepted')
except ValueError:
    pass
# ... and look like the first one (trailing shape, dtype, layout): a piece that does not is refused, not misplaced
g = ChunkedAllGather(4 * world, 2)
g.add(0, torch.zeros(2, 3))
for bad in (torch.zeros(2, 4), torch.zeros(2, 3, dtype=torch.float64)):
    try:
        g.add(1, bad)
        raise SystemExit('a piece of another layout was accepted')
    except ValueError:
        g._next = 1
g.add(1, torch.zeros(2, 3))
g.finish()


class Counting(torch.nn.Module):            # a stateful pipeline: how often, and on how many rows, was it called?
    def __init__(self):
        super().__init__()
        self.calls, self.rows = 0, 0

    def forward(self, x):
        self.calls += 1
        self.rows += x.shape[0]
        return x + 1.0


# a rank never runs the pipeline on rows it does not own more than the ONE borrowed row an empty shard needs for the layout
for rows in (1, 2, 5, 7):
    whole = torch.arange(rows * 3, dtype=torch.float32).reshape(rows, 3)
    own = shard_batch(whole, world, rank).shape[0]
    for chunks in (1, 3, 8):
        pipe = Counting()
        got = ShardedPipeline(pipe, overlap=True, chunks=chunks)(whole)
        assert torch.equal(got, whole + 1.0)
        assert pipe.rows == (own if own else 1) and pipe.calls <= max(1, min(chunks, own if own else 1)), (rank, rows, chunks, pipe.calls, pipe.rows)
# the real chain (CPU route of the ops): overlapped == serial == whole batch
mel = torch.nn.Sequential(*tac.Melspectrogram(num_mels=8, sample_rate=8000, fft_length=64, hop_length=16), tac.AmplitudeToDb())
wave = torch.arange(5 * 1 * 400, dtype=torch.float32).reshape(5, 1, 400).sin()
ref = mel(wave)
for method in ('rccl', 'p2p'):
    assert torch.equal(ShardedPipeline(mel, overlap=True, chunks=2, method=method)(wave), ref)
    assert torch.equal(ShardedPipeline(mel, overlap=False, method=method)(wave), ref)
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_overlapped_allgather(tmp_path, world):
    """ChunkedAllGather / ShardedPipeline(overlap=True): the exchange of row piece k posted while piece k + 1 is computed gives
    the same batch as the one-shot gather — chunk order, uneven chunk tails, uneven and empty shards, both methods."""
    script = tmp_path / 'worker_overlap.py'
    script.write_text(_OVERLAP_WORKER % ROOT)
    port = str(33500 + (os.getpid() + 7 * world) % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ok' % r in o


def test_compiled_binding_loads_and_registers_its_op():
    """csrc/binding/tac_ext.cpp: the extension module loads next to the ctypes binding and registers tac_amd::melspec_planned
    (CUDA + Meta kernels) with the dispatcher; TAC_AMD_EXT=0 falls back to ctypes."""
    import torchaudio_contrib_amd as tac
    assert os.path.exists(tac._native.EXT_PATH), 'run `make -C torchaudio-contrib_amd/csrc` (or __graft_entry__.build())'
    assert tac._native.binding() == 'compiled' and tac._native.ext().ABI == 1
    schema = str(torch.ops.tac_amd.melspec_planned.default._schema)
    assert 'Tensor wave' in schema and 'int plan' in schema
    with pytest.raises(RuntimeError, match='unknown plan'):
        torch.ops.tac_amd.melspec_planned(torch.zeros(2, 8, device='meta'), 12345)
    out = subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); import torchaudio_contrib_amd as t; '
                          'print(t._native.binding())' % ROOT], env=dict(os.environ, TAC_AMD_EXT='0'),
                         stdout=subprocess.PIPE, text=True)
    assert out.stdout.strip() == 'ctypes'


def test_invalidate_through_an_alias_drops_everything():
    """invalidate(t) drops only t's tables when t is the object they hang on; an alias that carries none (t.data, a view) cannot
    name that object, so it invalidates globally instead of silently doing nothing (round-4 advisor finding)."""
    from torchaudio_contrib_amd import _hip
    t = torch.rand(8, 4)
    t._tac_pack = ('stamp', 'tables')
    e0 = _hip._epoch
    _hip.invalidate(t)
    assert not hasattr(t, '_tac_pack') and _hip._epoch == e0              # targeted
    t._tac_pack = ('stamp', 'tables')
    _hip.invalidate(t.data)                                                # an alias: nothing cached on it
    assert _hip._epoch == e0 + 1, 'global invalidation'
    assert _hip._stamp(t)[2] == e0 + 1                                     # every stamp taken before is now stale
    _hip.invalidate(t[2:])
    assert _hip._epoch == e0 + 2


def test_forced_collective_in_a_group_of_one():
    """all_gather_batch(force_collective=True) runs the real collective at world size 1 (what the RCCL smoke test of a 1-GPU
    box relies on) and returns the shard unchanged, for both exchange methods and for the strided views the layers return."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from torchaudio_contrib_amd.distributed import all_gather_batch
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], RANK='0', WORLD_SIZE='1')
dist.init_process_group('gloo', rank=0, world_size=1)
calls = []
real = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
phys = torch.arange(4 * 1 * 6 * 5, dtype=torch.float32).reshape(4, 1, 6, 5)
view = phys.transpose(-2, -1)
assert all_gather_batch(view) is view and not calls                           # the early return stays the default
out = all_gather_batch(view, total_rows=4, force_collective=True)
assert len(calls) == 1 and out.shape == view.shape and out.stride() == view.stride() and torch.equal(out, view)
assert out.data_ptr() != view.data_ptr()
assert torch.equal(all_gather_batch(view, force_collective=True, method='p2p'), view)
dist.destroy_process_group()
print('ok')
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code, str(33500 + os.getpid() % 2000)], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=180)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stdout


def test_bench_gpus_flag_is_real():
    """`python bench.py --gpus N` is its own launcher (one rank per GPU) and refuses to run fewer ranks than asked:
    the spawn / rendezvous / max-over-ranks / single-JSON-line control flow on CPU tensors over gloo, and the loud
    failure when the GPUs are not there."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run-cpu', '--steps', '2',
                          '--warmup', '1'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['valid'] is False and rec['value'] > 0
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
        assert bad.returncode != 0 and 'GPU(s) visible' in bad.stderr and not bad.stdout.strip()
    mismatch = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--dry-run-cpu'],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(env, WORLD_SIZE='1', RANK='0'), timeout=300)
    assert mismatch.returncode != 0 and 'WORLD_SIZE=1' in mismatch.stderr


_MEDIAN_CHECK = r'''
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "median_run.hpp"
template <int K> int check(int trials) {
    int bad = 0;
    for (int t = 0; t < trials; ++t) {
        float w[K + 3];
        const int mode = t % 4;
        for (int i = 0; i < K + 3; ++i) {
            if (mode == 0) w[i] = (float)rand() / RAND_MAX;
            else if (mode == 1) w[i] = (float)(rand() % 5);                 // many ties
            else if (mode == 2) w[i] = (rand() % 7 == 0) ? INFINITY : (float)(rand() % 100);
            else w[i] = (float)i * ((t & 8) ? 1.f : -1.f);                  // sorted / reversed
        }
        float med[4];
        median_run4<K>(w, med);
        for (int j = 0; j < 4; ++j) {
            std::vector<float> v(w + j, w + j + K);
            std::nth_element(v.begin(), v.begin() + K / 2, v.end());
            if (v[K / 2] != med[j]) ++bad;
        }
    }
    for (int t = 0; t < trials; ++t) {                                      // runs of eight windows (two levels of sharing)
        float w[K + 7];
        const int mode = t % 4;
        for (int i = 0; i < K + 7; ++i) {
            if (mode == 0) w[i] = (float)rand() / RAND_MAX;
            else if (mode == 1) w[i] = (float)(rand() % 5);
            else if (mode == 2) w[i] = (rand() % 7 == 0) ? INFINITY : (float)(rand() % 100) - 50.f;
            else w[i] = (float)i * ((t & 8) ? 1.f : -1.f);
        }
        float med[8];
        median_run8<K>(w, med);
        for (int j = 0; j < 8; ++j) {
            std::vector<float> v(w + j, w + j + K);
            std::nth_element(v.begin(), v.begin() + K / 2, v.end());
            if (v[K / 2] != med[j]) ++bad;
        }
    }
    return bad;
}
int main() {
    int b = check<9>(4000) + check<11>(4000) + check<13>(4000) + check<15>(4000) + check<17>(4000) + check<19>(4000) +
            check<21>(4000) + check<23>(4000) + check<25>(4000) + check<27>(4000) + check<29>(4000) + check<31>(4000) +
            check<33>(1000) + check<35>(1000) + check<37>(1000) + check<39>(1000) + check<41>(1000) + check<43>(1000) +
            check<45>(1000) + check<47>(1000) + check<49>(1000) + check<51>(1000) + check<53>(1000) + check<55>(1000) +
            check<57>(1000) + check<59>(1000) + check<61>(1000) + check<63>(1000);
    float a[32];
    for (int t = 0; t < 2000; ++t) {
        for (int i = 0; i < 32; ++i) a[i] = (float)(rand() % 50);
        std::vector<float> v(a, a + 32);
        std::sort(v.begin(), v.end());
        sort_net<32>(a);
        for (int i = 0; i < 32; ++i) b += a[i] != v[i];
    }
    printf("bad=%d\n", b);
    return b != 0;
}
'''


def test_median_run_matches_nth_element(tmp_path):
    """csrc/median_run.hpp (the HPSS kernels' register sorting network and their shared-sort median selection of four /
    eight overlapping windows) is plain C++: compiled for the host and checked against std::nth_element for every width the
    tile kernel is instantiated for — random values, heavy ties, infinities, sorted and reversed runs."""
    src = tmp_path / 'check.cpp'
    src.write_text(_MEDIAN_CHECK)
    exe = str(tmp_path / 'check')
    subprocess.run(['g++', '-O1', '-I', os.path.join(ROOT, 'torchaudio-contrib_amd', 'csrc'), '-o', exe, str(src)], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and out.stdout.strip() == 'bad=0', out.stdout


def _emulate_pieces(plan, row):
    """What melspec_stream3_kernel's piece contraction computes for one |X|^p row (float64): per lane and segment the dot
    product of L four-tap steps, the shifted adds over a band's pieces, the staging write of the band's last piece."""
    steps, first, band, index, w = plan
    out = {}
    base = 0
    padded = np.concatenate([row, np.zeros(16)])
    for s in range(3):
        part = np.zeros(64)
        for l in range(64):
            f = first[s * 64 + l]
            for j in range(steps[s]):
                part[l] += float(np.dot(w[base + j, l].astype(np.float64), padded[f + 4 * j:f + 4 * j + 4]))
        for l in range(64):
            e = s * 64 + l
            i, last = index[e] & 255, index[e] >= 256
            if band[e] < 0:
                assert part[l] == 0.0 and index[e] == 0                 # an unused lane-segment carries zero weights
                continue
            assert 0 <= i <= 2 and l % 16 >= i                          # the shifted reads stay inside the lane's 16-lane row
            tot = part[l]
            for d in range(1, i + 1):
                assert band[e - d] == band[e] and (index[e - d] & 255) == i - d
                tot += part[l - d]
            if last:
                assert band[e] not in out
                out[band[e]] = tot
        base += steps[s]
    return out


@pytest.mark.parametrize('n_mels,sr,htk', [(128, 16000, False), (128, 44100, False), (96, 22050, True), (128, 16000, True),
                                            (96, 22050, False), (160, 16000, False)])
def test_piece_layout_of_the_filterbank_matches_the_dense_product(tac, n_mels, sr, htk):
    """tac_melbank_plan_pieces_host (csrc/mel_pieces.hpp): every band is stored exactly once, its pieces sit in adjacent
    lanes of one 16-lane row, every read stays inside the row buffer, and the contraction emulated on the plan equals
    row @ bank (reference functional.py:183-184) for random rows — for the standard banks the benchmark and the tests use."""
    h = tac._native.lib()
    fb = tac.create_mel_filter(1025, n_mels, 0.0, sr / 2.0, htk).numpy().astype(np.float32)
    fb = np.ascontiguousarray(fb)
    steps = (ctypes.c_int32 * 3)()
    first, band, index = (np.zeros(192, dtype=np.int32) for _ in range(3))
    w = np.zeros(256 * 24, dtype=np.float32)
    rc = h.tac_melbank_plan_pieces_host(fb.ctypes.data, 1025, n_mels, ctypes.cast(steps, ctypes.c_void_p), first.ctypes.data,
                                        band.ctypes.data, index.ctypes.data, w.ctypes.data, w.size)
    assert rc == tac._native.TAC_OK, rc
    steps = list(steps)
    total = sum(steps)
    assert total < 18 and all(1 <= x <= 7 for x in steps)
    assert sorted(set(b for b in band if b >= 0)) == list(range(n_mels))
    assert max(np.bincount(band[band >= 0])) <= 3                          # at most three pieces per band
    assert (first % 4 == 0).all() and (first >= 0).all()
    for s in range(3):
        assert (first[s * 64:(s + 1) * 64] + 4 * steps[s] <= 1032).all()      # inside the 1025 + 7 floats of a row buffer
    plan = (steps, first, band, index, w[:256 * total].reshape(total, 64, 4))
    rng = np.random.default_rng(5)
    for _ in range(3):
        row = rng.random(1025)
        got = _emulate_pieces(plan, row)
        want = row @ fb.astype(np.float64)
        err = max(abs(got[m] - want[m]) for m in range(n_mels))
        assert err < 1e-9 * max(1.0, np.abs(want).max()), err
    # starts of a 16-byte read group fall into different bank groups wherever the slack allows
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    worst = 0
    for s in range(3):
        for gi in range(4):
            lanes = [l + 32 * (gi >> 1) for l in groups[gi & 1]]
            res = [(first[s * 64 + l] // 4) % 16 for l in lanes]
            worst = max(worst, max(res.count(r) for r in set(res)))
    assert worst <= 6, worst


def test_piece_layout_refuses_what_it_cannot_hold(tac):
    h = tac._native.lib()
    fb = np.ascontiguousarray(tac.create_mel_filter(1025, 40, 0.0, 8000.0, False).numpy().astype(np.float32))   # bands of 29 quads
    steps = (ctypes.c_int32 * 3)()
    first, band, index = (np.zeros(192, dtype=np.int32) for _ in range(3))
    w = np.zeros(256 * 24, dtype=np.float32)
    rc = h.tac_melbank_plan_pieces_host(fb.ctypes.data, 1025, 40, ctypes.cast(steps, ctypes.c_void_p), first.ctypes.data,
                                        band.ctypes.data, index.ctypes.data, w.ctypes.data, w.size)
    assert rc == tac._native.TAC_E_UNSUPPORTED


@pytest.mark.parametrize('n_fft,n_mels,sr,htk', [(2048, 128, 16000, False), (2048, 80, 16000, False), (2048, 40, 16000, False),
                                                 (2048, 160, 22050, True), (4096, 128, 48000, False), (4096, 80, 44100, False),
                                                 (4096, 200, 48000, False), (4096, 40, 16000, False), (4096, 64, 22050, True)])
def test_lane_tables_of_the_fused_kernels_match_the_dense_product(tac, n_fft, n_mels, sr, htk):
    """tac_melbank_pack_host (round 6): the filterbank tables of the fused fft_length-2048 / 4096 kernels, built without a device — cells
    from the widest end of the bank (2048: info[2] = 64 + 256; 4096: info[7]), per-slot step counts, bank-aware first bins, and at 4096
    the cut layout with its mix table (40 bands).  The kernels' contraction emulated on the tables equals row @ bank (reference
    functional.py:183-184) for random rows; every read stays inside the row and its three zeroed slack floats."""
    h = tac._native.lib()
    n_freqs = n_fft // 2 + 1
    fb = np.ascontiguousarray(tac.create_mel_filter(n_freqs, n_mels, 0.0, sr / 2.0, htk).numpy().astype(np.float32))
    wpack, desc, info = np.zeros(24576, dtype=np.float32), np.zeros(8192, dtype=np.int32), (ctypes.c_int32 * 8)()
    h.tac_melbank_pack_host.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    rc = h.tac_melbank_pack_host(fb.ctypes.data, n_freqs, n_mels, n_fft, wpack.ctypes.data, wpack.size, desc.ctypes.data, desc.size,
                                 ctypes.cast(info, ctypes.c_void_p))
    assert rc == tac._native.TAC_OK, rc
    info = [int(v) for v in info]
    rng = np.random.default_rng(5)
    rows = rng.random((3, n_freqs)).astype(np.float64)
    want = rows @ fb.astype(np.float64)

    def cell_sum(row, first, base, steps, lane):                         # one cell: `steps` four-tap steps from bin `first`
        assert first % 4 == 0 and 0 <= first and first + 4 * steps <= n_freqs + (7 if n_fft == 2048 else 3)
        acc = 0.0
        for j in range(steps):
            taps = wpack[((base + j) * 64 + lane) * 4:((base + j) * 64 + lane) * 4 + 4].astype(np.float64)
            bins = np.arange(first + 4 * j, first + 4 * j + 4)
            vals = np.where(bins < n_freqs, row[np.minimum(bins, n_freqs - 1)], 0.0)
            acc += float(taps @ vals)
        return acc

    got = np.zeros_like(want)
    if n_fft == 2048:
        assert info[2] in (64, 64 + 256) and info[0] == 256 * info[3]
        rev, nslot, steps = info[2] != 64, info[1], info[4:4 + info[1]]
        assert rev == (n_mels % 64 != 0)
        for r, row in enumerate(rows):
            base = 0
            for s in range(nslot):
                for lane in range(64):
                    c = 64 * s + lane
                    if c < n_mels:
                        got[r, n_mels - 1 - c if rev else c] = cell_sum(row, int(desc[c]), base, steps[s], lane)
                base += steps[s]
    else:
        floats, slots, mark, total, waves, np_, rounds, rev = info
        assert mark == 1000 + 4096 and floats == 256 * total and waves in (12, 11, 8) and rounds == (n_mels + 63) // 64
        pairs = [int(v) for v in desc[64 * 6:64 * 6 + 6]]
        assert sum(2 * p for p in pairs) == total and all(p == 0 for p in pairs[slots:])
        for r, row in enumerate(rows):
            cells, base = np.zeros(64 * 6 + 1), 0
            for s in range(slots):
                for lane in range(64):
                    cells[64 * s + lane] = cell_sum(row, int(desc[64 * s + lane]), base, 2 * pairs[s], lane)
                base += 2 * pairs[s]
            if np_ == 0:                                                  # uncut: cell c is band c (or n - 1 - c)
                for c in range(n_mels):
                    got[r, n_mels - 1 - c if rev else c] = cells[c]
            else:                                                         # cut: every band gathers its pieces through the mix table
                assert np_ % 4 == 0 and not rev
                mix = desc[400:400 + 64 * rounds * np_].reshape(rounds, np_, 64)
                assert mix.min() >= 0 and mix.max() <= 64 * 6
                for b in range(n_mels):
                    got[r, b] = cells[mix[b // 64, :, b % 64]].sum()
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()
    if n_fft == 4096 and n_mels == 40:
        assert info[5] > 0                                                # (bands of 300 bins: the cut layout)
    if n_fft == 4096 and n_mels == 128:
        assert info[5] == 0 and info[4] == 11                             # (uncut, eleven waves beside the 40 KB table)


@pytest.mark.parametrize('n_fft,n_mels,sr', [(1024, 80, 22050), (1024, 40, 16000), (1024, 128, 22050), (512, 80, 16000), (512, 40, 16000),
                                             (400, 80, 16000), (400, 40, 16000), (256, 40, 16000)])
def test_lane_layout_tables_match_the_dense_product(tac, n_fft, n_mels, sr):
    """tac_melbank_pack_host for the kernels that keep several frames per wave (fft_length 256 / 400 / 512 / 1024: 8 ... 32 lanes per
    frame, csrc/mel_lanes.hpp): cell c = lanes i + l holds band c, or band n - 1 - c (info[5], round 6) when the band count is not a multiple
    of the lanes; a uniform S steps per band in the table, the slots' own step pairs behind the first bins.  The contraction emulated on
    the table equals row @ bank; at 1024 the 40-band bank (bands of 70 - 80 bins: S = 18 ... 20) is accepted since round 6."""
    h = tac._native.lib()
    n_freqs = n_fft // 2 + 1
    fb = np.ascontiguousarray(tac.create_mel_filter(n_freqs, n_mels, 0.0, sr / 2.0, False).numpy().astype(np.float32))
    wpack, desc, info = np.zeros(24576, dtype=np.float32), np.zeros(8192, dtype=np.int32), (ctypes.c_int32 * 8)()
    h.tac_melbank_pack_host.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    rc = h.tac_melbank_pack_host(fb.ctypes.data, n_freqs, n_mels, n_fft, wpack.ctypes.data, wpack.size, desc.ctypes.data, desc.size,
                                 ctypes.cast(info, ctypes.c_void_p))
    assert rc == tac._native.TAC_OK, rc
    wtot, nslot, mark, _, S, rev = [int(v) for v in info][:6]
    lanes = mark - 1000
    assert lanes == (8 if n_fft == 400 else n_fft // 32) and nslot == (n_mels + lanes - 1) // lanes
    assert bool(rev) == (n_mels % lanes != 0)
    if n_fft == 1024 and n_mels == 40:
        assert S > 12                                                     # (the wide-band step counts of round 6)
    rng = np.random.default_rng(6)
    rows = rng.random((3, n_freqs)).astype(np.float64)
    want = rows @ fb.astype(np.float64)
    got = np.zeros_like(want)
    for r, row in enumerate(rows):
        for i in range(nslot):
            for l in range(lanes):
                c = lanes * i + l
                if c >= n_mels:
                    continue
                first = int(desc[lanes * i + l])
                assert first % 4 == 0 and first >= 0
                acc = 0.0
                for j in range(S):
                    at = ((i * S + j) * lanes + l) * 4
                    bins = np.arange(first + 4 * j, first + 4 * j + 4)
                    vals = np.where(bins < n_freqs, row[np.minimum(bins, n_freqs - 1)], 0.0)
                    acc += float(wpack[at:at + 4].astype(np.float64) @ vals)
                got[r, n_mels - 1 - c if rev else c] = acc
    assert np.abs(got - want).max() < 1e-5 * np.abs(want).max()


def test_float64_size_coverage_matches_the_kernel_plan(tac):
    """``_hip64.covers``: even lengths <= 8192 with a 5-smooth half take the LDS Stockham transform of csrc/chain_f64.hip; other
    lengths go to its O(N^2) direct transform only up to ``DIRECT_MAX`` (longer ones stay on the announced stock-torch route,
    which is faster there although the kernel would accept any length <= 4096)."""
    h64 = tac._ops.H64
    assert h64.DIRECT_MAX == 512
    assert all(h64.covers(n) for n in (1, 4, 77, 134, 400, 511, 512, 2048, 4096, 6000, 8192, 5000))
    assert not any(h64.covers(n) for n in (513, 1001, 4094, 4097, 8190, 8194, 16384, 4099 * 2))


_OLA_RUNS_CHECK = r"""
#include "ola_runs.hpp"
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
// ola_direct() of csrc/ola_plan.hpp restated (it is device code): run f of a row was stored by the backward kernel itself
static bool direct(int f, int S, int hop, int pad, int L, int T, int open, bool plain) {
    if (f >= T) return false;
    const int sg = f / S;
    if (sg > 0 && (f - sg * S) * hop < open) return false;
    const int jlo = f * hop - pad, jhi = jlo + hop - 1;
    if (plain) return jlo >= 0 && jhi < L;
    return jlo > pad && jhi < L - 1 - pad;
}
int main() {
    srand(7);
    int bad = 0, cases = 0;
    const int sizes[5] = {2048, 1024, 512, 400, 256};
    for (int it = 0; it < 6000; ++it) {
        const int n = sizes[rand() % 5];
        const int hop = n == 400 ? 4 * (13 + rand() % 88) : (n / 16) * (1 + rand() % 16);
        if (hop % 4) continue;
        const bool center = rand() % 10 < 7, plain_mode = rand() % 2;
        const int pad = center ? n / 2 : 0;
        const bool plain = pad == 0 || plain_mode;
        const int L = n + 1 + rand() % (60 * hop);
        const int T = 1 + (L + 2 * pad - n) / hop;
        const int open = n - hop, smin = open > 0 ? (open + hop - 1) / hop : 1;
        const int S = n == 400 ? 8 : smin + rand() % 10;
        if (S < smin || T < 1) continue;
        const int spr = (T + S - 1) / S;
        std::set<int> want;
        for (int j = 0; j < L; ++j)
            if (!direct((j + pad) / hop, S, hop, pad, L, T, open, plain)) want.insert(j);
        const tac::OlaRuns r = tac::ola_runs_for(L, pad, hop, n, T, plain);
        std::vector<int> got;
        auto emit = [&](int fc) {
            for (int i = 0; i < hop; ++i) {
                const int j = fc * hop + i - pad;
                if (j >= 0 && j < L) got.push_back(j);
            }
        };
        for (int fc = 0; fc < r.head; ++fc) emit(fc);                       // ola_fold_runs_kernel's three kinds of slots
        for (int fc = r.tail_first; fc <= r.last_run; ++fc) emit(fc);
        for (int s = 1; s < spr; ++s)
            for (int q = 0; q < r.zone_frames; ++q) {
                const int fc = s * S + q;
                if (fc >= r.head && fc < r.tail_first) emit(fc);
            }
        std::set<int> uniq(got.begin(), got.end());
        ++cases;
        if (uniq.size() != got.size() || uniq != want) {
            ++bad;
            if (bad < 5) printf("n=%d hop=%d L=%d pad=%d plain=%d S=%d: got %zu (%zu unique), want %zu\n", n, hop, L, pad, (int)plain, S,
                                got.size(), uniq.size(), want.size());
        }
    }
    printf("cases=%d bad=%d\n", cases, bad);
    return bad != 0 || cases < 3000;
}
"""


def test_fold_kernel_visits_exactly_what_the_backward_kernels_left(tmp_path):
    """csrc/ola_runs.hpp (plain C++, used by the host launch code of backward.hip): the head / tail / border-zone runs the fold
    kernel walks are exactly the samples ola_direct() says the overlap-adding backward kernels did not store themselves — no
    sample missed, none visited twice — over random fft sizes, hops, lengths, paddings and segment lengths.  (On the device
    the gradient tests run with NaN-filled outputs for the same reason.)"""
    src = tmp_path / 'runs.cpp'
    src.write_text(_OLA_RUNS_CHECK)
    exe = str(tmp_path / 'runs')
    subprocess.run(['g++', '-O1', '-std=c++17', '-I', os.path.join(ROOT, 'torchaudio-contrib_amd', 'csrc'), '-o', exe, str(src)],
                   check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith('bad=0'), out.stdout


def test_ring_loader_assembly_has_no_compiler_inserted_memory_waits(tmp_path):
    """The hop-ring kernel's loader wave (csrc/stft_ring3.hpp) counts its LDS-DMA loads itself: one s_waitcnt vmcnt(2 (PF - 1)) per
    hop, vmcnt(0) at the flush and at the end.  The compiler's wait-count pass tracks the address registers of
    __builtin_amdgcn_global_load_lds and inserts vmcnt(0) wherever the register allocator reuses one of them — in front of the marks
    poll or of the next load, which serialises the loader on the memory latency (+106 % on the whole kernel, measured in round 5 on a
    source change that only touched the loader's bookkeeping: tools/ablation/README.md).  Whether that happens depends on register
    allocation, so the build is checked: inside the loader's loops of every instantiation exactly the three expected waits."""
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    out = tmp_path / 'stft_kernels.s'
    csrc = os.path.join(ROOT, 'torchaudio-contrib_amd', 'csrc')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-slp-vectorize', '--cuda-device-only', '-S',
                    '-w', os.path.join(csrc, 'stft_kernels.hip'), '-o', str(out)], check=True, cwd=csrc)
    text = out.read_text()
    import re
    seen = 0
    for m in re.finditer(r'\n(_ZN3tac17stft_ring3_kernelILi1024ELi16ELi(\d)ELi12ELi(\d)E\w*):', text):
        hpf = int(m.group(3))
        body = text[m.end():text.index('.Lfunc_end', m.end())]
        loader = body[body.index('s_setprio 3'):]
        loops = loader[loader.index('\n.LBB'):]                     # (the straight-line prologue may wait for the set-up loads)
        waits = re.findall(r's_waitcnt vmcnt\((\d+)\)', loops)
        lph, ring = (2, 21) if hpf == 4 else (1, 43)
        pf = min(24, ring - 12 - (hpf - 1))
        assert waits == ['0', str(lph * (pf - 1)), '0'], (m.group(1), waits)
        assert loops.count('global_load_lds_dwordx4') == lph
        seen += 1
    assert seen == 10                                               # five row modes x two hops per frame


def test_chain_end_detection(tac, monkeypatch):
    """_lazy.ends_chain: a deferring layer is told that it is the LAST stage of the user's nn.Sequential (through nesting) — then it
    hands back an ordinary tensor (reference layers are eager: layers.py:84-102) — and only then.  The compiled walk
    (csrc/binding/tac_ext.cpp chain_end) and the Python walk must agree."""
    import torch
    from torchaudio_contrib_amd import _lazy
    seen = []

    class Probe(torch.nn.Module):
        def forward(self, x):
            seen.append(bool(_lazy.ends_chain(self)))
            return x

    class Factory(torch.nn.Sequential):
        _tac_realizes = True

    class Wrapper(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, x):
            return self.inner(x)

    a, b, c = Probe(), Probe(), Probe()
    x = torch.zeros(1)

    def run(m):
        del seen[:]
        m(x)
        return list(seen)

    def check():
        S = torch.nn.Sequential
        assert run(S(a, b, c)) == [False, False, True]
        assert run(S(S(a, b), c)) == [False, False, True]
        assert run(S(c, S(a, b))) == [False, False, True]                 # call order c, a, b: b ends the chain through the nesting
        assert run(S(S(S(a)))) == [True]
        assert run(a) == [False]                                          # a direct call keeps deferring (tests/test_layers.py:98-101 style)
        assert run(Wrapper(S(a, b))) == [False, True]                     # a user module around the container: its output is an end
        assert run(Factory(a, b)) == [False, False] and run(S(Factory(a, b))) == [False, False]   # the factories' container realises by itself
        hook = b.register_forward_hook(lambda m, i, o: None)              # (nn.Module._call_impl's slow path: more frames in between)
        try:
            assert run(S(a, b)) == [False, True]
        finally:
            hook.remove()

    check()
    compiled = tac._native.binding() == 'compiled' and getattr(tac._native.ext(), 'chain_end_supported', False)
    if compiled:
        assert _lazy.ends_chain is tac._native.ext().ends_chain          # the layers now call the compiled walk directly
        import importlib, types
        # ... and the Python walk gives the same answers: a fresh copy of the function, with the hand-over switched off
        src = importlib.util.find_spec('torchaudio_contrib_amd._lazy').origin
        text = open(src).read()
        start = text.index('def ends_chain(module):')
        end = text.index('class PlannedChain')
        ns = {'sys': sys, '_SEQ_FORWARD_CODE': _lazy._SEQ_FORWARD_CODE, '_MODULE_PY': _lazy._MODULE_PY, '_ends_chain_resolved': True}
        exec(compile(text[start:end], src, 'exec'), ns)
        monkeypatch.setattr(_lazy, 'ends_chain', ns['ends_chain'])
        check()


def test_planned_chain_falls_back_on_cpu(tac):
    """tac.planned(model, example): on a CPU example there is nothing to bind — the callable is the model (BASELINE configs[0] is a
    CPU configuration), same values, an ordinary tensor."""
    import torch
    model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=32, sample_rate=16000, fft_length=512, hop_length=256), tac.AmplitudeToDb())
    x = torch.from_numpy(np.random.RandomState(0).randn(2, 1, 4000).astype(np.float32))
    fast = tac.planned(model, x)
    assert isinstance(fast, tac.PlannedChain) and not fast.fused()
    y = fast(x)
    assert type(y) is torch.Tensor and torch.equal(y, model(x))
    assert type(torch.nn.Sequential(*tac.Melspectrogram(num_mels=32, sample_rate=16000, fft_length=512, hop_length=256))(x)) is torch.Tensor
