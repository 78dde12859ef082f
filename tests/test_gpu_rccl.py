"""RCCL on the GPU box: the 1-GPU boxes the `-m gpu` suite runs on cannot form a group of eight, but they
can load RCCL, build a communicator and run the very collective the N > 1 path issues — on the strided output of the
BASELINE configs[2] per-GPU shard — and `bench.py --gpus 1` can run under the driver's torchrun launch line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torchaudio_contrib_amd as tac
from torchaudio_contrib_amd.distributed import all_gather_batch, ShardedPipeline
tac.set_strict(True)
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == 'nccl'
# BASELINE configs[2], one GPU's shard: 256 rows x 1 323 000 samples (44.1 kHz x 30 s), 2048 / 512, 128 mel + dB
rows, length = 256, 44100 * 30
frames = 1 + length // 512
x = torch.rand(rows, 1, length, device=dev) * 2 - 1
model = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=44100, fft_length=2048, hop_length=512),
                            tac.AmplitudeToDb()).to(dev)
y = tac.realize(model(x))
assert tuple(y.shape) == (rows, 1, 128, frames) and y.stride(-1) == 128 and y.stride(-2) == 1     # frame-major physical layout
calls = []
real = dist.all_gather_into_tensor
dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
assert all_gather_batch(y) is y and not calls                                   # a group of one returns early by default
g = all_gather_batch(y, total_rows=rows, force_collective=True)                   # ... the forced form runs RCCL on 338.7 MB
torch.cuda.synchronize()
assert len(calls) == 1 and g.data_ptr() != y.data_ptr()
assert g.shape == y.shape and g.stride() == y.stride() and torch.equal(g, y)
g2 = all_gather_batch(y, total_rows=rows, force_collective=True, method='p2p')  # (no peers: the local copy of the direct exchange)
assert torch.equal(g2, y)
# the wrapper a caller would use, gather included
pipe = ShardedPipeline(model, gather=True)
assert torch.equal(pipe(x), y)
# ... and its overlapped form: the shard computed in 8 row pieces, every piece's collective (async, RCCL's stream) posted while the
# next piece is computed, staged pieces moved to their place on a side stream — same tensor, same layout
n0 = len(calls)
for method in ('rccl', 'p2p'):
    yo = ShardedPipeline(model, gather=True, overlap=True, chunks=8, method=method, force_collective=True)(x)
    torch.cuda.synchronize()
    assert yo.shape == y.shape and yo.stride() == y.stride() and torch.equal(yo, y), method
assert len(calls) == n0 + 8, 'eight chunk collectives went through RCCL'
t0 = time.perf_counter()
for _ in range(5):
    ShardedPipeline(model, gather=True, overlap=True, chunks=8, force_collective=True)(x)
torch.cuda.synchronize()
print('RCCL_OVERLAPPED_STEP_MS %%.3f (compute + 8 chunk gathers at world 1)' %% ((time.perf_counter() - t0) / 5 * 1e3))
# timing of the forced collective (device-local at world 1: what RCCL's launch + copy costs without a wire)
for _ in range(3):
    all_gather_batch(y, total_rows=rows, force_collective=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    all_gather_batch(y, total_rows=rows, force_collective=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if ('rccl' in l.lower() or 'nccl' in l.lower()) and l.split()[-1].startswith('/')})
print('RCCL_LIBS', libs)
print('RCCL_SELF_GATHER_MS %%.3f for %%.1f MB' %% (ms, y.numel() * 4 / 1e6))
assert libs, 'no RCCL / NCCL library mapped into the process'
dist.destroy_process_group()
print('ok')
'''


def _port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return str(s.getsockname()[1])


def test_rccl_world1_forced_allgather_on_cfg3_shard(tmp_path):
    script = tmp_path / 'rccl_worker.py'
    script.write_text(_WORKER % {'root': ROOT})
    out = subprocess.run([sys.executable, str(script), _port()], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600)
    print(out.stdout[-3000:])
    # (RCCL prints its version banner when the communicator is torn down: 'ok' is not the last line)
    assert out.returncode == 0 and '\nok\n' in out.stdout, out.stdout[-3000:]
    assert 'RCCL_LIBS' in out.stdout and 'librccl' in out.stdout


def test_bench_under_torchrun_one_rank():
    """The driver's N > 1 launch line at N = 1: torchrun rendezvous on 127.0.0.1, the `nccl` process group initialised,
    barrier-bracketed timing, and the all-gather leg measured through RCCL (forced: a group of one would return early)."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', _port(), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '5',
           '--warmup', '2', '--repeats', '3', '--no-cpu-baseline', '--no-stages']
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 1 and rec['steps'] == 5 and rec['value'] > 1e8
    assert rec['process_group'] == 'nccl'
    gather = rec['with_allgather']
    assert 'error' not in gather and gather['value'] > 0 and gather['forced_at_world_1'] is True
