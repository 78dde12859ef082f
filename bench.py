#!/usr/bin/env python
"""bench.py — mel frames/sec of Sequential(*Melspectrogram(128 mel, 2048/512), AmplitudeToDb())
on N MI355X (BASELINE.json metric; workload = configs[1]: 256 x 1ch x 16 kHz x 10 s per GPU).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the fused hot path over one device-resident synthetic batch per rank
(weak scaling: per-GPU work fixed; ranks shard the batch axis, no data-path collective).  Rank 0
prints ONE JSON line.  The K-step timed region (barrier + synchronize on both sides, max over ranks) is
repeated `--repeats` times back to back and the MEDIAN region decides `value` / `ms_per_step`
(`value_p10` / `value_p90` beside it), so the headline is not one 2 ms window.  `roofline` is measured
live with HIP events on the launch stream around the dominant (only) kernel, names the kernel the
library actually launched (tac_last_route), the shader clock that kernel ran at (tac_debug_clock_probe)
and what binds it according to the committed counter passes; `cpu_baseline` times the CPU oracle
(torch-CPU restatement of the reference's op sequence) on a bounded sample of the same workload on this
box's host cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, N_FFT, HOP, N_MELS = 16000, 2048, 512, 128
BATCH, CHANNELS, SECONDS = 256, 1, 10
LENGTH = SR * SECONDS
FRAMES = 1 + LENGTH // HOP                         # 313
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# BASELINE configs[2] per rank: 2048 x 1ch x 44.1 kHz x 30 s batch-sharded over 8 GPUs = 256 rows of 1 323 000 samples
CFG3_SR, CFG3_LENGTH = 44100, 44100 * 30
# The timed steps walk over NBUF distinct device-resident input batches (NBUF x 163.8 MB = 655 MB > the 256 MiB
# Infinity Cache), so a step's samples cannot have stayed on-die from the step before: the headline and the stage
# figures are HBM numbers.  The same loop over ONE batch (which fits the Infinity Cache) is reported beside them.
NBUF = 4
GATHER_CHUNKS = 8                                   # row pieces of the overlapped all-gather leg


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=25,
                    help='back-to-back repetitions of the K-step timed region; the median region is reported')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stages', action='store_true')
    ap.add_argument('--no-live-counters', action='store_true',
                    help='do not run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic in this '
                         'run; the committed counter pass of the same kernel is quoted instead')
    ap.add_argument('--config', choices=('cfg2', 'cfg3'), default='cfg2',
                    help='cfg2 (default): BASELINE configs[1], 256 x 16 kHz x 10 s per GPU — the configuration the metric '
                         'is quoted on, at every N (weak scaling).  cfg3: configs[2]\'s per-GPU shard, 256 x 44.1 kHz x '
                         '30 s, as the headline.  With --gpus > 1 and cfg2 the cfg-3 shard is measured as an extra '
                         '"cfg3" object (compute only and with its 338.7 MB/rank all-gather).')
    ap.add_argument('--dry-run-cpu', action='store_true',
                    help='control-flow check without a GPU: gloo backend, CPU tensors, a tiny batch (tests/ use it to '
                         'cover the --gpus N spawn / reduce / JSON path); the printed line is marked invalid')
    return ap.parse_args()


def event_ms(fn, iters):
    """GPU duration of fn() (one kernel launch) from per-launch HIP event pairs recorded on the stream the kernel is
    launched on (torch's current stream == the stream passed through the C ABI): (mean of the fastest 90 %, median)."""
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        starts[i].record()
        fn()
        ends[i].record()
    torch.cuda.synchronize()
    times = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    keep = times[:max(1, (len(times) * 9) // 10)]          # drop the slowest 10 %: launches that caught a clock ramp / preemption
    return sum(keep) / len(keep), times[len(times) // 2]


def cpu_baseline(x):
    """Time the CPU oracle (torch-CPU restatement of the reference op sequence) on the IDENTICAL host tensor the GPU
    run uses: all 256 rows with the best thread count, and a 1-thread figure on a slice of it.
    The host may expose far more hardware threads than torch's CPU kernels scale to, so a few thread counts are tried
    briefly and the best one is reported (cores = threads actually used)."""
    from oracle import torch_ref
    avail = os.cpu_count() or 1
    rows = x.shape[0]

    def run(t):
        return torch_ref.melspectrogram_db(t, n_fft=N_FFT, hop=HOP, num_mels=N_MELS, sample_rate=SR)

    best, best_threads, reps_total = float('inf'), 1, 0
    t_all = time.perf_counter()
    for threads in sorted({8, 16, 32, 64, avail}, reverse=True):
        if threads > avail or time.perf_counter() - t_all > 15.0:
            continue
        torch.set_num_threads(threads)
        run(x)
        for _ in range(3):
            t0 = time.perf_counter()
            run(x)
            dt = time.perf_counter() - t0
            reps_total += 1
            if dt < best:
                best, best_threads = dt, threads
    torch.set_num_threads(1)
    one_rows = max(1, rows // 16)
    run(x[:one_rows])
    t0 = time.perf_counter()
    run(x[:one_rows])
    one = time.perf_counter() - t0
    torch.set_num_threads(best_threads)
    frames = x.shape[1] * FRAMES
    return {'value': rows * frames / best, 'unit': 'frames/s', 'cores': best_threads, 'kind': 'port',
            'single_thread_value': one_rows * frames / one,
            'sample': 'the identical %dx%dx%d f32 uniform(-1,1) host tensor the GPU run was fed (all %d rows; %d/%d/%d '
                      'mel + dB), best of %d runs over thread counts up to %d (best at %d threads); single_thread_value '
                      'from the first %d rows at 1 thread; torch %s CPU ops in the reference op order '
                      '(oracle/torch_ref.py)'
                      % (rows, x.shape[1], x.shape[2], rows, N_FFT, HOP, N_MELS, reps_total, avail, best_threads,
                         one_rows, torch.__version__)}


N_SIMD = 256 * 4                                    # MI355X: 256 CUs x 4 SIMDs
F32_VECTOR_PEAK_TFLOPS = 157.3                      # MI355X_MICROARCH.md: packed f32 vector peak


def bound_from_counters(roof, route, clock_mhz, live=None, live_note=None):
    """Fill roofline.traffic / valu_floor_ms / lds_busy / bound from rocprofv3 counter passes of THIS kernel: `live` (round 6: the
    passes live_counters() ran as child processes of this very run) or, without it, the committed ones (profiles/rNN/pmc_mel.json;
    instruction counts and busy cycles per launch are properties of the kernel binary and the workload, the clock is measured live).
    The entry must carry the name of the kernel the library just launched; otherwise the fields stay null and say why."""
    entry, source = None, None
    if live is not None:
        roof['counters_committed_pass'] = {k: roof.get(k) for k in ('traffic', 'valu_floor_ms', 'lds_busy', 'counters_source')}
        roof['counters_source'] = live_note
        entry = live
    for rnd in (() if live is not None else ('r06', 'r05', 'r04', 'r03')):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', rnd, 'pmc_mel.json')))
        except Exception:            # noqa: BLE001
            continue
        for kname, d in pmc.items():
            if route and route in kname:
                entry, source = d, 'profiles/%s/pmc_mel.json' % rnd
                break
        if entry:
            break
    if entry is None:
        roof['counters_note'] = 'no committed counter pass names the launched kernel (%s): traffic / bound not derived' % route
        return
    if live is None:
        roof['counters_source'] = source + ' (rocprofv3 --pmc, separate passes; FETCH_SIZE x2 gfx950 correction applied)'
    if 'hbm_traffic_bytes_per_launch' in entry:
        roof['traffic'] = entry['hbm_traffic_bytes_per_launch']
    busy = {}
    kernel_cycles = None
    if 'SQ_BUSY_CU_CYCLES' in entry:
        kernel_cycles = entry['SQ_BUSY_CU_CYCLES'] / 256.0                     # per CU
    if 'SQ_ACTIVE_INST_VALU' in entry:
        valu_cycles = entry['SQ_ACTIVE_INST_VALU'] * 4.0 / N_SIMD               # counted in quad-cycles over all SIMDs
        roof['valu_busy_cycles_per_simd'] = valu_cycles
        if clock_mhz:
            roof['valu_floor_ms'] = valu_cycles / (clock_mhz * 1e3)
            roof['frac_of_valu_floor'] = roof['valu_floor_ms'] / roof['kernel_ms_mean']
            busy['valu'] = roof['frac_of_valu_floor']
        elif kernel_cycles:
            busy['valu'] = valu_cycles / kernel_cycles
    if 'SQ_LDS_IDX_ACTIVE' in entry and kernel_cycles:
        roof['lds_busy'] = entry['SQ_LDS_IDX_ACTIVE'] / 256.0 / kernel_cycles
        busy['lds'] = roof['lds_busy']
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in entry and kernel_cycles:
        busy['mfma'] = entry['SQ_VALU_MFMA_BUSY_CYCLES'] / N_SIMD / kernel_cycles
    busy['hbm'] = roof['frac']
    roof['busy_fractions'] = busy
    roof['bound'] = max(busy, key=busy.get)
    roof['bound_note'] = ('the busiest resource by the counters; `frac` / `peak` stay the HBM figure the contract asks for '
                          '(algorithmic bytes / kernel time / 8 TB/s)')


def live_counters(route, timeout=75):
    """The counter entry of the launched kernel measured in THIS run: three child processes `rocprofv3 --pmc ...` (separate passes, as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE; WRITE_SIZE; the SQ busy counters) around `tools/prof_driver.py mel 3` — the same kernel
    on one cfg-2 batch — per-launch means of the kernel's rows, HBM bytes with the guide's gfx950 corrections (FETCH_SIZE in KB, x2 for
    wide coalesced reads; WRITE_SIZE in KB).  Returns (entry, note) or (None, why).  Never raises; the timed regions are over when it runs."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None, 'rocprofv3 not on PATH'
    tmp = tempfile.mkdtemp(prefix='tac_bench_pmc_')
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    entry = {}
    try:
        for tag, counters in (('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE']),
                              ('sq', ['SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CU_CYCLES', 'SQ_LDS_IDX_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES'])):
            out = os.path.join(tmp, tag)
            cmd = [exe, '--pmc'] + counters + ['--output-format', 'csv', '-d', out, '-o', 'p', '--', sys.executable,
                                               os.path.join(ROOT, 'tools', 'prof_driver.py'), 'mel', '3']
            proc = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
            vals = {}
            for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name') in counters and route and route in r.get('Kernel_Name', ''):
                        vals.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            if tag != 'sq' and not vals:
                return None, 'rocprofv3 --pmc %s (exit %d) produced no row for %s' % (counters[0], proc.returncode, route)
            for c, v in vals.items():
                entry[c] = sum(v) / len(v)
        entry['hbm_read_bytes_corrected'] = entry['FETCH_SIZE'] * 2048.0
        entry['hbm_write_bytes'] = entry['WRITE_SIZE'] * 1024.0
        entry['hbm_traffic_bytes_per_launch'] = entry['hbm_read_bytes_corrected'] + entry['hbm_write_bytes']
        return entry, ('measured in this run: rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ busy counters) as separate child processes '
                       'around tools/prof_driver.py mel 3 (the same kernel on one cfg-2 batch), per-launch means, FETCH_SIZE x2 gfx950 '
                       'correction applied; read %.1f MB + written %.1f MB'
                       % (entry['hbm_read_bytes_corrected'] / 1e6, entry['hbm_write_bytes'] / 1e6))
    except Exception as exc:            # noqa: BLE001
        return None, '%s: %s' % (type(exc).__name__, exc)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def rccl_debug_lines(path, limit=16):
    """What RCCL said about itself (NCCL_DEBUG=INFO to a file): the channel / ring summary of the communicator and the
    distinct "<bytes> Bytes -> Algo .. proto .." decisions, prefix stripped, first `limit` of each kind."""
    import re
    try:
        text = open(path, errors='replace').read().splitlines()
    except OSError as exc:
        return {'error': str(exc)}
    strip = lambda l: re.sub(r'^.*?NCCL INFO ', '', l).strip()
    algo, topo, seen = [], [], set()
    for line in text:
        msg = strip(line)
        if 'Bytes -> Algo' in msg or 'Algo' in msg and 'proto' in msg.lower():
            key = re.sub(r'time [0-9.e+-]+', '', msg)
            if key not in seen and len(algo) < limit:
                seen.add(key)
                algo.append(msg)
        elif re.search(r'coll channels|p2p channels|Ring 0+ :|Trees? \[|NCCL version|RCCL version|nranks', msg):
            if msg not in seen and len(topo) < limit:
                seen.add(msg)
                topo.append(msg)
    return {'source': 'NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,TUNING,GRAPH of rank 0 (%d lines)' % len(text),
            'algo_proto': algo, 'communicator': topo}


def extra_stages(stages, tac, dev, gen, model):
    """Driver-timed figures for BASELINE configs[3] (multi-channel STFT + ComplexNorm at 4096 / 1024, full size: 17.7 GB of
    algorithmic traffic in ONE launch), configs[4] (mu-law encode / decode, 12 B per sample each way) and the host-bound end of
    the headline chain (1 / 4 / 16 rows: microseconds per call).  Each guarded by a free-memory check and its own try."""
    def free_gb():
        return torch.cuda.mem_get_info(dev)[0] / 1e9

    try:        # configs[3]: 64 x 8ch x 48 kHz x 60 s, fft_length 4096, hop 1024, magnitude (reference layers.py:267-304)
        b4, c4, l4, n4, h4 = 64, 8, 48000 * 60, 4096, 1024
        t4, f4 = 1 + l4 // h4, n4 // 2 + 1
        need = (b4 * c4 * l4 * 4 + b4 * c4 * t4 * f4 * 4) / 1e9 * 1.15
        if free_gb() < need + 2:
            stages['cfg4_spectrogram_4096'] = {'skipped': 'needs %.1f GB of device memory, %.1f free' % (need, free_gb())}
        else:
            torch.cuda.empty_cache()
            x4 = torch.rand(b4, c4, l4, device=dev, generator=gen) * 2 - 1
            spec4 = tac.Spectrogram(n4, h4, power=1.).to(dev)
            fn = lambda: tac.realize(spec4(x4))
            for _ in range(3):
                fn()
            ms, med = event_ms(fn, 12)
            per_frame = 4 * h4 + 4 * f4
            alg = b4 * c4 * t4 * per_frame
            stages['cfg4_spectrogram_4096'] = {
                'workload': 'Spectrogram(4096, 1024, power=1) on %d x %dch x %d samples (BASELINE configs[3], full size, one launch)' % (b4, c4, l4),
                'kernel': 'stft_n4096_s3_kernel<2, 12, false> (csrc/stft_n4096_s3.hpp: twelve waves per CU, both 1024-point transforms through one exchange area), |X| rows',
                'frames': b4 * c4 * t4, 'kernel_ms_mean': ms, 'kernel_ms_median': med, 'alg_bytes_per_frame': per_frame,
                'alg_bytes_per_launch': alg, 'achieved_GBs': alg / (ms * 1e-3) / 1e9,
                'frac_of_hbm_peak': alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'frames_per_s': b4 * c4 * t4 / (ms * 1e-3)}
            del x4, spec4, fn
            torch.cuda.empty_cache()
    except Exception as exc:            # noqa: BLE001
        stages['cfg4_spectrogram_4096'] = {'error': '%s: %s' % (type(exc).__name__, exc)}

    try:        # the Melspectrogram chain at fft_length 4096 (reference layers.py:307-381): ONE launch since round 6
        x4m = torch.rand(8, 8, 480000, device=dev, generator=gen) * 2 - 1
        mel4 = torch.nn.Sequential(*tac.Melspectrogram(num_mels=128, sample_rate=48000, fft_length=4096, hop_length=1024),
                                   tac.AmplitudeToDb()).to(dev)
        fn = lambda: mel4(x4m)
        spin(fn, 0.2)
        before = dict(tac._hip.launches)
        fn()
        calls = {k: v - before.get(k, 0) for k, v in tac._hip.launches.items() if v != before.get(k, 0)}
        ms, med = event_ms(fn, 50)
        frames = 64 * (1 + 480000 // 1024)
        stages['mel4096_one_launch'] = {
            'workload': 'Sequential(*Melspectrogram(128, 48 kHz, 4096 / 1024), AmplitudeToDb()) on 8 x 8ch x 480 000 samples',
            'kernel': 'stft_n4096_s3_kernel<1, W, true> (the twelve-wave form of the 4096 rows + band-sparse contraction + dB)',
            'launches_per_call': calls, 'frames': frames, 'kernel_ms_mean': ms, 'kernel_ms_median': med,
            'frames_per_s': frames / (ms * 1e-3),
            'note': 'rounds 2 - 5: two launches (spectrogram rows through HBM + streaming filterbank), 0.163 - 0.166 ms'}
        del x4m, mel4, fn
    except Exception as exc:            # noqa: BLE001
        stages['mel4096_one_launch'] = {'error': '%s: %s' % (type(exc).__name__, exc)}

    try:        # configs[4]: 1024 x 1ch x 24 kHz x 5 s, 256 levels (reference functional.py:317-354): f32 -> int64 -> f32
        x5 = torch.rand(1024, 1, 24000 * 5, device=dev, generator=gen) * 2 - 1
        enc, dec = tac.MuLawEncoding(256).to(dev), tac.MuLawDecoding(256).to(dev)
        codes = enc(x5)
        assert codes.dtype == torch.int64
        for name, fn in (('cfg5_mulaw_encode', lambda: enc(x5)), ('cfg5_mulaw_decode', lambda: tac.realize(dec(codes)))):     # (MuLawDecoding defers: realize launches it)
            spin(fn, 0.2)
            ms, med = event_ms(fn, 50)
            alg = x5.numel() * 12
            stages[name] = {'workload': '1024 x 1ch x 120 000 samples, 256 levels (BASELINE configs[4])', 'kernel_ms_mean': ms,
                            'kernel_ms_median': med, 'alg_bytes_per_sample': 12, 'alg_bytes_per_launch': alg,
                            'achieved_GBs': alg / (ms * 1e-3) / 1e9, 'frac_of_hbm_peak': alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            'samples_per_s': x5.numel() / (ms * 1e-3)}
        del x5, codes
    except Exception as exc:            # noqa: BLE001
        stages['cfg5_mulaw'] = {'error': '%s: %s' % (type(exc).__name__, exc)}

    try:        # the host-bound end of the headline chain: wall time per call, launches back to back, 1 / 4 / 16 rows of 10 s
        small = {}
        for rows in (1, 4, 16):
            xr = torch.rand(rows, CHANNELS, LENGTH, device=dev, generator=gen) * 2 - 1
            fn = lambda: model(xr)
            spin(fn, 0.2)
            torch.cuda.synchronize()
            n = 2000
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            t_host = time.perf_counter() - t0            # launches issued (the queue may still hold some)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            small['rows_%d' % rows] = {'us_per_call_host': t_host / n * 1e6, 'us_per_call': t_all / n * 1e6,
                                        'frames_per_s': rows * CHANNELS * FRAMES * n / t_all}
            # the same chain bound to one call (tac.planned, round 6): what a caller of many small batches uses
            fast = tac.planned(model, xr)
            spin(lambda: fast(xr), 0.1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fast(xr)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            small['rows_%d' % rows]['planned'] = {'fused': bool(fast.fused()), 'us_per_call_host': t_host / n * 1e6,
                                                   'us_per_call': t_all / n * 1e6, 'frames_per_s': rows * CHANNELS * FRAMES * n / t_all}
        small['note'] = ('Sequential(*Melspectrogram, AmplitudeToDb) on rows x 1 x 160 000 samples, 2000 calls back to back: '
                         'us_per_call_host = Python + dispatch + launch per call (the floor for tiny batches), us_per_call = '
                         'including the GPU draining the queue; planned = the same chain through tac.planned(model, x): one bound call')
        stages['small_batch'] = small
    except Exception as exc:            # noqa: BLE001
        stages['small_batch'] = {'error': '%s: %s' % (type(exc).__name__, exc)}


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def _spawned(local_rank, a, port):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(a.gpus),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    run(a)


def main():
    a = parse()
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher — one process per GPU on this node, as torchrun would
        if not a.dry_run_cpu and torch.cuda.device_count() < a.gpus:
            raise SystemExit('bench.py: --gpus %d requested but only %d GPU(s) visible' % (a.gpus, torch.cuda.device_count()))
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        import torch.multiprocessing as mp
        mp.spawn(_spawned, args=(a, _free_port()), nprocs=a.gpus, join=True)
        return
    run(a)


def spin(fn, seconds):
    """bring the GPU out of its idle clocks (not timed, fixed wall budget)"""
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()


def rotating(make, xs):
    """fn() that runs make(x) on the next buffer of xs each call"""
    state = [0]

    def fn():
        i = state[0]
        state[0] = (i + 1) % len(xs)
        return make(xs[i])
    return fn


def run(a):
    global BATCH, LENGTH, FRAMES
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d — launch one rank per GPU' % (a.gpus, world))
    # one process per GPU over RCCL whenever there is a rendezvous to join: N > 1 always, and N = 1 when launched through
    # torchrun (the driver's launch line at N = 1 then exercises the same process-group / barrier / collective code)
    distributed = world > 1 or ('MASTER_ADDR' in os.environ and 'RANK' in os.environ)
    import torch.distributed as dist
    if a.dry_run_cpu:
        BATCH, LENGTH = 4, 8192
        FRAMES = 1 + LENGTH // HOP
        return dry_run_cpu(a, rank, world)
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit('bench.py: rank %d has no GPU (%d visible)' % (local_rank, torch.cuda.device_count()))
    rccl_log = None
    if distributed:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if 'NCCL_DEBUG' not in os.environ:
            # make the run self-explaining: RCCL's own account of the rings / channels it built and of the algorithm and protocol
            # it picks per message size goes to a per-process file; rank 0 quotes it in the JSON line (`rccl_debug`)
            rccl_log = '/tmp/tac_rccl_%d.log' % os.getpid()
            os.environ.update(NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,TUNING,GRAPH', NCCL_DEBUG_FILE=rccl_log)
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    import torchaudio_contrib_amd as tac
    tac._native.lib()                                   # fail loudly without the HIP library

    sr, length, seconds, label = SR, LENGTH, SECONDS, 'BASELINE configs[1]'
    if a.config == 'cfg3':
        sr, length, seconds, label = CFG3_SR, CFG3_LENGTH, 30, 'BASELINE configs[2], one GPU\'s shard'
    frames = 1 + length // HOP
    nbuf = NBUF if a.config == 'cfg2' else 1            # (one cfg-3 shard is 1.35 GB: far beyond the Infinity Cache)

    gen = torch.Generator(device=dev).manual_seed(rank)
    x_host = torch.rand(BATCH, CHANNELS, length, generator=torch.Generator().manual_seed(rank)) * 2 - 1
    xs = [x_host.to(dev)]                       # generated once on the host: the CPU baseline times the same tensor
    for _ in range(1, nbuf):
        xs.append(torch.rand(BATCH, CHANNELS, length, device=dev, generator=gen) * 2 - 1)
    x = xs[0]
    if rank != 0 or a.no_cpu_baseline or a.config != 'cfg2':
        x_host = None

    def pipeline(rate):
        return torch.nn.Sequential(
            *tac.Melspectrogram(num_mels=N_MELS, sample_rate=rate, fft_length=N_FFT, hop_length=HOP),
            tac.AmplitudeToDb()).to(dev)

    model = pipeline(sr)
    step = rotating(model, xs)                 # the reference's call site, nothing else (returns an ordinary tensor)

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(fn, steps):
        """(wall seconds of `steps` calls between barriers — max over ranks —, HIP-event ms per call, last result):
        HIP events on the launch stream bracket the same region: a step is ONE kernel launch and the launches are back
        to back, so region time / K is that kernel's average launch duration over the timed region"""
        sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            y = fn()
        ev1.record()
        sync()
        elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, ev0.elapsed_time(ev1) / steps, y

    def quantile(vals, q):
        v = sorted(vals)
        return v[min(len(v) - 1, max(0, int(round(q * (len(v) - 1)))))]

    def repeated(fn, steps, repeats):
        """`repeats` timed regions of `steps` calls, back to back: (list of wall seconds, list of HIP-event ms per call, last y)"""
        walls, evs, y = [], [], None
        for _ in range(max(1, repeats)):
            e, ms, y = timed(fn, steps)
            walls.append(e)
            evs.append(ms)
        return walls, evs, y

    spin(step, 0.5)
    for _ in range(a.warmup):
        y = step()
    walls, region_mss, y = repeated(step, a.steps, a.repeats)
    assert type(y) is torch.Tensor and tuple(y.shape) == (BATCH, CHANNELS, N_MELS, frames)
    frames_per_step = world * BATCH * CHANNELS * frames
    elapsed = quantile(walls, 0.5)                                       # the median region
    value = frames_per_step * a.steps / elapsed
    region_ms = quantile(region_mss, 0.5)
    route = tac._native.lib().tac_last_route().decode()                  # the kernel instantiation the launches took

    # ---- the shader clock the kernel runs at, measured by the kernel itself: every workgroup's wave 0 records the shader
    #      cycles and the 100 MHz ticks of its frame loop (tac_debug_clock_probe) during K more steps
    clock_mhz = None
    try:
        probe = torch.zeros(2 * 1024, dtype=torch.int64, device=dev)
        lib = tac._native.lib()
        tac._native.check(lib.tac_debug_clock_probe(ctypes.c_void_p(probe.data_ptr()), 1024), 'tac_debug_clock_probe')
        ratios = []
        for _ in range(max(3, min(a.steps, 20))):
            step()
            torch.cuda.synchronize()
            pr = probe.view(-1, 2).cpu()
            ok = pr[:, 1] > 0
            if bool(ok.any()):
                ratios.append(float((pr[ok, 0].double() / pr[ok, 1].double()).median()) * 100.0)
            probe.zero_()
        lib.tac_debug_clock_probe(None, 0)
        if ratios:
            clock_mhz = quantile(ratios, 0.5)
    except Exception:            # noqa: BLE001 — a diagnostic field
        clock_mhz = None

    # ---- roofline of the dominant kernel (the fused melspec kernel is the only launch in a step)
    per_launch_mean_ms, med_ms = event_ms(step, min(a.steps, 50))      # (event pairs around single launches: + ~3 us each)
    mean_ms = region_ms
    alg_bytes = BATCH * CHANNELS * frames * (4 * HOP + 4 * N_MELS)      # 2560 B/frame
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
    result = {
        'metric': 'mel frames/sec', 'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'repeats': len(walls), 'value_p10': frames_per_step * a.steps / quantile(walls, 0.9),
        'value_p90': frames_per_step * a.steps / quantile(walls, 0.1),
        'value_note': 'median of `repeats` back-to-back timed regions of `steps` steps each (barrier + synchronize on both '
                      'sides of every region, max over ranks); p10 / p90 over the same regions',
        'process_group': 'nccl' if distributed else None,
        'config': {'workload': 'Melspectrogram+AmplitudeToDb batch=%d/GPU x %dch x %dHz x %ds, fft_len=%d hop=%d '
                               '%d mel (%s)' % (BATCH, CHANNELS, sr, seconds, N_FFT, HOP, N_MELS, label),
                   'global_batch': world * BATCH, 'frames_per_step': frames_per_step,
                   'parallelism': 'batch-sharded x%d, no data-path collective' % world,
                   'input_buffers': '%d distinct device-resident batches of %.1f MB visited round-robin (%.0f MB > the '
                                    '256 MiB Infinity Cache)' % (nbuf, x.numel() * 4 / 1e6, nbuf * x.numel() * 4 / 1e6)},
        'roofline': {'kernel': '%s (fused STFT + power + band-sparse mel + dB, one launch per step)' % route,
                     'kernel_route': route, 'bound': 'hbm',
                     'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': None, 'alg_bytes_per_launch': alg_bytes, 'kernel_ms_mean': mean_ms,
                     'kernel_ms_p10': quantile(region_mss, 0.1), 'kernel_ms_p90': quantile(region_mss, 0.9),
                     'kernel_ms_median': med_ms, 'kernel_ms_per_launch_events': per_launch_mean_ms,
                     'shader_clock_mhz': clock_mhz,
                     'median_vs_mean_note': 'kernel_ms_mean is a K-launch region / K (launches back to back: no gap, no event '
                                            'overhead); kernel_ms_median / kernel_ms_per_launch_events bracket EVERY launch with its '
                                            'own event pair, which adds ~3 us of event processing per launch and lets the '
                                            'queue drain between launches — it reads higher and is kept only as a cross-check',
                     'timing': 'median over `repeats` regions of: one HIP event pair on the launch stream around the K timed '
                               'steps (one launch per step) / K; kernel_ms_per_launch_events / kernel_ms_median: event '
                               'pairs around single launches; shader_clock_mhz: shader cycles / 100 MHz ticks of every '
                               'workgroup\'s frame loop, recorded by the kernel (tac_debug_clock_probe)'},
    }
    bound_from_counters(result['roofline'], route, clock_mhz)
    if nbuf > 1:
        # the same loop re-reading ONE batch (163.8 MB: it fits the Infinity Cache) — what rounds 1 and 2 reported
        one = lambda: model(x)
        spin(one, 0.2)
        w1, m1, _ = repeated(one, a.steps, max(1, a.repeats // 3))
        e1, ms1 = quantile(w1, 0.5), quantile(m1, 0.5)
        result['single_buffer'] = {'value': frames_per_step * a.steps / e1, 'ms_per_step': e1 / a.steps * 1e3,
                                   'kernel_ms_mean': ms1,
                                   'frac_of_hbm_peak': alg_bytes / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   'note': 'same steps on one re-read input batch (Infinity-Cache resident): the figure '
                                           'rounds 1 and 2 reported as roofline.frac'}

    if rank == 0 and not a.no_stages and a.config == 'cfg2':
        stages = {}
        try:            # secondary figures: a failure here must not cost the headline line
            # secondary stages named by north_star, each one kernel: complex STFT and power spectrogram — on the rotating
            # batches, and on one re-read batch beside it
            spec = tac.Spectrogram(N_FFT, HOP, power=2.).to(dev)
            f_bins = N_FFT // 2 + 1
            for name, make, per_frame in (('stft_complex', lambda t: tac.stft(t, N_FFT, HOP), 4 * HOP + 8 * f_bins),
                                          ('spectrogram_power', spec, 4 * HOP + 4 * f_bins)):
                fn = rotating(make, xs)
                # three rounds of (spin-up, 50 launches with per-launch events): these write-heavy kernels show slow
                # transients after a change of working set (allocator blocks, TLB, clocks) that one round can land in —
                # every round is reported, the stage figure is the median round
                rounds = []
                for _ in range(3):
                    spin(fn, 0.3)
                    rounds.append(event_ms(fn, 50))
                ms, med = sorted(rounds)[1]
                # the same launches timed the way the headline is: ONE event pair around 50 back-to-back launches (no per-launch
                # event processing, no queue drain between launches) — reported beside the per-launch figure, which stays the
                # one `frac_of_hbm_peak` is computed from
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                for _ in range(50):
                    fn()
                ev1.record()
                torch.cuda.synchronize()
                ms_region = ev0.elapsed_time(ev1) / 50
                ms1, _ = event_ms(lambda: make(x), 50)
                gbs = BATCH * CHANNELS * frames * per_frame / (ms * 1e-3) / 1e9
                stages[name] = {'kernel_route': tac._native.lib().tac_last_route().decode() or None, 'kernel_ms_mean': ms, 'kernel_ms_median': med, 'alg_bytes_per_frame': per_frame,
                                'achieved_GBs': gbs, 'frac_of_hbm_peak': gbs / HBM_PEAK_GBS,
                                'kernel_ms_region': ms_region,
                                'frac_of_hbm_peak_region': BATCH * CHANNELS * frames * per_frame / (ms_region * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                'kernel_ms_mean_rounds': [r[0] for r in rounds],        # (in the order they ran)
                                'single_buffer_kernel_ms_mean': ms1}
            # the filterbank stage as a dense fp32 MFMA GEMM (a random 1025 x 128 bank is not band-sparse, so
            # apply_filterbank takes the GEMM kernel): executed flops = 2*F*M per frame against the 157.3 TFLOP/s f32 MFMA peak
            fb_dense = torch.rand(f_bins, N_MELS, device=dev, generator=gen)
            p_spec = spec(x)
            fn = lambda: tac.apply_filterbank(p_spec, fb_dense)
            spin(fn, 0.3)
            ms, med = event_ms(fn, 50)
            flops = 2.0 * f_bins * N_MELS * BATCH * CHANNELS * frames
            stages['filterbank_mfma_dense'] = {'kernel_ms_mean': ms, 'kernel_ms_median': med, 'flops': flops,
                                               'achieved_TFLOPs': flops / (ms * 1e-3) / 1e12,
                                               'frac_of_f32_mfma_peak': flops / (ms * 1e-3) / 1e12 / 157.3}
            del p_spec
            # training step of the same chain (waveform requires grad): fused forward + dB op, then the HIP gradient kernels
            # (dB adjoint, ONE backward kernel: filterbank adjoint + frame re-transform + norm adjoint + inverse FFT + overlap-add,
            # border fold); wall time per step between two events, gradient buffer released each step
            try:
                xg = x.clone().requires_grad_(True)
                ones = torch.ones((BATCH, CHANNELS, N_MELS, frames), device=dev)

                def train():
                    xg.grad = None
                    y = model(xg)
                    y.backward(ones)
                    return y
                spin(train, 0.3)
                ms, med = event_ms(train, 30)
                stages['train_step_fwd_bwd'] = {'ms_mean': ms, 'ms_median': med,
                                                'note': 'forward + backward of Sequential(*Melspectrogram, AmplitudeToDb) at cfg-2, '
                                                        'event pair around each step (kernels + launch gaps of the autograd graph)'}
                del xg, ones
            except Exception as exc:            # noqa: BLE001 — a secondary figure
                stages['train_step_fwd_bwd'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
        except Exception as exc:            # noqa: BLE001 — reported in the line, not swallowed
            stages['error'] = '%s: %s' % (type(exc).__name__, exc)
        # ---- the other BASELINE configs, each at its full size, one launch per call (round 5)
        extra_stages(stages, tac, dev, gen, model)
        result['stages'] = stages

    def gather_leg(mdl, inputs, rows_total, frames_step, steps):
        """compute + the ONE exchange of the (B/N, C, M, T) output shards, every form side by side: `rccl`
        (all_gather_into_tensor) and `p2p` (direct sends to every peer), each serial (compute, then gather) and OVERLAPPED
        (ShardedPipeline(overlap=True): the shard computed in row pieces, piece k exchanged while piece k + 1 is computed).
        Top-level fields = the default method, serial (what earlier rounds reported)."""
        forced = world == 1                  # a group of one would return before the collective: run it anyway (RCCL smoke)

        def serial(method):
            return rotating(lambda t: tac.distributed.all_gather_batch(mdl(t), total_rows=rows_total, method=method,
                                                                       force_collective=forced), inputs)

        def overlapped(method):
            # (what ShardedPipeline(overlap=True) does, on a batch that is already this rank's shard)
            def one(t):
                g = tac.distributed.ChunkedAllGather(rows_total, GATHER_CHUNKS, method=method, force_collective=forced)
                for k in range(g.chunks):
                    lb, le = g.local_rows(k)
                    g.add(k, mdl(t[lb:le]))
                return g.finish()
            return rotating(one, inputs)

        out = {}
        try:        # secondary figures: a failure here must not cost the headline line above
            for name, make in (('rccl', lambda: serial('rccl')), ('p2p', lambda: serial('p2p')),
                               ('rccl_overlap', lambda: overlapped('rccl')), ('p2p_overlap', lambda: overlapped('p2p'))):
                try:
                    fn = make()
                    for _ in range(3):
                        fn()
                    wg, _, _ = repeated(fn, steps, max(1, a.repeats // 5))
                    eg = quantile(wg, 0.5)
                    out[name] = {'value': frames_step * steps / eg, 'unit': 'frames/s', 'ms_per_step': eg / steps * 1e3}
                except Exception as exc:            # noqa: BLE001
                    out[name] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            default = tac.distributed.default_method()
            top = dict(out.get(default, {}))
            top.update({'method': default, 'forced_at_world_1': forced, 'chunks_overlapped': GATHER_CHUNKS, 'forms': out,
                        'note': 'forms: serial = compute then one exchange; *_overlap = the shard computed in %d row pieces, each '
                                'piece exchanged (async, the communicator\'s stream) while the next is computed — expected '
                                '~max(compute, exchange) instead of their sum (DESIGN.md §6)' % GATHER_CHUNKS})
            return top
        except Exception as exc:            # noqa: BLE001 — reported in the line, not swallowed
            return {'error': '%s: %s' % (type(exc).__name__, exc)}

    if distributed:
        result['with_allgather'] = gather_leg(model, xs, world * BATCH, frames_per_step, a.steps)
        if a.config == 'cfg2' and world > 1:
            # BASELINE configs[2]: every rank's 256 x 1 323 000 shard (sr 44 100), compute only and with the one
            # all-gather SURVEY §8e says dominates there (338.7 MB per rank)
            del xs[1:]
            try:
                x3 = [torch.rand(BATCH, CHANNELS, CFG3_LENGTH, device=dev, generator=gen) * 2 - 1]
                m3 = pipeline(CFG3_SR)
                f3 = world * BATCH * CHANNELS * (1 + CFG3_LENGTH // HOP)
                k3 = max(5, a.steps // 5)
                s3 = rotating(m3, x3)
                spin(s3, 0.2)
                w3, m3s, _ = repeated(s3, k3, max(1, a.repeats // 5))
                e3, ms3 = quantile(w3, 0.5), quantile(m3s, 0.5)
                result['cfg3'] = {'workload': 'batch=%d/GPU x 1ch x %dHz x 30s (BASELINE configs[2] sharded over %d GPUs)'
                                              % (BATCH, CFG3_SR, world), 'steps': k3,
                                  'value': f3 * k3 / e3, 'unit': 'frames/s', 'ms_per_step': e3 / k3 * 1e3,
                                  'kernel_ms_mean': ms3,
                                  'with_allgather': gather_leg(m3, x3, world * BATCH, f3, k3)}
            except Exception as exc:        # noqa: BLE001
                result['cfg3'] = {'error': '%s: %s' % (type(exc).__name__, exc)}

    if distributed:
        # every rank's shader clock under the headline kernel (eight GPUs in one chassis need not clock alike: the first thing to
        # compare when the N = 8 efficiency is below the N = 1 line's)
        clocks = [None] * world
        dist.all_gather_object(clocks, clock_mhz)
        result['roofline']['shader_clock_mhz_per_rank'] = clocks
        if rccl_log:
            result['rccl_debug'] = rccl_debug_lines(rccl_log)
    if rank == 0 and world == 1 and x_host is not None:
        result['cpu_baseline'] = cpu_baseline(x_host)
    if rank == 0 and world == 1 and not a.no_live_counters and os.environ.get('TAC_BENCH_LIVE_COUNTERS', '1') != '0':
        roof = result['roofline']
        live, note = live_counters(route)
        if live is not None:
            bound_from_counters(roof, route, clock_mhz, live, note)
        else:
            roof['counters_note'] = 'committed counter pass quoted; a live pass was not possible: %s' % note
    if rank == 0:
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


def dry_run_cpu(a, rank, world):
    """The launcher / rendezvous / max-over-ranks / one-JSON-line control flow of run(), on CPU tensors over gloo."""
    import torch.distributed as dist
    import torchaudio_contrib_amd as tac
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    x = torch.rand(BATCH, CHANNELS, LENGTH, generator=torch.Generator().manual_seed(rank)) * 2 - 1
    model = torch.nn.Sequential(
        *tac.Melspectrogram(num_mels=N_MELS, sample_rate=SR, fft_length=N_FFT, hop_length=HOP), tac.AmplitudeToDb())
    for _ in range(a.warmup):
        model(x)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = model(x)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gathered = tac.distributed.all_gather_batch(y, total_rows=world * BATCH)
        assert tuple(gathered.shape) == (world * BATCH, CHANNELS, N_MELS, FRAMES)
        for method in ('rccl', 'p2p'):          # the overlapped forms of the gather leg: same tensor
            g = tac.distributed.ChunkedAllGather(world * BATCH, 2, method=method)
            for k in range(g.chunks):
                lb, le = g.local_rows(k)
                g.add(k, model(x[lb:le]))
            assert torch.equal(g.finish(), gathered), method
    if rank == 0:
        print(json.dumps({'metric': 'mel frames/sec', 'value': world * BATCH * CHANNELS * FRAMES * a.steps / float(t.item()),
                          'unit': 'frames/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
                          'valid': False, 'data': 'dry run on CPU tensors (control flow only)'}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
