#!/usr/bin/env python
"""profiles/<tag>/{bench_N1.json,kernel_stats.csv,pmc_*.json} -> profiles/<tag>/README.md (numbers only come from
those artefacts; the prose around them is fixed).   python tools/profiles_readme.py r01"""
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
d = os.path.join('profiles', tag)
bench = json.load(open(os.path.join(d, 'bench_N1.json')))
stats = {}
for r in csv.DictReader(open(os.path.join(d, 'kernel_stats.csv'))):
    stats[r['Name']] = r
stage_stats = {}
if os.path.exists(os.path.join(d, 'kernel_stats_stages.csv')):      # (round 5: the headline trace runs without the stages)
    for r in csv.DictReader(open(os.path.join(d, 'kernel_stats_stages.csv'))):
        stage_stats[r['Name']] = r
FRAMES = 256 * 313


def kern(sub):
    if isinstance(sub, tuple):
        for one in sub:
            got = kern(one)
            if got[0]:
                return got
        return None, 0, float('nan')
    table = stats if ('melspec_stream3' in sub or not stage_stats) else stage_stats
    for name, r in table.items():
        if sub in name:
            return name.split('(')[0].replace('void tac::', ''), int(r['Calls']), float(r['AverageNs']) / 1e6
    return None, 0, float('nan')


def pmc(k):
    p = os.path.join(d, 'pmc_%s.json' % k)
    if not os.path.exists(p):
        return {}
    j = json.load(open(p))
    return next(iter(j.values()))


rows = []
for label, sub, key, per_frame in (('fused STFT + power + band-sparse mel + dB', 'melspec_stream3_kernel<1024', 'mel', 2560),
                                   ('complex STFT', ('stft_ring3_kernel<1024, 16, 0', 'stft_stream3_kernel<1024, 16, 0'), 'stft', 10248),
                                   ('power spectrogram', ('stft_ring3_kernel<1024, 16, 1', 'stft_stream3_kernel<1024, 16, 1'), 'spec', 6148)):
    name, calls, ms = kern(sub)
    c = pmc(key)
    alg = FRAMES * per_frame
    rows.append((label, name, calls, ms, alg, c.get('hbm_traffic_bytes_per_launch', float('nan')), c))

out = []
out.append('# profiles/%s — MI355X (gfx950), ROCm 7.2, collected by `tools/collect_profiles.sh %s`, summarised by '
           '`tools/summarize_profiles.py` and this table by `tools/profiles_readme.py`\n' % (tag, tag))
out.append('* `bench_N1.json` — the JSON line of `python bench.py` (N=1, cfg-2) on the same box: **%.0f M mel frames/s**, '
           '%.4f ms per step; CPU baseline on that box %.0f K frames/s (%d threads).'
           % (bench['value'] / 1e6, bench['ms_per_step'], bench['cpu_baseline']['value'] / 1e3, bench['cpu_baseline']['cores']))
out.append('* `kernel_stats.csv` — `rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 '
           '--no-cpu-baseline --no-stages` (the headline steps alone; long torch kernel names truncated); `kernel_stats_stages.csv` '
           '— the same command with its stages (the stage kernels\' rows below come from it).')
out.append('* `pmc_mel.json`, `pmc_stft.json`, `pmc_spec.json` — per-launch means from separate `--pmc` passes '
           '(FETCH_SIZE; WRITE_SIZE; SQ_* set 1; SQ_*/LDS set 2) around `tools/prof_driver.py {mel,stft,spec}`.  HBM bytes '
           'per launch use the corrections of `MI355X_MICROARCH.md` (FETCH_SIZE in KB, x2 on gfx950 for wide coalesced '
           'reads; WRITE_SIZE in KB).\n')
out.append('| kernel (cfg-2: 256x1x160000, 2048/512) | rocprofv3 avg (launches) | algorithmic bytes | HBM traffic (PMC) | alg GB/s | frac of 8 TB/s |')
out.append('|---|---|---|---|---|---|')
for label, name, calls, ms, alg, traffic, c in rows:
    gbs = alg / (ms * 1e-3) / 1e9
    out.append('| %s (`%s`) | %.4f ms (%d) | %.1f MB | %.1f MB | %.0f | %.1f %% |'
               % (label, name, ms, calls, alg / 1e6, traffic / 1e6, gbs, 100 * gbs / 8000))
out.append('')
# ---- the fft_length-4096 kernels (round 6): cfg-4 slice 64 x 480 000 samples, hop 1024 -> 30 016 frames per launch
f4096 = 64 * (1 + 480000 // 1024)
rows4 = []
for label, key, per_frame in (('\\|X\\| rows (`tools/prof_driver.py spec4096`)', 'spec4096', 4 * 1024 + 4 * 2049),
                              ('Melspectrogram 128 bands + dB, one launch (`mel4096`)', 'mel4096', 4 * 1024 + 4 * 128)):
    pth = os.path.join(d, 'pmc_%s.json' % key)
    if not os.path.exists(pth):
        continue
    for kname, c in json.load(open(pth)).items():
        if 'n4096' in kname:
            rows4.append((label, kname.replace('void tac::', ''), per_frame, c))
if rows4:
    out.append('fft_length 4096 (`pmc_spec4096.json`, `pmc_mel4096.json`: the cfg-4 slice, 64 x 480 000 samples = %d frames per launch; per frame):\n' % f4096)
    out.append('| kernel | VALU / frame | LDS instr / frame | LDS index cycles / frame | bank-conflict cycles / frame | VALU busy (SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES) | SQ_WAIT_ANY / SQ_WAVE_CYCLES | algorithmic bytes | HBM traffic (PMC) |')
    out.append('|---|---|---|---|---|---|---|---|---|')
    for label, kname, per_frame, c in rows4:
        g = lambda k: c.get(k, float('nan'))
        out.append('| %s `%s` | %.0f | %.0f | %.0f | %.0f | %.0f %% | %.0f %% | %.1f MB | %.1f MB |'
                   % (label, kname, g('SQ_INSTS_VALU') / f4096, g('SQ_INSTS_LDS') / f4096, g('SQ_LDS_IDX_ACTIVE') / f4096,
                      g('SQ_LDS_BANK_CONFLICT') / f4096, 100 * g('SQ_ACTIVE_INST_VALU') / g('SQ_BUSY_CU_CYCLES'),
                      100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'), f4096 * per_frame / 1e6, g('hbm_traffic_bytes_per_launch') / 1e6))
    for name, r4 in stage_stats.items():
        if 'stft_n4096_s3_kernel<2, 12, false>' in name:
            ms4 = float(r4['AverageNs']) / 1e6
            alg4 = 512 * (1 + 2880000 // 1024) * (4 * 1024 + 4 * 2049)
            out.append('\nBASELINE configs[3] at full size in the stage trace (`kernel_stats_stages.csv`): `stft_n4096_s3_kernel<2, 12, false>` %.3f ms average over %s launches = %.0f GB/s = %.1f %% of 8 TB/s (17.70 GB algorithmic per launch).'
                       % (ms4, r4['Calls'], alg4 / ms4 / 1e6, alg4 / ms4 / 1e6 / 80))
    out.append('`steady_state/mel4096_banks.txt` — `tools/r06/mel4096_pack_info.py`: what `tac_melbank_pack(..., n_fft = 4096)` builds for the usual mel banks and the time of the one-launch chain with each.\n')
r = bench['roofline']
sb = bench.get('single_buffer')
if sb:
    out.append('`bench.py` visits %s; on ONE re-read batch (it fits the 256 MiB Infinity Cache — what rounds 1 and 2 timed) the same '
               'loop runs %.4f ms per step (%.0f M frames/s): the fused kernel barely notices, the stage kernels do (below).'
               % (bench['config'].get('input_buffers', 'several input batches'), sb['ms_per_step'], sb['value'] / 1e6))
    st = bench.get('stages', {})
    if 'stft_complex' in st and 'single_buffer_kernel_ms_mean' in st['stft_complex']:
        out.append('Stage kernels, rotating batches vs one re-read batch: complex STFT %.4f vs %.4f ms, power spectrogram %.4f vs '
                   '%.4f ms (complex STFT from HBM: %.0f %% of 8 TB/s).'
                   % (st['stft_complex']['kernel_ms_mean'], st['stft_complex']['single_buffer_kernel_ms_mean'],
                      st['spectrogram_power']['kernel_ms_mean'], st['spectrogram_power']['single_buffer_kernel_ms_mean'],
                      100 * st['stft_complex']['frac_of_hbm_peak']))
out.append('`bench.py` on the same box, HIP events on the launch stream, un-profiled: fused kernel %.4f ms mean / %.4f ms '
           'median -> %.0f GB/s = %.1f %% of 8 TB/s; complex STFT %.4f ms (%.1f %%), power spectrogram %.4f ms (%.1f %%).'
           % (r['kernel_ms_mean'], r['kernel_ms_median'], r['achieved'], 100 * r['frac'],
              bench['stages']['stft_complex']['kernel_ms_median'], 100 * bench['stages']['stft_complex']['frac_of_hbm_peak'],
              bench['stages']['spectrogram_power']['kernel_ms_median'], 100 * bench['stages']['spectrogram_power']['frac_of_hbm_peak']))
fm = bench.get('stages', {}).get('filterbank_mfma_dense')
if fm:
    out.append('Filterbank stage as a dense fp32 MFMA GEMM (`gemm_fb_kernel`, random 1025 x 128 bank, cfg-2 power spectrogram): '
               '%.4f ms = %.1f TFLOP/s = %.0f %% of the 157.3 TFLOP/s f32 MFMA peak.' % (fm['kernel_ms_mean'], fm['achieved_TFLOPs'], 100 * fm['frac_of_f32_mfma_peak']))
pf = os.path.join(d, 'pmc_fb.json')
if os.path.exists(pf):
    for name, c in json.load(open(pf)).items():
        if 'gemm_fb_kernel' in name and c.get('SQ_INSTS_VALU_MFMA_F32'):
            out.append('Matrix-core counters of that GEMM (`pmc_fb.json`, per launch): SQ_INSTS_VALU_MFMA_F32 = %.2f M '
                       '(v_mfma_f32_32x32x2_f32, %.1f %% of its %.2f M VALU instructions), SQ_VALU_MFMA_BUSY_CYCLES / '
                       '(4 SIMDs x SQ_BUSY_CU_CYCLES) = %.0f %% matrix-pipe utilisation.  The kernels of the measured hot path '
                       'issue no MFMA (SQ_INSTS_VALU_MFMA_F32 = 0 in `pmc_mel.json`): the band-sparse contraction is 2 050 '
                       'MACs per frame on the packed-f32 VALU; DESIGN.md §3.3 has the A/B against the MFMA form.'
                       % (c['SQ_INSTS_VALU_MFMA_F32'] / 1e6, 100 * c['SQ_INSTS_VALU_MFMA_F32'] / c['SQ_INSTS_VALU'],
                          c['SQ_INSTS_VALU'] / 1e6, 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES'])))
out.append('')
out.append('Counters per launch (per frame = / 80 128):')
out.append('')
out.append('| kernel | VALU / frame | LDS instr / frame | VMEM rd / frame | VMEM wr / frame | LDS busy (IDX_ACTIVE / CU / kernel cycles) | LDS bank-conflict cycles / LDS cycles | SQ_WAIT_ANY / SQ_WAVE_CYCLES |')
out.append('|---|---|---|---|---|---|---|---|')
for label, name, calls, ms, alg, traffic, c in rows:
    if not c:
        continue
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0
    out.append('| %s | %.0f | %.0f | %.1f | %.1f | %.0f %% | %.0f %% | %.0f %% |'
               % (label, c['SQ_INSTS_VALU'] / FRAMES, c['SQ_INSTS_LDS'] / FRAMES, c['SQ_INSTS_VMEM_RD'] / FRAMES,
                  c['SQ_INSTS_VMEM_WR'] / FRAMES, 100 * c['SQ_LDS_IDX_ACTIVE'] / 256 / cyc,
                  100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']))
out.append('')
if all('SQ_WAIT_INST_ANY' in c for _, _, _, _, _, _, c in rows if c):
    out.append('Where a wave\'s cycles go (disjoint SQ buckets, fractions of SQ_WAVE_CYCLES): parked at s_waitcnt / s_barrier '
               '(SQ_WAIT_ANY), ready but not issued (SQ_WAIT_INST_ANY; of which waiting for the LDS queue SQ_WAIT_INST_LDS), '
               'issuing (SQ_ACTIVE_INST_ANY; VALU / LDS / scalar shares):')
    out.append('')
    out.append('| kernel | parked | issue-stalled (LDS queue) | issuing (VALU / LDS / scalar) |')
    out.append('|---|---|---|---|')
    for label, name, calls, ms, alg, traffic, c in rows:
        if not c:
            continue
        wc = c['SQ_WAVE_CYCLES']
        out.append('| %s | %.0f %% | %.0f %% (%.0f %%) | %.0f %% (%.0f / %.0f / %.0f %%) |'
                   % (label, 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_WAIT_INST_ANY'] / wc, 100 * c['SQ_WAIT_INST_LDS'] / wc,
                      100 * c['SQ_ACTIVE_INST_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_VALU'] / wc,
                      100 * c['SQ_ACTIVE_INST_LDS'] / wc, 100 * c['SQ_ACTIVE_INST_SCA'] / wc))
    out.append('')
out.append('The fused kernel and the complex STFT run 3 waves/SIMD, the power spectrogram 4 (one 12- / 16-wave workgroup per CU whose '
           'waves draw frames from a workgroup counter), no scratch.  HBM traffic equals the algorithmic bytes to within 0.5 %: '
           'nothing is re-read; the fused kernel is bound by its VALU + LDS instruction streams — with three waves per SIMD the '
           'SIMD issues VALU work 3 x the per-wave VALU share above of its cycles and the LDS is busy as tabulated — not by memory.  '
           '`ab/` holds the raw same-process A/B outputs of the round (tables: tools/ablation/README.md), `ubench/` the '
           'micro-benchmark outputs; DESIGN.md §3.2 / §3.3 the reasoning.')
for fname, what in (('kernel_stats_backward.csv',
                     '`tools/prof_driver.py grad 120`: forward + backward of the reference idiom `Sequential(*Melspectrogram(...), '
                     'AmplitudeToDb())` at cfg-2 with `requires_grad` on the waveform: the chain is deferred as usual (one fused '
                     'forward kernel for the linear mel values + the dB op, which keeps them for its gradient) and differentiates through '
                     'the `tac_amd::melspectrogram` op — ONE backward kernel (`melspec_backward_ring3_kernel`: filterbank adjoint per '
                     'bin pair, frames re-transformed, adjoint of the norm, inverse transform, overlap-add in a register ring; twelve '
                     'waves per CU) and the unpadding / border fold'),
                    ('kernel_stats_backward_fused_op.csv',
                     '`tools/prof_driver.py gradf 120`: the same through the factory container (`Melspectrogram(...)` called as '
                     'ONE `tac_amd::melspectrogram` op) followed by `AmplitudeToDb`: fused forward kernel; backward = the linear mel '
                     'values recomputed by the forward kernel for the dB adjoint, the backward kernel, unpadding / border fold'),
                    ('kernel_stats_backward_spectrogram.csv',
                     '`tools/prof_driver.py gradspec 60`: training step through `Spectrogram(2048, 512, power=2)`: forward kernel, the same '
                     'backward kernel fed with the gradient of the power spectrogram (no filterbank stage), border fold'),
                    ('kernel_stats_backward_n1024.csv',
                     '`tools/prof_driver.py grad1024 60`: the 80-band chain at fft_length 1024 / hop 256 trained through the reference '
                     'idiom: fused forward kernel (two frames per wave), `melspec_backward_ring3_multi_kernel` (two lane groups per wave, '
                     'each with its own register ring; filterbank adjoint inside), border fold'),
                    ('kernel_stats_backward_n512.csv',
                     '`tools/prof_driver.py grad512 60`: the same at fft_length 512 / hop 128 (four frames / lane groups per wave)'),
                    ('kernel_stats_backward_n400.csv',
                     '`tools/prof_driver.py grad400h160 60`: the 80-band speech front end (fft_length 400, hop 160) trained through '
                     'the reference idiom: fused forward kernel, complex stft recomputed by the mixed-radix kernel, the inverse '
                     'form of that kernel with the norm\'s adjoint folded into its load (`stft_n400_backward_kernel`), gather '
                     'overlap-add')):
    pb = os.path.join(d, fname)
    if not os.path.exists(pb):
        continue
    out.append('')
    out.append('`%s` — `rocprofv3 --kernel-trace --stats` of %s:' % (fname, what))
    out.append('')
    out.append('| kernel | calls | average |')
    out.append('|---|---|---|')
    tot = 0.0
    recs = [r for r in csv.DictReader(open(pb)) if 'tac::' in r['Name']]
    steps = max(int(r['Calls']) for r in recs)
    for r in recs:
        if int(r['Calls']) >= steps:
            tot += float(r['TotalDurationNs']) / steps / 1e6
            out.append('| `%s` | %s | %.4f ms |' % (r['Name'].split('(')[0].replace('void ', ''), r['Calls'], float(r['AverageNs']) / 1e6))
    out.append('| kernel time per step (total / %d steps) | | %.3f ms |' % (steps, tot))
pbk = os.path.join(d, 'pmc_backward.json')
if os.path.exists(pbk):
    for name, c in json.load(open(pbk)).items():
        if ('melspec_backward_ring3_kernel' in name or 'spectrogram_backward_ola_kernel' in name) and 'SQ_WAVE_CYCLES' in c:
            wc = c['SQ_WAVE_CYCLES']
            out.append('')
            out.append('Counters of `%s` (`pmc_backward.json`, per launch, separate `--pmc` passes around '
                       '`tools/prof_driver.py grad 3`): %.0f VALU and %.0f LDS instructions per frame; a wave is parked %.0f %%, '
                       'issue-stalled %.0f %% (LDS queue %.0f %%) and issuing %.0f %% of its cycles; LDS bank-conflict cycles %.0f %% of '
                       'LDS cycles; HBM traffic %.0f MB per launch (%.0f MB read + %.0f MB written; the samples are 164 MB, the '
                       'gradient of the mel values 41 MB; written: the waveform gradient\'s clean interior + the padded border runs and '
                       'segment-border sums the fold kernel finishes).'
                       % (name.split('<')[0].replace('void tac::', ''), c['SQ_INSTS_VALU'] / FRAMES, c['SQ_INSTS_LDS'] / FRAMES, 100 * c['SQ_WAIT_ANY'] / wc,
                          100 * c['SQ_WAIT_INST_ANY'] / wc, 100 * c['SQ_WAIT_INST_LDS'] / wc, 100 * c['SQ_ACTIVE_INST_ANY'] / wc,
                          100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], c['hbm_traffic_bytes_per_launch'] / 1e6,
                          c['hbm_read_bytes_corrected'] / 1e6, c['hbm_write_bytes'] / 1e6))
out.append('')
out.append('`steady_state/time_steady.txt`, `steady_state/time_others.txt` — `tools/time_steady.py` / `tools/time_others.py` on the same '
           'box (0.5 s spin-up per kernel, then 100 / 60 back-to-back launches with per-launch HIP events; medians): every other '
           'kernel of the library, incl. the fft_length-4096 Melspectrogram chain and `hpss`.')
out.append('')
out.append('Micro-benchmarks behind the design decisions (sources in `tools/ubench/`, outputs under `ubench/` of the round that ran '
           'them): `valu_rate` (scalar vs packed f32 issue rates), `lds_rate` (LDS access shapes at 8 waves/CU), `hbm_rate` (write / '
           'read / 1:4 mix ceilings: 4.4-5.6 / 6.5 / 5.1-5.6 TB/s), `row_store_rate` (the STFT rows\' access pattern without '
           'arithmetic: 8200-byte rows 4.3 TB/s, 128-byte aligned 8192-byte rows 4.8-5.0 TB/s).')
out.append('')
out.append('`../r02/`, `../r01/` hold the same measurements for rounds 2 and 1 (two-waves-per-SIMD streaming kernel 0.127 ms on one re-read batch; three-phase fused kernel, 0.174 ms) and `../r01_baseline_v0/` for the first correct version (0.61 ms).')
if os.path.exists(os.path.join(d, 'kernel_stats_big_fft.csv')):
    out.append('')
    out.append('`kernel_stats_big_fft.csv`, `kernel_stats_nonpow2.csv` — `rocprofv3 --kernel-trace --stats` of `tools/r05/time_big_fft.py` / '
               '`tools/r05/time_nonpow2.py`: the kernel families added in round 5 beside torch.stft\'s rocFFT kernels in the same process — '
               '`stft_big_kernel<S, mode, waves>` (fft_length 8192 / 16384 / 32768 = S 4 / 8 / 16: 0.317 / 0.349 / 0.562 ms complex rows, '
               '0.286 / 0.322 / 0.547 ms |X|^2 rows for 16 x 2.88 M samples), `stft_smooth_kernel<mode, threads per frame, full table>` '
               '(480 ... 2048: 0.542 ms mean over the sizes timed; above: 0.670 ms; 6000: 0.844 ms), the float64 kernels and the HPSS widths '
               '33 ... 63.  `ab/batch48 ... 66`: the store-pattern micro-benchmark in global order and the kernel A/Bs that followed.')
open(os.path.join(d, 'README.md'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
