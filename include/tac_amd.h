/* tac_amd.h — C ABI of libtac_amd.so: the MI355X (gfx950) engine behind the
 * torchaudio-contrib Melspectrogram hot path.
 *
 * The reference (keunwoochoi/torchaudio-contrib) is pure Python over stock torch ops and has
 * no FFI of its own; the boundary it exposes is its functional API
 * (torchaudio_contrib/functional.py).  Each entry point below replaces the torch-op body of
 * one (or a fused chain) of those functions; the citation names the reference lines.
 *
 * Conventions
 *   - plain C types only: raw DEVICE pointers, sizes, scalars.  No torch types.
 *   - every call is asynchronous on the caller-supplied HIP stream (`stream` is a
 *     hipStream_t passed as void*; NULL = the default stream).
 *   - outputs are caller-allocated; inputs are never written.
 *   - return value: TAC_OK (0) or a negative TAC_E_* code; tac_strerror() gives text.
 *     The only global state is an immutable, mutex-guarded twiddle-table cache keyed by
 *     (n_fft, device); the first call for a new n_fft allocates + uploads it (do that
 *     warm-up before capturing a hipGraph).
 *   - "rows" = product of all leading dims (batch x channel), functional.py:89-91.
 *   - frame-major physical layouts: complex STFT out[rows][T][F][2], spectrogram
 *     out[rows][T][F], mel out[rows][T][M].  The Python layer returns them as the logical
 *     (rows, F|M, T[,2]) strided views the reference itself produces (torch.stft /
 *     transpose-matmul-transpose give exactly these strides).
 */
#ifndef TAC_AMD_H
#define TAC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAC_OK 0
#define TAC_E_INVALID (-1)      /* bad argument (null pointer, non-positive size, ...)        */
#define TAC_E_UNSUPPORTED (-2)  /* n_fft neither a power of two in [32, 32768] nor even with a 7-smooth half, n_mels too large…  */
#define TAC_E_SHORT_INPUT (-3)  /* signal too short for the requested padding / n_fft          */
#define TAC_E_LAUNCH (-4)       /* HIP runtime error; see tac_last_hip_error()                 */

/* pad_mode values (torch.nn.functional.pad modes accepted by torch.stft, functional.py:57-58) */
#define TAC_PAD_CONSTANT 0
#define TAC_PAD_REFLECT 1
#define TAC_PAD_REPLICATE 2
#define TAC_PAD_CIRCULAR 3

const char* tac_strerror(int code);
int tac_last_hip_error(void);
int tac_abi_version(void);

/* Number of STFT frames and bins for the given geometry (0 on invalid geometry).
 * T = 1 + (L + 2*(center ? n_fft/2 : 0) - n_fft) / hop   (tests/test_functional.py:14-15). */
int64_t tac_num_frames(int64_t L, int n_fft, int hop, int center);
int tac_num_bins(int n_fft, int onesided);

/* Geometry shared by the three STFT-family entry points. */
typedef struct tac_stft_desc {
    int64_t rows;        /* batch*channel                                              */
    int64_t length;      /* samples per row (L)                                        */
    int64_t row_stride;  /* elements between consecutive rows of `wave`                */
    int32_t n_fft;       /* power of two, 32..4096; or 400 (STFT / spectrogram, one-sided); any even length <= 8192 whose half is
                            7-smooth (480, 882, 960, 1200, 1920 ...) and 8192: the STFT / spectrogram rows AND their gradient
                            (tac_stft_f32, tac_spectrogram_f32, tac_stft_backward_f32: stft_smooth.hip); 16384 / 32768: the forward
                            rows only (stft_big.hip) */
    int32_t hop;         /* > 0                                                        */
    int32_t win_length;  /* 1..n_fft; window is zero-padded centred to n_fft           */
    int32_t center;      /* 1: pad n_fft/2 both sides with pad_mode                    */
    int32_t pad_mode;    /* TAC_PAD_*                                                  */
    int32_t normalized;  /* 1: multiply by n_fft^-0.5                                  */
    int32_t onesided;    /* 1: F = n_fft/2+1, 0: F = n_fft                             */
    int32_t reserved;
} tac_stft_desc;

/* (1) functional.stft, functional.py:48-113 (the torch.stft call at :99-107).
 *     out: float[rows][T][F][2]. */
int tac_stft_f32(const float* wave, const float* window, const tac_stft_desc* d,
                 float* out, void* stream);

/* (2) Spectrogram = stft + complex_norm(power) (functional.py:116-128, layers.py:267-304),
 *     magnitude/power taken in the FFT epilogue; optionally followed by amplitude_to_db
 *     (functional.py:277-296) when db != 0.   out: float[rows][T][F]. */
int tac_spectrogram_f32(const float* wave, const float* window, const tac_stft_desc* d,
                        float power, int db, float db_ref, float db_amin,
                        float* out, void* stream);

/* (3) Melspectrogram chain fused in one kernel: stft -> complex_norm(power) ->
 *     apply_filterbank (functional.py:172-184) [-> amplitude_to_db], layers.py:307-381.
 *     fb: DEVICE float[F][n_mels] row-major dense filterbank exactly as create_mel_filter returns it
 *     (functional.py:131-169); fb_plan_host: HOST int32[2*ceil(n_mels/16)] from tac_filterbank_plan
 *     (passed to the kernel by value).  Requires onesided geometry with n_fft <= 2048 and a filterbank
 *     sparse enough for the register-resident weights (sum over 16-band tiles of ceil(range/4) <= 384
 *     at n_fft = 2048) and power in {1, 2}; returns TAC_E_UNSUPPORTED otherwise — callers then chain (2) and (4).
 *     out: float[rows][T][n_mels]. */
int tac_melspec_f32(const float* wave, const float* window, const tac_stft_desc* d, float power,
                    const float* fb, const int32_t* fb_plan_host, int32_t n_mels,
                    int db, float db_ref, float db_amin, float* out, void* stream);

/* (3b) The same fused chain with a band-sparse VALU contraction instead of the MFMA tile: every (frame, band)
 *      output is one thread's dot product over the band's contiguous bin run, weights packed in LDS.  For
 *      triangular mel banks (1.5 % non-zero) this is the faster form; tac_melbank_pack returns
 *      TAC_E_UNSUPPORTED for banks that are not band-sparse enough (sum over bands of the padded support
 *      lengths > 3072), in which case callers use (3).
 *      tac_melbank_pack: one-off per (filterbank, n_fft); copies fb to the host (synchronises `stream`), deals the
 *      bands to lane groups longest-first and uploads wpack (DEVICE float[wpack_cap >= 3072]) and desc (DEVICE
 *      int32[desc_cap >= 4096]); info_host: HOST int32[8] = {weight floats, desc stride, lane groups, max group load, 0, 0, 0, 0};
 *      for n_fft = 2048 the pack is the lane layout of the streaming kernel (lane l owns bands l, 64 + l, ...; at most
 *      256 bands): wpack = float[steps][64][2] zero-padded pair weights, desc = int32[slots][64] first bins,
 *      info_host = {weight floats, slots, 64 (+ 256 since round 6 when cell 64 s + l holds band n_mels - 1 - (64 s + l): banks whose band
 *      count is not a multiple of 64, so that their widest bands share slot 0), total steps, steps of slot 0..3}; wpack_cap >= 8192.
 *      For n_fft = 4096 (round 6: the chain in ONE launch, csrc/stft_n4096_s3.hpp; at most 256 bands, power in {1, 2}, frames 16-byte
 *      aligned, rows of at least one frame — TAC_E_UNSUPPORTED otherwise and callers chain (2) and (4b)) cells (slot, lane) of up to six
 *      slots, every slot storing the step pairs of ITS longest run: uncut, cell c is band c (or n_mels - 1 - c); where that table does
 *      not fit the LDS, bands are cut into pieces that a mix table gathers.  wpack = float[steps][64][4] (wpack_cap >= 256 * steps),
 *      desc = int32[6][64] first bins, int32[6] step pairs per slot, 10 ints of padding, int32[rounds][pieces][64] mix (desc_cap >=
 *      400 + 64 * rounds * pieces), info_host = {weight floats, slots, 1000 + 4096, total steps, waves per workgroup the table leaves
 *      room for (12 / 11 / 8), pieces per band in the mix table (0: uncut), rounds of 64 bands, uncut cells in reversed band order};
 *      TAC_E_UNSUPPORTED when no layout fits the LDS beside eight waves (dense banks).
 *      tac_melbank_plan_pieces_host (a host tool, no device access): the PIECE layout of round 4 for a host copy of the bank —
 *      a lane runs three segments of L0 / L1 / L2 four-tap steps, each holding a piece of a band; a band takes up to three
 *      pieces in adjacent lanes; 12 steps instead of 18 for the standard 128-band bank.  seg_steps int32[3]; first / band /
 *      index int32[192]; weights float[weights_cap >= 256 * (L0 + L1 + L2)]; TAC_E_UNSUPPORTED when no segment triple fits.
 *      The kernel form that contracted this layout measured 3 - 5 % slower than the lane layout and is not shipped
 *      (tools/ablation/README.md); tests emulate its contraction on the plan. */
int tac_melbank_pack(const float* fb, int32_t n_freqs, int32_t n_mels, int32_t n_fft, float* wpack,
                     int32_t wpack_cap, int32_t* desc, int32_t desc_cap, int32_t* info_host, void* stream);
/* ... the same tables without a device (round 6; n_fft = 256, 400, 512, 1024, 2048 or 4096): fb_host, wpack_host, desc_host are HOST buffers, nothing is launched
 *      or copied — what tests/test_host_api.py emulates the kernels' contraction on. */
int tac_melbank_pack_host(const float* fb_host, int32_t n_freqs, int32_t n_mels, int32_t n_fft, float* wpack_host,
                          int32_t wpack_cap, int32_t* desc_host, int32_t desc_cap, int32_t* info_host);
int tac_melbank_plan_pieces_host(const float* fb_host, int32_t n_freqs, int32_t n_mels, int32_t* seg_steps, int32_t* first,
                                 int32_t* band, int32_t* index, float* weights, int32_t weights_cap);
int tac_melspec_sparse_f32(const float* wave, const float* window, const tac_stft_desc* d, float power,
                           const float* wpack, const int32_t* desc, const int32_t* info_host, int32_t n_mels,
                           int db, float db_ref, float db_amin, float* out, void* stream);

/* (3c) The fused chain reading the waveform in its stored sample format (the step before the path —
 *      PCM / mu-law decode, functional.py:338-354 — folded into the frame load, the samples are converted in registers):
 *      TAC_SAMPLES_I16 = int16 PCM, value = sample * 2^-15; TAC_SAMPLES_MULAW_U8 / _I64 = 8-bit mu-law codes stored as
 *      uint8 / int64 (what mu_law_encoding returns), value = decode_lut[code & 255] with decode_lut the DEVICE float[256]
 *      table of (8).  d->row_stride and d->length count samples.  Served by the fft_length 2048 streaming kernel and, for
 *      power 2, by the fft_length 256 / 400 / 512 / 1024 kernels; TAC_E_UNSUPPORTED otherwise (callers then convert with (8) /
 *      tac_pcm16_to_f32 and use (3b)). */
#define TAC_SAMPLES_F32 0
#define TAC_SAMPLES_I16 1
#define TAC_SAMPLES_MULAW_U8 2
#define TAC_SAMPLES_MULAW_I64 3
int tac_melspec_sparse_coded_f32(const void* samples, int32_t sample_format, const float* decode_lut,
                                 const float* window, const tac_stft_desc* d, float power, const float* wpack,
                                 const int32_t* desc, const int32_t* info_host, int32_t n_mels, int db,
                                 float db_ref, float db_amin, float* out, void* stream);
/* int16 PCM -> float32 (x * 2^-15), for the kernels without a coded frame load */
int tac_pcm16_to_f32(const int16_t* x, int64_t n, float* out, void* stream);

/* TAC_OK when (3) can run this geometry + filterbank plan, TAC_E_UNSUPPORTED when it cannot
 * (no launch, no device access). */
int tac_melspec_supported(const tac_stft_desc* d, float power, const int32_t* fb_plan_host,
                          int32_t n_mels);

/* Per 16-band tile [first, last+1) non-zero bin range of a dense filterbank, computed by a device
 * kernel into plan (DEVICE int32[2*ceil(n_mels/16)]).  When plan_host is non-NULL the stream is
 * synchronised and the plan is also copied there (one-off per filterbank). */
int tac_filterbank_plan(const float* fb, int32_t n_freqs, int32_t n_mels, int32_t* plan,
                        int32_t* plan_host, void* stream);

/* (4) functional.apply_filterbank, functional.py:172-184: out[r][t][m] = sum_f spec[r][f][t]*fb[f][m]
 *     as an fp32 MFMA (v_mfma_f32_32x32x2_f32) tile with zero-block skipping driven by fb_plan
 *     (NULL = dense).  spec element (r, f, t) lives at spec[r*stride_r + f*stride_f + t*stride_t].
 *     out: float[rows][T][n_mels]. */
int tac_apply_filterbank_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames,
                             int64_t stride_r, int64_t stride_f, int64_t stride_t,
                             const float* fb, const int32_t* fb_plan, int32_t n_mels,
                             float* out, void* stream);

/* (4b) functional.apply_filterbank for a frame-major spectrogram (the bins of a frame contiguous, frames stride_t
 *      floats apart — the layout the kernels of this library write) and a band-sparse bank packed with
 *      tac_melbank_pack(fb, n_freqs, n_mels, n_fft = 0, ...): streams the spectrogram once and runs the fused
 *      kernel's contraction.  out: frame-major [rows][n_frames][n_mels]. */
int tac_apply_filterbank_sparse_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames,
                                    int64_t stride_r, int64_t stride_t, const float* wpack,
                                    const int32_t* desc, const int32_t* info_host, int32_t n_mels,
                                    float* out, void* stream);
/* ... followed by functional.amplitude_to_db (db != 0: 10 (log10(max(x^2, db_amin)) - log10 db_ref)) in the same pass: the
 *      filterbank + dB tail of Melspectrogram -> AmplitudeToDb for a spectrogram that already exists (the chain from the waveform
 *      is one launch of (3b) up to fft_length 4096). */
int tac_apply_filterbank_sparse_db_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames,
                                       int64_t stride_r, int64_t stride_t, const float* wpack, const int32_t* desc,
                                       const int32_t* info_host, int32_t n_mels, int db, float db_ref, float db_amin,
                                       float* out, void* stream);

/* (5) functional.complex_norm, functional.py:116-128: out[i] = |(x[2i], x[2i+1])|^power. */
int tac_complex_norm_f32(const float* x, int64_t n, float power, float* out, void* stream);

/* (5b) functional.angle / functional.magphase, functional.py:187-201: phase[i] =
 *      atan2(x[2i+1], x[2i]); when mag != NULL also mag[i] = |(x[2i], x[2i+1])|^power, in the same pass. */
int tac_magphase_f32(const float* x, int64_t n, float power, float* mag, float* phase, void* stream);

/* (5c) functional.phase_vocoder, functional.py:204-274.  spec: rows x n_freqs x n_frames complex
 *      pairs with arbitrary element strides (in floats; the pair itself contiguous); phase_advance: n_freqs floats;
 *      idx0/idx1/alpha (device, n_out each): for output frame i the two source frames floor(t_i), floor(t_i + 1)
 *      (indices >= n_frames are the reference's zero padding) and the weight t_i mod 1, with t = arange(0, n_frames,
 *      rate) evaluated the way the reference evaluates it (functional.py:233-237).  out: frame-major
 *      [rows][n_out][n_freqs][2]. */
int tac_phase_vocoder_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames,
                          int64_t stride_r, int64_t stride_f, int64_t stride_t,
                          const float* phase_advance, const int32_t* idx0, const int32_t* idx1,
                          const float* alpha, int64_t n_out, float* out, void* stream);
/*      The float32 recurrence as written is ill-conditioned (which is why the reference's own test runs in float64,
 *      tests/test_functional.py:69-116); the _f32 kernel carries exp(i phase) as a unit phasor instead of the running sum (only
 *      the sum modulo one turn reaches the output) and agrees with the float64 evaluation to ~1e-6 rad; the _f64 entry point
 *      is that float64 call site itself: same arguments with double data (strides in doubles), the formula as written. */
int tac_phase_vocoder_f64(const double* spec, int64_t rows, int32_t n_freqs, int64_t n_frames,
                          int64_t stride_r, int64_t stride_f, int64_t stride_t,
                          const double* phase_advance, const int32_t* idx0, const int32_t* idx1,
                          const double* alpha, int64_t n_out, double* out, void* stream);
/* ... and its gradient with respect to the spectrogram (round 6; float32): grad_out [rows][n_out][n_freqs][2] and grad_spec
 *      [rows][n_frames][n_freqs][2] frame-major and dense, grad_spec ZERO-INITIALISED by the caller (the kernel accumulates: a source frame
 *      is read by several output frames when rate < 1, by none when rate > 2); spec, strides and the grid (idx0, idx1, alpha) as in the
 *      forward call.  The gradient with respect to phase_advance is zero (the wrap and the advance cancel). */
int tac_phase_vocoder_backward_f32(const float* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                                   int64_t stride_f, int64_t stride_t, const int32_t* idx0, const int32_t* idx1, const float* alpha,
                                   int64_t n_out, const float* grad_out, float* grad_spec, void* stream);

/* (1d)-(6d) The path in float64 (the reference keeps f64 -> f64: functional.py:48-113, :116-128, :172-184, :187-201,
 *      :277-314).  Same argument meaning as the _f32 entry points with double data; d->n_fft: any even length <= 8192 whose
 *      half is 5-smooth (N/2-point mixed-radix complex Stockham transform per workgroup in LDS), or any other length
 *      <= 4096 (direct O(N^2) transform per frame: meant for short odd sizes, see _hip64.py for the sizes Python routes there);
 *      tac_apply_filterbank_f64 takes the dense bank (no plan) and optionally applies amplitude_to_db to its result;
 *      tac_magphase_f64: mag and / or phase may be NULL (complex_norm / angle alone). */
int tac_stft_f64(const double* wave, const double* window, const tac_stft_desc* d, double* out, void* stream);
int tac_spectrogram_f64(const double* wave, const double* window, const tac_stft_desc* d, double power, int db,
                        double db_ref, double db_amin, double* out, void* stream);
int tac_apply_filterbank_f64(const double* spec, int64_t rows, int32_t n_freqs, int64_t n_frames, int64_t stride_r,
                             int64_t stride_f, int64_t stride_t, const double* fb, int32_t n_mels, int db,
                             double db_ref, double db_amin, double* out, void* stream);
int tac_magphase_f64(const double* z, int64_t n, double power, double* mag, double* phase, void* stream);
int tac_amplitude_to_db_f64(const double* x, int64_t n, double ref, double amin, double* out, void* stream);
int tac_db_to_amplitude_f64(const double* x, int64_t n, double ref, double* out, void* stream);

/* (6) functional.amplitude_to_db, functional.py:277-296: 10*(log10(max(x^2, amin)) - log10(ref)). */
int tac_amplitude_to_db_f32(const float* x, int64_t n, float ref, float amin, float* out,
                            void* stream);

/* (6b) functional.db_to_amplitude, functional.py:299-314: (10^(x/10 + log10 ref))^0.5. */
int tac_db_to_amplitude_f32(const float* x, int64_t n, float ref, float* out, void* stream);

/* (7) functional.mu_law_encoding, functional.py:317-335.  out: int64 codes.
 *     thresholds (optional, device): int32[n_pos + n_neg] magnitude bit patterns at which the
 *     reference's code changes for x >= 0 (ascending, first n_pos) and x <= 0 (next n_neg);
 *     they are a fast path for |x| <= 1 (a table search instead of the logarithm).  Without them,
 *     and for |x| > 1 / NaN, the closed form is evaluated with the reference CPU path's exact
 *     float32 roundings; either way the codes are bit-identical to the reference's.
 *     zero_code = code of x == 0. */
int tac_mulaw_encode_f32_i64(const float* x, int64_t n, int32_t n_quantize,
                             const int32_t* thresholds, int32_t n_pos, int32_t n_neg,
                             int32_t zero_code, int64_t* out, void* stream);

/* (8) functional.mu_law_decoding, functional.py:338-354, for int64 codes.  lut (optional,
 *     device): float[n_quantize] table used for codes in [0, n_quantize); codes outside it
 *     (or lut == NULL) go through the closed form. */
int tac_mulaw_decode_i64_f32(const int64_t* codes, int64_t n, int32_t n_quantize,
                             const float* lut, float* out, void* stream);

/* (8b) same for float-valued codes (the reference accepts float input, functional.py:349, and its own test
 *      bit-compares that form, tests/test_functional.py:182-193): a code that is an exact integer in
 *      [0, n_quantize) is decoded through lut (optional, device, float[n_quantize]) like (8); everything else
 *      through the closed form. */
int tac_mulaw_decode_f32_f32(const float* codes, int64_t n, int32_t n_quantize, const float* lut,
                             float* out, void* stream);
/*      float64 (round 5): the same formulas evaluated in double, as the reference's CPU path does for double input
 *      (functional.py:329-335, 349-354); codes as int64 (codes_are_i64 != 0) or double. */
int tac_mulaw_encode_f64_i64(const double* x, int64_t n, int32_t n_quantize, int64_t* out, void* stream);
int tac_mulaw_decode_f64(const void* codes, int32_t codes_are_i64, int64_t n, int32_t n_quantize, double* out, void* stream);

/* (9) Gradients (the reference differentiates through stock torch ops).  All asynchronous on `stream`,
 *     caller-allocated outputs, float32, the frame-major layouts of the forward entry points.
 *     tac_stft_backward_f32: adjoint of (1) up to the overlap-add: grad_spec[rows][T][F][2] (one-sided) ->
 *       grad_frames[rows][T][n_fft] = window[n] * scale * Re sum_k grad_spec[k] e^{+2 pi i k n / n_fft}  (one inverse
 *       real FFT per frame on the same wave-level FFT as the forward pass; the even lengths with a 7-smooth half of (1): the
 *       generic Stockham passes of stft_smooth.hip, which also serve fft_length 8192; not 16384 / 32768).
 *     tac_stft_norm_backward_f32: the same with the gradient spectrum formed on load from the spectrum itself,
 *       spec[rows][T][F][2], and the gradient of |spec|^power, grad_norm[rows][T][F] (the adjoint of
 *       functional.py:116-128 folded in: Spectrogram's backward in one pass, no gradient spectrum in memory).
 *     tac_spectrogram_backward_f32: the same with NO spectrum in memory: the frame is re-read from the waveform
 *       (wave / d exactly as given to (2)), transformed again in the kernel, and the spectrum values the norm's adjoint
 *       needs are formed from the FFT's exchange area while the inverse's operands are gathered.
 *     tac_spectrogram_backward_ola_f32: the whole adjoint of (2) — tac_spectrogram_backward_f32 + tac_overlap_add_f32 —
 *       for fft_length 256 / 512 / 1024 / 2048 with a hop that is a multiple of fft_length / 16, and fft_length 400 with
 *       50 <= hop <= 400 and hop, centre padding multiples of 4 (TAC_E_UNSUPPORTED otherwise): every wave walks runs of consecutive frames and keeps their overlap-add in LDS, so no frame gradients
 *       exist in memory.  `workspace`
 *       (device) must hold tac_spectrogram_backward_ola_workspace(d) bytes (that call returns a negative TAC_E_* code
 *       for geometries the form does not cover); grad_wave[r][j] at grad_wave + r * grad_row_stride + j.
 *     tac_melspectrogram_backward_ola_f32: the same for the mel chain (layers.py:333-339, functional.py:183-184): `grad_mel`
 *       is the gradient of the (linear) mel values, (rows, n_frames, n_mels) frame-major, and the filterbank stage's adjoint
 *       — two multiply-adds per bin through the table of tac_filterbank_adjoint_pack — is formed inside the kernel, per
 *       frame: the gradient of the power spectrogram never exists in memory.  fft_length 2048 (n_mels <= 256), 512 / 1024 (hop = N/8, N/4, N/2) and 400
 *       (n_mels <= 128), banks with at most two non-zero weights per bin; TAC_E_UNSUPPORTED otherwise (callers then run tac_apply_filterbank_adjoint_f32 +
 *       tac_spectrogram_backward_ola_f32).  Same workspace as tac_spectrogram_backward_ola_f32.
 *     tac_overlap_add_f32: adjoint of framing + padding: grad_wave[r][j] = sum of grad_frames over every (frame, tap)
 *       that read sample j, reflect / replicate / circular images included (a gather: deterministic, no atomics).
 *     tac_complex_norm_backward_f32: grad_z[i] = grad_out[i] * power * |z_i|^(power-2) * z_i (0 where z_i == 0),
 *       functional.py:126-128.
 *     tac_amplitude_to_db_backward_f32: grad_x = grad_out * 20 / (ln 10 * x) where x^2 >= amin, else 0,
 *       functional.py:291-296.
 *     tac_magphase_backward_f32 (round 6): gradient of magphase / angle (functional.py:187-201): grad_z[i] = grad_mag[i] * power *
 *       |z_i|^(power-2) * z_i + grad_phase[i] * (-im_i, re_i) / |z_i|^2, 0 where z_i == 0; grad_mag or grad_phase may be NULL.
 *     tac_db_to_amplitude_backward_f32 (round 6): grad_x = grad_out * ln(10) / 20 * (10^(x/10 + log10 ref))^0.5, functional.py:299-314.
 *     The filterbank stage's adjoint is (4) with the transposed matrix — or, for banks with at most two non-zero
 *     weights per bin (every triangular mel bank), tac_apply_filterbank_adjoint_f32: grad_spec[i][f] = w0[f] *
 *     grad_mel[i][band0[f]] + w1[f] * grad_mel[i][band1[f]] over i < rows*T frame-major rows, with the per-bin table
 *     built on the device by tac_filterbank_adjoint_pack (table: 16 * n_freqs + 16 bytes, device; *max_nonzeros_host
 *     receives the non-zero count of the fullest bin — the table is only valid when that is <= 2; synchronous). */
int tac_stft_backward_f32(const float* grad_spec, const float* window, const tac_stft_desc* d,
                          float* grad_frames, void* stream);
int tac_stft_norm_backward_f32(const float* spec, const float* grad_norm, float power, const float* window,
                               const tac_stft_desc* d, float* grad_frames, void* stream);
int tac_spectrogram_backward_f32(const float* wave, const float* window, const tac_stft_desc* d,
                                 const float* grad_norm, float power, float* grad_frames, void* stream);
int64_t tac_spectrogram_backward_ola_workspace(const tac_stft_desc* d);
int tac_spectrogram_backward_ola_f32(const float* wave, const float* window, const tac_stft_desc* d,
                                     const float* grad_norm, float power, void* workspace, int64_t workspace_bytes,
                                     float* grad_wave, int64_t grad_row_stride, void* stream);
int tac_melspectrogram_backward_ola_f32(const float* wave, const float* window, const tac_stft_desc* d,
                                        const float* grad_mel, int32_t n_mels, const void* adjoint_table,
                                        int32_t n_freqs, float power, void* workspace, int64_t workspace_bytes,
                                        float* grad_wave, int64_t grad_row_stride, void* stream);
/* fft_length 400 (frame gradients through memory, then tac_overlap_add_f32): tac_spectrogram_backward_f32 with the filterbank
 * adjoint of the mel chain formed inside the kernel from grad_mel (rows, n_frames, n_mels <= 128) and the
 * tac_filterbank_adjoint_pack table; TAC_E_UNSUPPORTED for other sizes. */
int tac_melspectrogram_backward_f32(const float* wave, const float* window, const tac_stft_desc* d,
                                    const float* grad_mel, int32_t n_mels, const void* adjoint_table, int32_t n_freqs,
                                    float power, float* grad_frames, void* stream);
int tac_filterbank_adjoint_pack(const float* fb, int32_t n_freqs, int32_t n_mels, void* table,
                                int32_t* max_nonzeros_host, void* stream);
int tac_apply_filterbank_adjoint_f32(const float* grad_mel, int64_t rows_times_frames, int32_t n_mels,
                                     const void* table, int32_t n_freqs, float* grad_spec, void* stream);
int tac_overlap_add_f32(const float* grad_frames, const tac_stft_desc* d, float* grad_wave,
                        int64_t grad_row_stride, void* stream);
int tac_complex_norm_backward_f32(const float* z, const float* grad_out, int64_t n, float power,
                                  float* grad_z, void* stream);
/* (9b) The general gradient routes (every fft_length, two-sided outputs, gradients of the window and the filterbank —
 *     functional.py:99-107, 183-184 differentiate through every argument):
 *     tac_overlap_add_f32 above takes ANY fft_length (framing only).
 *     tac_fold_twosided_f32: gradient of a two-sided output grad[frames][n_fft][width] (width 2: complex pairs, 1: |X|^p)
 *       folded onto the n_fft/2+1 one-sided bins: out[k] = grad[k] + grad[n_fft-k] (imaginary parts: minus).
 *     tac_window_grad_partials / tac_window_grad_f32: with grad_frames_unwindowed[rows][T][n_fft] the gradient w.r.t.
 *       the windowed frames (tac_stft_backward_f32 run with a window of ones), partial[p][n] = the sum over the p-th
 *       chunk of (row, frame) of grad_frames * padded signal; tac_sum_slabs_f32 adds the n_partials rows up.
 *     tac_sum_slabs_f32: out[i] = sum_s x[s][i] in slab order. */
int tac_fold_twosided_f32(const float* grad, int64_t n_frames_total, int32_t n_fft, int32_t width, float* out,
                          void* stream);
int64_t tac_window_grad_partials(const tac_stft_desc* d);
int tac_window_grad_f32(const float* grad_frames_unwindowed, const float* wave, const tac_stft_desc* d,
                        float* partial, int64_t n_partials, void* stream);
int tac_sum_slabs_f32(const float* x, int64_t n_slabs, int64_t slab_elems, float* out, void* stream);
int tac_amplitude_to_db_backward_f32(const float* x, const float* grad_out, int64_t n, float amin,
                                     float* grad_x, void* stream);
int tac_magphase_backward_f32(const float* z, const float* grad_mag, const float* grad_phase, int64_t n, float power,
                              float* grad_z, void* stream);
int tac_db_to_amplitude_backward_f32(const float* x, const float* grad_out, int64_t n, float ref, float* grad_x, void* stream);

/* (10) hpss, beta_hpss.py:35-127: median-filter harmonic / percussive separation of a magnitude
 *      spectrogram.  mag element (r, f, t) at mag[r*stride_r + f*stride_f + t*stride_t]; the four outputs use the same
 *      strides.  kernel_f (percussive filter, along frequency) and kernel_t (harmonic filter, along time) odd, <= 63
 *      (TAC_E_UNSUPPORTED above; equal widths 9 ... 31 are one launch, every other combination two);
 *      reflect padding (needs kernel/2 < size: TAC_E_SHORT_INPUT otherwise); masks soft ((h+eps)/(h+p+eps), eps 1e-6) or
 *      hard (1.0 / 0.0); harm / perc may both be NULL (masks only).  The outputs must not overlap mag or one another
 *      (the two-launch route parks the first launch's medians in mask_perc): overlapping address ranges are
 *      refused with TAC_E_INVALID.  The test is conservative: it compares the whole address SPANS of the five tensors (first to last
 *      element touched), so interleaved outputs that never share an element — channel slices of one stacked buffer — are refused
 *      as well, and it is skipped when a stride is negative. */
int tac_hpss_f32(const float* mag, int64_t rows, int32_t n_freqs, int32_t n_frames, int64_t stride_r,
                 int64_t stride_f, int64_t stride_t, int32_t kernel_f, int32_t kernel_t, float power,
                 int hard, float* harm, float* perc, float* mask_harm, float* mask_perc, void* stream);
/* ... and its gradient with respect to mag (round 6): the four incoming gradients (each may be NULL) and grad_mag use mag's strides;
 *      grad_mag ZERO-INITIALISED by the caller (the kernel scatters with float atomics: the gradient of a median goes to the element it
 *      selected; the sum order, hence the last bits, are not reproducible from run to run).  hard != 0: the masks are not differentiable,
 *      only harm = mag * mask / perc = mag * mask are. */
int tac_hpss_backward_f32(const float* mag, int64_t rows, int32_t n_freqs, int32_t n_frames, int64_t stride_r, int64_t stride_f,
                          int64_t stride_t, int32_t kernel_f, int32_t kernel_t, float power, int hard, const float* grad_harm,
                          const float* grad_perc, const float* grad_mask_harm, const float* grad_mask_perc, float* grad_mag,
                          void* stream);

/* (11) Diagnostics (bench.py's roofline object; no effect on results).
 *      tac_last_route: name of the kernel instantiation the calling thread's last fused-chain launch (3b / 3c at fft_length
 *        2048) or last fft_length-2048 STFT / spectrogram launch (1, 2: "stft_ring3_kernel<...>" / "stft_stream3_kernel<...>"; fft_length >= 8192: "stft_big_kernel<...>") took, as rocprofv3 prints it, e.g. "melspec_stream3_kernel<1024, 16, true, 0, 14, 12>" ("" before any).
 *      tac_debug_clock_probe: while `buf` (DEVICE uint64[2 * capacity_pairs]) is set for the calling thread, every workgroup
 *        b < capacity_pairs of those launches records buf[2b] = shader cycles (s_memtime) and buf[2b + 1] = ticks of the
 *        100 MHz constant clock (s_memrealtime) its wave 0 spent in the frame loop: cycles / ticks x 100 MHz is the shader
 *        clock the kernel actually ran at.  NULL clears it.  Four scalar instructions and one 16-byte store per workgroup. */
const char* tac_last_route(void);
int tac_debug_clock_probe(uint64_t* buf, int32_t capacity_pairs);

/* (12) Route selection of the fft_length-2048 kernels (process-wide; results agree to float32 rounding, ~2e-7 of a frame's
 *      largest bin): where the 1024-point transform of a frame runs.  mode 0 = VALU (radix 16 . 16 . 4 through the LDS,
 *      melspec_stream3_kernel / stft_ring3_kernel: rounds 3 - 5), 1 = matrix pipe (two chained 32 x 32 complex DFT products on
 *      v_mfma_f32_32x32x16_f16 with fp16 hi / lo operand pairs, melspec_mfma_kernel: round 6), -1 = the default (environment
 *      TAC_FFT_PIPE=valu|mfma, else the build's default).  Returns the previous mode. */
int tac_set_fft_pipe(int mode);

#ifdef __cplusplus
}
#endif
#endif /* TAC_AMD_H */
