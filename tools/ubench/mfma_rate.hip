// mfma_rate.hip — micro-benchmark (round 6): what v_mfma_f32_32x32x16_f16 sustains per SIMD in the shapes melspec_mfma_kernel
// issues it, at ITS occupancy (W waves per SIMD, one workgroup per CU, every CU busy).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Every wave runs ITER bursts of 24 MFMAs:
//   PAT 0: one accumulator, back to back          PAT 1: two accumulators alternating (dr, di, dr, di ...)
//   PAT 2: twelve + twelve (dr x 12, di x 12)     PAT 3: four accumulators rotating
// ACC 0: accumulators in VGPRs ("v" constraints), 1: in AGPRs ("a").
// VALU = n: n independent v_pk_fma_f32 issued by the wave between bursts (the split / twiddle / R2C work of the real kernel).
// DEN 1: the fp16 operands are denormals (the lo halves of small samples are).
// Prints shader cycles per MFMA per SIMD (wall cycles of the slowest wave x 1 / (W x 24 x ITER)) — 32 is the pipe's rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 400;

template <int ACC>
__device__ __forceinline__ void mma(f16v& d, h8 a, h8 b) {
    if constexpr (ACC == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b));
}

// VK: kind of the filler: 0 v_pk_fma_f32, 1 v_fma_f32 (two per count: same flops), 2 v_fma_mixlo_f16, 3 v_pk_add_f32, 4 v_add_f32 x 2,
//     5 v_pk_mul_f32, 6 v_mov_b32
template <int VK>
__device__ __forceinline__ void filler(f2& x, f2 y) {
    if constexpr (VK == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y));
    else if constexpr (VK == 1) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x.x) : "v"(y.x)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x.y) : "v"(y.y)); }
    else if constexpr (VK == 2) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, 0 op_sel_hi:[0,0,0]" : "+v"(x.x) : "v"(y.x));
    else if constexpr (VK == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (VK == 4) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.x) : "v"(y.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.y) : "v"(y.y)); }
    else if constexpr (VK == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    else asm volatile("v_mov_b32 %0, %1" : "+v"(x.x) : "v"(y.x));
}

template <int W, int PAT, int ACC, int VALU, int DEN, int VK = 0, int ILV = 0>
__global__ void __launch_bounds__(W * 256, W) k(float* out, unsigned long long* cyc, int prio = 0) {
    const int lane = threadIdx.x & 63;
    if (prio) {                                        // static priorities by wave slot of the SIMD (waves w, w + 4, w + 8 share one)
        const int slot = (threadIdx.x >> 6) >> 2;
        if (slot == 0) __builtin_amdgcn_s_setprio(3);
        else if (slot == 1) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
    }
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (_Float16)(DEN ? 1e-7f * (float)(lane + i + j) : 0.01f * (float)((lane + i + j) & 15));
            b[i][j] = (_Float16)(DEN ? 2e-7f * (float)(lane - i + j) : 0.02f * (float)((lane - i + j) & 7));
        }
    f16v d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) d[i][j] = 0.f;
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2{(float)lane, (float)i};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (ILV > 0) {            // ILV fillers after every MFMA of the wave's own stream (accumulators by PAT: 1 two alternating, 2 twelve + twelve, 3 four rotating, 0 one)
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                mma<ACC>(d[PAT == 1 ? (i & 1) : (PAT == 2 ? i / 12 : (PAT == 3 ? (i & 3) : 0))], a[i & 3], b[(i >> 2) & 3]);
#pragma unroll
                for (int j = 0; j < ILV; ++j) filler<VK>(v[(i * ILV + j) & 7], v[(i * ILV + j + 1) & 7]);
            }
        } else if constexpr (PAT == 0) {
#pragma unroll
            for (int i = 0; i < 24; ++i) mma<ACC>(d[0], a[i & 3], b[(i >> 2) & 3]);
        } else if constexpr (PAT == 1) {
#pragma unroll
            for (int i = 0; i < 24; ++i) mma<ACC>(d[i & 1], a[i & 3], b[(i >> 2) & 3]);
        } else if constexpr (PAT == 2) {
#pragma unroll
            for (int i = 0; i < 24; ++i) mma<ACC>(d[i / 12], a[i & 3], b[(i >> 2) & 3]);
        } else {
#pragma unroll
            for (int i = 0; i < 24; ++i) mma<ACC>(d[i & 3], a[i & 3], b[(i >> 2) & 3]);
        }
#pragma unroll
        for (int i = 0; i < VALU; ++i) filler<VK>(v[i & 7], v[(i + 1) & 7]);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += d[i][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// ROLES: the waves of a SIMD do different things — wave slot 0 of every SIMD (threadIdx / 64 < 4) only MFMA bursts, the other W - 1 only
// fillers (VALU per burst, kind VK): do the matrix pipe and the vector ALU of ONE SIMD run side by side when different waves feed them?
template <int W, int VALU, int VK>
__global__ void __launch_bounds__(W * 256, W) k_roles(float* out, unsigned long long* cyc, int mfma_on, int valu_on) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (_Float16)(0.01f * (float)((lane + i + j) & 15));
            b[i][j] = (_Float16)(0.02f * (float)((lane - i + j) & 7));
        }
    f16v d[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) d[i][j] = 0.f;
    f2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2{(float)lane, (float)i};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wv < 4) {
        if (mfma_on)
            for (int it = 0; it < ITER; ++it) {
#pragma unroll
                for (int i = 0; i < 24; ++i) mma<0>(d[i / 12], a[i & 3], b[(i >> 2) & 3]);
            }
    } else if (valu_on) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < VALU; ++i) filler<VK>(v[i & 7], v[(i + 1) & 7]);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += d[i][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wv] = t1 - t0;
}
template <int W, int VALU, int VK>
void run_roles(const char* what) {
    const int blocks = 256, waves = W * 4;
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * waves * 64 * 4));
    CHECK(hipMalloc(&cyc, (size_t)blocks * waves * 8));
    std::vector<unsigned long long> h((size_t)blocks * waves);
    unsigned long long res[3][2];
    for (int mode = 0; mode < 3; ++mode) {          // 0: MFMA waves alone, 1: filler waves alone, 2: both
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_roles<W, VALU, VK>), dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, mode != 1, mode != 0);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long mm = 0, mv = 0;
        for (size_t i = 0; i < h.size(); ++i) {
            const bool is_m = (i % waves) < 4;
            if (is_m) mm = h[i] > mm ? h[i] : mm; else mv = h[i] > mv ? h[i] : mv;
        }
        res[mode][0] = mm; res[mode][1] = mv;
    }
    printf("%-40s W=%d  MFMA wave alone %llu cycles, filler waves alone %llu | together: MFMA wave %llu, filler waves %llu (sum of the two alone: %llu)\n",
           what, W, res[0][0], res[1][1], res[2][0], res[2][1], res[0][0] + res[1][1]);
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
}

template <int W, int PAT, int ACC, int VALU, int DEN, int VK = 0, int ILV = 0>
void run(const char* what, int prio = 0) {
    const int blocks = 256, waves = W * 4;
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * waves * 64 * 4));
    CHECK(hipMalloc(&cyc, (size_t)blocks * waves * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<W, PAT, ACC, VALU, DEN, VK, ILV>), dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, prio);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<W, PAT, ACC, VALU, DEN, VK, ILV>), dim3(blocks), dim3(waves * 64), 0, 0, out, cyc, prio);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * waves);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    double sum = 0;
    for (auto c : h) { mx = c > mx ? c : mx; sum += (double)c; }
    const double per = (double)mx / ((double)W * 24 * ITER);
    printf("%-44s W=%d  %.1f cycles per MFMA per SIMD (slowest wave %llu cycles, mean %.0f), kernel %.3f ms -> %.2f GHz, %.0f TFLOP/s\n", what, W, per, mx,
           sum / h.size(), ms, (double)mx / (ms * 1e6), 256.0 * 4 * W * 24 * ITER * 32768.0 / (ms * 1e9));
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
}

int main() {
    run<1, 0, 0, 0, 0>("1 acc back to back, VGPR");
    run<1, 1, 0, 0, 0>("2 acc alternating, VGPR");
    run<1, 2, 0, 0, 0>("12 + 12, VGPR");
    run<1, 3, 0, 0, 0>("4 acc rotating, VGPR");
    run<1, 0, 1, 0, 0>("1 acc back to back, AGPR");
    run<1, 1, 1, 0, 0>("2 acc alternating, AGPR");
    run<1, 3, 1, 0, 0>("4 acc rotating, AGPR");
    run<2, 1, 0, 0, 0>("2 acc alternating, VGPR");
    run<3, 1, 0, 0, 0>("2 acc alternating, VGPR");
    run<3, 2, 0, 0, 0>("12 + 12, VGPR");
    run<3, 1, 1, 0, 0>("2 acc alternating, AGPR");
    run<1, 1, 0, 0, 1>("2 acc alternating, VGPR, denormal operands");
    run<3, 1, 0, 0, 1>("2 acc alternating, VGPR, denormal operands");
    run<1, 1, 0, 100, 0>("2 acc alt, VGPR + 100 pk_fma per burst");
    run<3, 1, 0, 100, 0>("2 acc alt, VGPR + 100 pk_fma per burst");
    run<3, 1, 0, 200, 0>("2 acc alt, VGPR + 200 pk_fma per burst");
    run<3, 1, 1, 200, 0>("2 acc alt, AGPR + 200 pk_fma per burst");
    run<3, 2, 0, 200, 0>("12 + 12, VGPR + 200 pk_fma per burst");
    run<3, 3, 0, 200, 0>("4 acc rot, VGPR + 200 pk_fma per burst");
    printf("---- kinds of filler, 3 waves per SIMD, between bursts (VALU count = pairs for the scalar kinds)\n");
    run<3, 1, 0, 200, 0, 1>("v_fma_f32 x 400 per burst");
    run<3, 1, 0, 200, 0, 2>("v_fma_mixlo_f16 x 200 per burst");
    run<3, 1, 0, 200, 0, 3>("v_pk_add_f32 x 200 per burst");
    run<3, 1, 0, 200, 0, 4>("v_add_f32 x 400 per burst");
    run<3, 1, 0, 200, 0, 5>("v_pk_mul_f32 x 200 per burst");
    run<3, 1, 0, 200, 0, 6>("v_mov_b32 x 200 per burst");
    printf("---- fillers inside the wave's own MFMA stream (after every MFMA), 1 and 3 waves per SIMD\n");
    run<1, 1, 0, 0, 0, 1, 2>("own stream: 4 v_fma_f32 per MFMA");
    run<1, 1, 0, 0, 0, 1, 4>("own stream: 8 v_fma_f32 per MFMA");
    run<1, 1, 0, 0, 0, 0, 2>("own stream: 2 v_pk_fma_f32 per MFMA");
    run<1, 1, 0, 0, 0, 0, 4>("own stream: 4 v_pk_fma_f32 per MFMA");
    run<1, 1, 0, 0, 0, 2, 4>("own stream: 4 v_fma_mixlo_f16 per MFMA");
    run<3, 1, 0, 0, 0, 1, 2>("own stream: 4 v_fma_f32 per MFMA");
    run<3, 1, 0, 0, 0, 1, 4>("own stream: 8 v_fma_f32 per MFMA");
    run<3, 1, 0, 0, 0, 0, 4>("own stream: 4 v_pk_fma_f32 per MFMA");
    run<3, 1, 0, 0, 0, 2, 4>("own stream: 4 v_fma_mixlo_f16 per MFMA");
    run<3, 1, 0, 0, 0, 2, 8>("own stream: 8 v_fma_mixlo_f16 per MFMA");
    printf("---- own stream, scalar fillers per MFMA by accumulator pattern (1 and 2 waves per SIMD)\n");
    run<1, 3, 0, 0, 0, 1, 3>("4 acc rotating: 6 v_fma_f32 per MFMA");
    run<1, 3, 0, 0, 0, 1, 4>("4 acc rotating: 8 v_fma_f32 per MFMA");
    run<1, 3, 0, 0, 0, 1, 5>("4 acc rotating: 10 v_fma_f32 per MFMA");
    run<1, 3, 0, 0, 0, 1, 6>("4 acc rotating: 12 v_fma_f32 per MFMA");
    run<1, 2, 0, 0, 0, 1, 3>("12 + 12: 6 v_fma_f32 per MFMA");
    run<1, 2, 0, 0, 0, 1, 4>("12 + 12: 8 v_fma_f32 per MFMA");
    run<1, 2, 0, 0, 0, 1, 5>("12 + 12: 10 v_fma_f32 per MFMA");
    run<1, 2, 0, 0, 0, 1, 6>("12 + 12: 12 v_fma_f32 per MFMA");
    run<1, 0, 0, 0, 0, 1, 5>("1 acc: 10 v_fma_f32 per MFMA");
    run<2, 3, 0, 0, 0, 1, 5>("4 acc rotating: 10 v_fma_f32 per MFMA");
    run<2, 2, 0, 0, 0, 1, 5>("12 + 12: 10 v_fma_f32 per MFMA");
    run<2, 2, 0, 0, 0, 1, 6>("12 + 12: 12 v_fma_f32 per MFMA");
    run<1, 2, 0, 0, 0, 2, 8>("12 + 12: 8 v_fma_mixlo_f16 per MFMA");
    run<1, 2, 0, 0, 0, 2, 10>("12 + 12: 10 v_fma_mixlo_f16 per MFMA");
    run<1, 2, 0, 0, 0, 6, 10>("12 + 12: 10 v_mov_b32 per MFMA");
    printf("---- identical waves (burst of 24 MFMAs + fillers), static priorities 3 / 2 / 1 by wave slot of the SIMD\n");
    run<3, 2, 0, 200, 0, 1>("v_fma_f32 x 400 per burst, no priorities", 0);
    run<3, 2, 0, 200, 0, 1>("v_fma_f32 x 400 per burst, priorities", 1);
    run<2, 2, 0, 200, 0, 1>("v_fma_f32 x 400 per burst, no priorities", 0);
    run<2, 2, 0, 200, 0, 1>("v_fma_f32 x 400 per burst, priorities", 1);
    run<3, 2, 0, 200, 0, 0>("v_pk_fma_f32 x 200 per burst, priorities", 1);
    printf("---- one MFMA-only wave + (W - 1) filler-only waves per SIMD\n");
    run_roles<2, 384, 1>("768 v_fma_f32 per burst, 1 filler wave");
    run_roles<2, 192, 0>("192 v_pk_fma_f32 per burst, 1 filler wave");
    run_roles<3, 192, 1>("384 v_fma_f32 per burst, 2 filler waves");
    run_roles<3, 96, 0>("96 v_pk_fma_f32 per burst, 2 filler waves");
    run_roles<3, 192, 2>("192 v_fma_mixlo_f16 per burst, 2 filler waves");
    return 0;
}
