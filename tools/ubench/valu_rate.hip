// valu_rate.hip — micro-benchmark: issue rate of scalar vs packed f32 VALU ops on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/build/valu_rate && ./valu_rate
// Each wave runs ITER iterations of 32 independent instructions of one kind (16 accumulators x 2), one or two
// waves per SIMD, all CUs busy; prints cycles per wave-instruction per SIMD (shader clock from s_memtime deltas
// is not used — wall time x nominal 2.4 GHz is reported next to the raw ns so the ratio between kinds is what matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITER = 16384;

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* out, float seed) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = v2{seed + i, seed - i};
    v2 b = v2{seed * 0.5f, seed * 0.25f}, c = v2{1e-3f, 2e-3f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (KIND == 0) {          // 2 x v_fma_f32
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(b.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(b.y), "v"(c.y));
            } else if constexpr (KIND == 1) {   // 1 x v_pk_fma_f32 (same flops as KIND 0)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(c));
            } else if constexpr (KIND == 2) {   // 2 x v_add_f32
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(b.x));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].y) : "v"(b.y));
            } else if constexpr (KIND == 3) {   // 1 x v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(b));
            } else if constexpr (KIND == 4) {   // 1 x v_pk_mul_f32 with op_sel broadcast + neg (complex-multiply shape)
                asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "+v"(acc[i]) : "v"(b));
            } else if constexpr (KIND == 6) {   // 2 x v_fmac_f32 (VOP2 encoding)
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(b.x), "v"(c.x));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].y) : "v"(b.y), "v"(c.y));
            } else if constexpr (KIND == 7) {   // 2 x v_fma_f32 with an inline constant as one multiplicand
                asm volatile("v_fma_f32 %0, %1, 0.5, %0" : "+v"(acc[i].x) : "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, 0.5, %0" : "+v"(acc[i].y) : "v"(b.y));
            } else if constexpr (KIND == 8) {   // 2 x v_fma_f32 acc = acc*b + c (accumulator as multiplicand)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(b.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].y) : "v"(b.y), "v"(c.y));
            } else if constexpr (KIND == 9) {   // DEPENDENT chain of v_pk_add_f32 on one accumulator (latency, not rate)
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[0]) : "v"(b));
            } else if constexpr (KIND == 10) {  // DEPENDENT chain of v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(b), "v"(c));
            } else if constexpr (KIND == 11) {  // two interleaved dependent chains of v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(b), "v"(c));
            } else if constexpr (KIND == 12) {  // four interleaved dependent chains
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(b), "v"(c));
            } else if constexpr (KIND == 13) {  // v_pk_add_f32 with op_sel / neg modifiers (the rotate-add of the butterflies)
                asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(acc[i]) : "v"(b));
            } else if constexpr (KIND == 14) {  // v_pk_fma_f32 with an SGPR-pair multiplicand (constant twiddles)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "s"(c));
            } else if constexpr (KIND == 5) {   // 2 x v_mul_f32
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(b.x));
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i].y) : "v"(b.y));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int insts_per_iter, int waves_per_simd, float* d) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(rate_kernel<KIND>, grid, block, 0, 0, d, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rate_kernel<KIND>, grid, block, 0, 0, d, 1.0f);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double wave_insts_per_simd = (double)ITER * insts_per_iter * waves_per_simd;
    const double ns_per_inst = ms * 1e6 / wave_insts_per_simd;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instruction per SIMD  (= %.2f cycles @2.4GHz)  pair-of-f32-ops: %.2f cycles\n",
           name, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 2.4 * insts_per_iter / 16.0);
}

int main() {
    float* d;
    CHECK(hipMalloc(&d, 4096));
    for (int w = 1; w <= 2; ++w) {
        run<9>("dependent v_pk_add_f32 chain", 16, w, d);
        run<10>("dependent v_pk_fma_f32 chain", 16, w, d);
        run<11>("2 interleaved pk_fma chains", 16, w, d);
        run<12>("4 interleaved pk_fma chains", 16, w, d);
        run<13>("1x v_pk_add_f32 op_sel/neg", 16, w, d);
        run<14>("1x v_pk_fma_f32 sgpr operand", 16, w, d);
    }
    for (int w = 1; w <= 8; w *= 2) {
        run<0>("2x v_fma_f32", 32, w, d);
        run<1>("1x v_pk_fma_f32", 16, w, d);
        run<6>("2x v_fmac_f32", 32, w, d);
        run<7>("2x v_fma_f32 inline-const", 32, w, d);
        run<8>("2x v_fma_f32 acc*b+c", 32, w, d);
        run<2>("2x v_add_f32", 32, w, d);
        run<3>("1x v_pk_add_f32", 16, w, d);
        run<5>("2x v_mul_f32", 32, w, d);
        run<4>("1x v_pk_mul_f32 op_sel/neg", 16, w, d);
    }
    return 0;
}
