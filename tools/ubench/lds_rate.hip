// lds_rate.hip — micro-benchmark: cost of the FFT kernels' LDS access shapes at THEIR occupancy
// (8 waves per CU = 2 per SIMD, 256 threads per workgroup x 2 workgroups, every CU busy).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_rate.hip -o tools/ubench/build/lds_rate
// Each wave owns an 8.7 KB slice (like one frame buffer) and repeats one block of 16 x 8-byte or 8 x 16-byte
// accesses per lane (8 KB per wave per block) followed by s_waitcnt lgkmcnt(0); prints wall cycles per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;
constexpr int SLICE = 2304;     // floats per wave (9 KB)

template <int KIND>
__global__ void __launch_bounds__(256, 2) lds_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) float smem[4 * SLICE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* base = smem + w * SLICE;
    for (int i = lane; i < SLICE; i += 64) base[i] = (float)i;
    __syncthreads();
    v2 a2[16];
    v4 a4[8];
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a2[i] = v2{(float)i, (float)lane};
#pragma unroll
    for (int i = 0; i < 8; ++i) a4[i] = v4{(float)i, (float)lane, 1.f, 2.f};
    for (int it = 0; it < ITER; ++it) {
        if constexpr (KIND == 0) {          // 16 x ds_read_b64, lanes contiguous, 65-complex row stride (the pass readback)
            const v2* p = reinterpret_cast<const v2*>(base) + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) a2[q] = p[q * 68];
        } else if constexpr (KIND == 1) {   // 8 x ds_read_b128, lane stride 144 B (read-contiguous layout)
            const v4* p = reinterpret_cast<const v4*>(base) + lane * 9;
#pragma unroll
            for (int q = 0; q < 8; ++q) a4[q] = p[q];
        } else if constexpr (KIND == 2) {   // 16 x ds_write_b64, per lane contiguous run of 16 at stride 17 (pass-0 write)
            v2* p = reinterpret_cast<v2*>(base) + lane * 17;
#pragma unroll
            for (int q = 0; q < 16; ++q) p[q] = a2[q];
        } else if constexpr (KIND == 3) {   // 16 x ds_write_b64, lanes contiguous per instruction (later-pass write)
            v2* p = reinterpret_cast<v2*>(base) + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) p[q * 68] = a2[q];
        } else if constexpr (KIND == 4) {   // 8 x ds_write_b128, lane stride 144 B
            v4* p = reinterpret_cast<v4*>(base) + lane * 9;
#pragma unroll
            for (int q = 0; q < 8; ++q) p[q] = a4[q];
        } else if constexpr (KIND == 5) {   // 8 x ds_read_b128, lanes contiguous per instruction
            const v4* p = reinterpret_cast<const v4*>(base) + lane;
#pragma unroll
            for (int q = 0; q < 8; ++q) a4[q] = p[q * 68];
        } else if constexpr (KIND == 6) {   // 16 x ds_read_b64 reversed lanes (R2C partner reads)
            const v2* p = reinterpret_cast<const v2*>(base) + (1100 - lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) a2[q] = p[-q * 68];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(a2[i]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a4[i]));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a2[i].x + a2[i].y;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a4[i].x + a4[i].w;
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <int KIND>
void run(const char* name, float* d) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(cus * 2), dim3(256), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(lds_kernel<KIND>, dim3(cus * 2), dim3(256), 0, 0, d);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    ms /= 5;
    const double ns_per_block = ms * 1e6 / ITER;
    printf("%-58s %.3f ms  %.1f ns per 8 KB block per wave (8 waves/CU)  = %.0f B/clk/CU @2.0GHz\n", name, ms, ns_per_block,
           8.0 * 8192.0 / (ns_per_block * 2.0));
}

int main() {
    float* d;
    CHECK(hipMalloc(&d, 4096));
    run<0>("16 x ds_read_b64  lanes contiguous (pass readback)", d);
    run<6>("16 x ds_read_b64  lanes reversed (R2C partner)", d);
    run<5>("8 x ds_read_b128  lanes contiguous", d);
    run<1>("8 x ds_read_b128  lane stride 144 B", d);
    run<2>("16 x ds_write_b64 lane stride 136 B (pass-0 write)", d);
    run<3>("16 x ds_write_b64 lanes contiguous (later-pass write)", d);
    run<4>("8 x ds_write_b128 lane stride 144 B", d);
    return 0;
}
