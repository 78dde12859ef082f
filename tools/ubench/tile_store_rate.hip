// tile_store_rate.hip — what the memory system gives hpss's STORE pattern alone: 256 x 313 rows of 1025 floats (4100-byte pitch) in each of
// NARR result arrays; a 256-thread workgroup writes one tile = ROWS rows x SEG bytes (ROWS x SEG = 16 KB per array), 16-byte stores, every
// store instruction covering whole segments (the kernel's transpose does that).  Tiles per XCD contiguous (block b -> XCD b % 8).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tile_store_rate.hip -o tools/ubench/build/tile_store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4), aligned(4)));

template <int SEGF /* floats per segment */, int NARR>
__global__ void __launch_bounds__(256) tiles_k(float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o3,
                                               int NA, int NB, int tiles_a, int tiles_b, long long total) {
    constexpr int ROWS = 4096 / SEGF;                        // rows per tile
    const long long per_xcd = (total + 7) / 8;
    const long long t = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= total) return;
    const int per_row = tiles_a * tiles_b;
    const long long row = t / per_row;
    const int rem = (int)(t - row * per_row);
    const int a0 = (rem / tiles_b) * ROWS, b0 = (rem % tiles_b) * SEGF;
    float* outs[4] = {o0, o1, o2, o3};
    constexpr int LPR = SEGF / 4;                            // lanes per row segment
    const int tid = threadIdx.x;
#pragma unroll
    for (int arr = 0; arr < NARR; ++arr)
#pragma unroll
        for (int k = 0; k < 4; ++k) {                        // 1024 chunks of 16 bytes per tile and array
            const int c = tid + 256 * k;
            const int r = c / LPR, col = (c - r * LPR) * 4;
            const int a = a0 + r, b = b0 + col;
            if (a < NA && b + 3 < NB) {
                const v4 val = {(float)a, (float)b, (float)arr, 1.0f};
                *reinterpret_cast<v4*>(outs[arr] + row * (long long)NA * NB + (long long)a * NB + b) = val;
            }
        }
}

template <int SEGF, int NARR>
static void run(float* const* o, int rows, int NA, int NB) {
    constexpr int ROWS = 4096 / SEGF;
    const int ta = (NA + ROWS - 1) / ROWS, tb = (NB + SEGF - 1) / SEGF;
    const long long total = (long long)rows * ta * tb;
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((tiles_k<SEGF, NARR>), dim3(grid), dim3(256), 0, 0, o[0], o[1], o[2], o[3], NA, NB, ta, tb, total);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < 40; ++i) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((tiles_k<SEGF, NARR>), dim3(grid), dim3(256), 0, 0, o[0], o[1], o[2], o[3], NA, NB, ta, tb, total);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float t;
        CHECK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)rows * NA * NB * 4.0 * NARR;
    printf("%4d-byte segments x %3d rows, %d arrays: median %.4f ms  %.2f TB/s of %.0f MB\n", SEGF * 4, ROWS, NARR, ms[20], bytes / ms[20] / 1e9, bytes / 1e6);
}

int main() {
    const int rows = 256, NA = 313, NB = 1025;               // frame-major |X|^2: A = frames, B = bins
    float* o[4];
    for (int i = 0; i < 4; ++i) CHECK(hipMalloc(&o[i], (size_t)rows * NA * NB * 4));
    for (int rep = 0; rep < 2; ++rep) {
        run<64, 4>(o, rows, NA, NB);
        run<128, 4>(o, rows, NA, NB);
        run<256, 4>(o, rows, NA, NB);
        run<1024, 4>(o, rows, NA, NB);
        run<64, 2>(o, rows, NA, NB);
        run<256, 2>(o, rows, NA, NB);
        run<64, 1>(o, rows, NA, NB);
        run<1024, 1>(o, rows, NA, NB);
    }
    return 0;
}
