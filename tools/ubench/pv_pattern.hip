// pv_pattern.hip — the phase vocoder's ACCESS PATTERN with no arithmetic: 256 x 1025 series, 313 input frames -> 241 output
// frames (rate 1.3); a lane walks the time axis of BINS consecutive bins (8 x BINS bytes per access), a wave's accesses are one
// contiguous piece of a frame-major row (8 200-byte rows), DEPTH input frames in flight; one input frame is read per output step (the
// kernel reads 313 frames for 241 steps at rate 1.3: its own arithmetic-free form is -DTAC_PV_ABL_COPY).  Is the kernel's floor
// a property of 512-byte pieces (one bin per lane), i.e. would wider lanes move it?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pv_pattern.hip -o tools/ubench/build/pv_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int BINS> struct Vec;
template <> struct Vec<1> { typedef float t __attribute__((ext_vector_type(2), aligned(8))); };
template <> struct Vec<2> { typedef float t __attribute__((ext_vector_type(4), aligned(8))); };

template <int BINS, int DEPTH, int THREADS>
__global__ void __launch_bounds__(THREADS) pv_k(const float* __restrict__ in, float* __restrict__ out, int rows, int F, int T,
                                                int n_out, const int* __restrict__ idx1) {
    typedef typename Vec<BINS>::t V;
    const int per_row = (F + BINS - 1) / BINS;                       // lane slots per row (the last one of an odd row is half used)
    const long long sid = (long long)blockIdx.x * THREADS + threadIdx.x;
    if (sid >= (long long)rows * per_row) return;
    const long long row = sid / per_row;
    int f = (int)(sid - row * per_row) * BINS;
    if (f + BINS > F) f = F - BINS;                                  // (probe: the last slot overlaps its neighbour)
    const float* base = in + (row * T * (long long)F + f) * 2;
    float* o = out + (row * n_out * (long long)F + f) * 2;
    V ahead[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) ahead[k] = *reinterpret_cast<const V*>(base + (long long)idx1[k < n_out ? k : n_out - 1] * F * 2);
    for (int i0 = 0; i0 < n_out; i0 += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const int i = i0 + k;
            if (i >= n_out) break;
            V cur = ahead[k];
            const int nx = i + DEPTH < n_out ? i + DEPTH : n_out - 1;
            ahead[k] = *reinterpret_cast<const V*>(base + (long long)idx1[nx] * F * 2);
            cur = cur * 1.0001f;
            *reinterpret_cast<V*>(o) = cur;
            o += 2 * (long long)F;
        }
    }
}

template <int BINS, int DEPTH, int THREADS>
static void run(const char* name, const float* in, float* out, const int* idx1, int rows, int F, int T, int n_out) {
    const int per_row = (F + BINS - 1) / BINS;
    const long long lanes = (long long)rows * per_row;
    const unsigned blocks = (unsigned)((lanes + THREADS - 1) / THREADS);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((pv_k<BINS, DEPTH, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, rows, F, T, n_out, idx1);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < 60; ++i) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((pv_k<BINS, DEPTH, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, in, out, rows, F, T, n_out, idx1);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float t;
        CHECK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)rows * F * 8.0 * (n_out + n_out);    // n_out frames read (the second frame of every step), n_out written
    printf("%-34s median %.4f ms   %.2f TB/s of %.0f MB\n", name, ms[30], bytes / ms[30] / 1e9, bytes / 1e6);
}

#include <algorithm>
int main() {
    const int rows = 256, F = 1025, T = 313;
    std::vector<int> h;
    for (int i = 0;; ++i) {
        const float t = (float)i * 1.3f;
        if (t >= (float)T) break;
        int i1 = (int)t + 1;
        h.push_back(i1 < T ? i1 : T - 1);
    }
    const int n_out = (int)h.size();
    float *in, *out;
    int* idx1;
    CHECK(hipMalloc(&in, (size_t)rows * T * F * 8));
    CHECK(hipMalloc(&out, (size_t)rows * n_out * F * 8));
    CHECK(hipMalloc(&idx1, n_out * sizeof(int)));
    CHECK(hipMemset(in, 0, (size_t)rows * T * F * 8));
    CHECK(hipMemcpy(idx1, h.data(), n_out * sizeof(int), hipMemcpyHostToDevice));
    printf("n_out %d\n", n_out);
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 4, 256>("1 bin / lane, 4 in flight", in, out, idx1, rows, F, T, n_out);
        run<1, 8, 256>("1 bin / lane, 8 in flight", in, out, idx1, rows, F, T, n_out);
        run<2, 4, 256>("2 bins / lane, 4 in flight", in, out, idx1, rows, F, T, n_out);
        run<2, 8, 256>("2 bins / lane, 8 in flight", in, out, idx1, rows, F, T, n_out);
        run<2, 4, 128>("2 bins / lane, 4, 128-thread WGs", in, out, idx1, rows, F, T, n_out);
        run<1, 4, 128>("1 bin / lane, 4, 128-thread WGs", in, out, idx1, rows, F, T, n_out);
        run<1, 4, 64>("1 bin / lane, 4, 64-thread WGs", in, out, idx1, rows, F, T, n_out);
    }
    return 0;
}
