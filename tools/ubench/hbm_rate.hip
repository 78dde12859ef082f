// hbm_rate.hip — what this box's HBM delivers for the STFT stage's traffic mix: write-only (fill), read-only, and
// "read 1 : write 4" (the complex STFT at 2048/512: 2 KB in, 8.2 KB out per frame).  16-byte accesses, grid-stride.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_rate.hip -o tools/ubench/build/hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) fill_k(v4* out, long long n4, float s) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = v4{s, s, s, s};
}
__global__ void __launch_bounds__(256) read_k(const v4* in, long long n4, float* sink) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    v4 acc = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += in[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}
// each thread: read one 16-byte element, write four (disjoint output slab): the 1:4 mix
__global__ void __launch_bounds__(256) mix_k(const v4* in, v4* out, long long n4_in) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4_in; i += stride) {
        const v4 x = in[i];
        out[i] = x;
        out[i + n4_in] = x;
        out[i + 2 * n4_in] = x;
        out[i + 3 * n4_in] = x;
    }
}

template <class F>
float time_ms(F f) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 10;
}

int main() {
    const long long out_bytes = 657LL << 20, in_bytes = out_bytes / 4;
    v4 *in, *out;
    float* sink;
    CHECK(hipMalloc(&in, in_bytes));
    CHECK(hipMalloc(&out, out_bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(in, 0, in_bytes));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    for (int per_cu : {2, 4, 8, 16}) {
        const int blocks = cus * per_cu;
        float t1 = time_ms([&] { hipLaunchKernelGGL(fill_k, dim3(blocks), dim3(256), 0, 0, out, out_bytes / 16, 1.0f); });
        float t2 = time_ms([&] { hipLaunchKernelGGL(read_k, dim3(blocks), dim3(256), 0, 0, out, out_bytes / 16, sink); });
        float t3 = time_ms([&] { hipLaunchKernelGGL(mix_k, dim3(blocks), dim3(256), 0, 0, in, out, in_bytes / 16); });
        printf("%2d WG/CU: fill %.3f ms = %.2f TB/s | read %.3f ms = %.2f TB/s | read1:write4 %.3f ms = %.2f TB/s total\n", per_cu,
               t1, out_bytes / t1 / 1e9, t2, out_bytes / t2 / 1e9, t3, (out_bytes + in_bytes) / t3 / 1e9);
    }
    return 0;
}
