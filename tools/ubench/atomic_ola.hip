// atomic_ola.hip — can the overlap-add of the STFT backward be a scatter of hardware float atomics?
// 80 128 frames x 2048 samples, hop 512 (every output sample receives four adds), one frame per wave-iteration,
// 32 global_atomic_add_f32 per lane, each instruction 256 contiguous bytes.  Prints ms per pass.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) scatter(float* __restrict__ out, int rows, int T, int L, int hop, int n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long total = (long long)rows * T;
    for (long long f = (long long)blockIdx.x * 4 + w; f < total; f += (long long)gridDim.x * 4) {
        const int row = (int)(f / T), t = (int)(f - (long long)row * T);
        float* base = out + (long long)row * L;
        const int s0 = t * hop - n / 2;
#pragma unroll 8
        for (int j = 0; j < n / 64; ++j) {
            int p = s0 + lane + 64 * j;
            p = p < 0 ? -p : (p >= L ? 2 * (L - 1) - p : p);
            unsafeAtomicAdd(base + p, 1.0f + (float)j);
        }
    }
}
__global__ void __launch_bounds__(256) plain(float* __restrict__ out, int rows, int T, int L, int hop, int n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long total = (long long)rows * T;
    for (long long f = (long long)blockIdx.x * 4 + w; f < total; f += (long long)gridDim.x * 4) {
        const int row = (int)(f / T), t = (int)(f - (long long)row * T);
        float* base = out + (long long)row * L;
        const int s0 = t * hop - n / 2;
#pragma unroll 8
        for (int j = 0; j < n / 64; ++j) {
            int p = s0 + lane + 64 * j;
            p = p < 0 ? -p : (p >= L ? 2 * (L - 1) - p : p);
            base[p] = 1.0f + (float)j;
        }
    }
}
int main() {
    const int rows = 256, L = 160000, hop = 512, n = 2048, T = 313;
    float* out; hipMalloc(&out, (size_t)rows * L * 4); hipMemset(out, 0, (size_t)rows * L * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int which = 0; which < 2; ++which) {
        for (int i = 0; i < 20; ++i) { if (which) hipLaunchKernelGGL(scatter, dim3(512), dim3(256), 0, 0, out, rows, T, L, hop, n); else hipLaunchKernelGGL(plain, dim3(512), dim3(256), 0, 0, out, rows, T, L, hop, n); }
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 50; ++i) { if (which) hipLaunchKernelGGL(scatter, dim3(512), dim3(256), 0, 0, out, rows, T, L, hop, n); else hipLaunchKernelGGL(plain, dim3(512), dim3(256), 0, 0, out, rows, T, L, hop, n); }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%s: %.4f ms per pass (%d frames x %d samples)\n", which ? "atomic scatter" : "plain stores  ", ms / 50, rows * T, n);
    }
    return 0;
}
