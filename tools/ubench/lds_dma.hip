// lds_dma.hip — does gfx950's global_load_lds_dwordx4 (global -> LDS without registers) put lane l's 16 bytes at M0 + 16 l, and
// does it reach LDS offsets beyond 64 KB (M0 as a full LDS address)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma.hip -o tools/ubench/build/lds_dma && tools/ubench/build/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) probe(const float* __restrict__ in, float* __restrict__ out, int lds_byte_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    for (int i = t; i < 40 * 1024; i += 64) reinterpret_cast<float*>(smem)[i] = -1.0f;       // 160 KB of -1
    __syncthreads();
    // lanes in REVERSED global order: lane l fetches chunk 63 - l, so the LDS image tells lane order from address order
    __builtin_amdgcn_global_load_lds(in + 4 * (63 - t), (__attribute__((address_space(3))) void*)(smem + lds_byte_offset), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    __syncthreads();
    for (int i = t; i < 40 * 1024; i += 64) out[i] = reinterpret_cast<float*>(smem)[i];
}

int main() {
    const int n = 40 * 1024;
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *din, *dout;
    CHECK(hipMalloc(&din, 256 * 4));
    CHECK(hipMalloc(&dout, n * 4));
    CHECK(hipMemcpy(din, h.data(), 256 * 4, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int offs[] = {0, 4096, 65536 - 1024, 65536, 100000 / 16 * 16, 160 * 1024 - 1024};
    for (int off : offs) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, din, dout, off);
        CHECK(hipDeviceSynchronize());
        std::vector<float> o(n);
        CHECK(hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost));
        int touched = 0, first = -1, ok = 1;
        for (int i = 0; i < n; ++i)
            if (o[i] != -1.0f) { if (first < 0) first = i; ++touched; }
        for (int l = 0; l < 64 && first >= 0; ++l)
            for (int e = 0; e < 4; ++e)
                if (first + 4 * l + e >= n || o[first + 4 * l + e] != (float)(4 * (63 - l) + e)) ok = 0;
        printf("lds offset %6d: %3d floats written, first at byte %6d, lane l -> M0 + 16 l with its own chunk: %s\n", off, touched,
               first * 4, (touched == 256 && ok && first * 4 == off) ? "yes" : "NO");
    }
    return 0;
}
