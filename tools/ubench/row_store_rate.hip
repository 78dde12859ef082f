// row_store_rate.hip — what the memory system gives the STFT kernels' ACCESS PATTERN with no arithmetic at all: one persistent
// workgroup per CU (12 or 16 waves), every wave takes frames from its workgroup's contiguous chunk, reads the frame's 2048
// samples (8-byte accesses; three quarters of them re-read from L1/L2: hop 512) and writes the frame's output row as 16-byte
// accesses — row lengths 8200 B (complex STFT), 8192 B (the same, 128-byte aligned), 4100 / 4096 B (power rows).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/row_store_rate.hip -o tools/ubench/build/row_store_rate
// Rows that are not a multiple of 16 bytes are written as floor(row / 16) 16-byte chunks starting at the row's first 16-byte
// boundary (the few head / tail bytes are skipped: this is a bandwidth probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

// ORDER 0: every workgroup walks its own contiguous chunk of frames (what the STFT kernels do: 256 far-apart write streams);
// ORDER 1 (round 5): frames dealt round-robin over ALL waves of the grid — turn i, workgroup c, wave w writes row
// i * (grid * WAVES) + c * WAVES + w: one tight window of grid * WAVES rows moving through the output.
template <int WAVES, bool NT, bool NEW_ONLY, int ORDER = 0>
__global__ void __launch_bounds__(WAVES * 64) rows_k(const float* __restrict__ in, char* __restrict__ out, int frames_per_row, int rows,
                                                     int row_bytes, long long in_row_stride) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ unsigned counter;
    if (threadIdx.x == 0) counter = WAVES;
    __syncthreads();
    const long long total = (long long)rows * frames_per_row;
    const long long chunk = (total + gridDim.x - 1) / gridDim.x;
    const long long begin = (long long)blockIdx.x * chunk;
    const int nloc = (int)((begin + chunk < total ? begin + chunk : total) - begin);
    int i = w;
    while (ORDER == 0 ? i < nloc : true) {
        unsigned ask = 0;
        if (lane == 0) ask = atomicAdd(&counter, 1u);
        // ORDER 2: the same with the workgroups of one XCD (blockIdx % 8) side by side: every XCD's L2 sees one contiguous window
        const long long slot = ORDER == 2 ? (long long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (long long)blockIdx.x;
        const long long gf = ORDER == 0 ? begin + i
                                        : (long long)(i / WAVES) * ((long long)gridDim.x * WAVES) + slot * WAVES + (i % WAVES);
        if (ORDER != 0 && gf >= total) break;
        const int r = (int)(gf / frames_per_row), f = (int)(gf - (long long)r * frames_per_row);
        const v2* src = reinterpret_cast<const v2*>(in + (long long)r * in_row_stride + (long long)f * 512);
        v2 v[16];
        if (NEW_ONLY) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = src[lane + 64 * q];                    // the hop's 512 new samples only
#pragma unroll
            for (int q = 4; q < 16; ++q) v[q] = v[q & 3];
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = src[lane + 64 * q];                   // the whole frame, like the kernels
        }
        char* row = out + gf * (long long)row_bytes;
        row += (16 - (reinterpret_cast<unsigned long long>(row) & 15)) & 15;
        const int nchunks = (row_bytes - 15) / 16;
        v4* dst = reinterpret_cast<v4*>(row);
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int c = lane + 64 * u;
            if (c < nchunks) {
                const v4 val = {v[(2 * u) & 15].x, v[(2 * u) & 15].y, v[(2 * u + 1) & 15].x, v[(2 * u + 1) & 15].y};
                if (NT) __builtin_nontemporal_store(val, &dst[c]);
                else dst[c] = val;
            }
        }
        i = __builtin_amdgcn_readfirstlane(ask);
    }
}

template <class F>
float time_ms(F f, int n) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f(i);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < n; ++i) f(i);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / n;
}

int main() {
    const int rows = 256, frames = 313, nbuf = 4;
    const long long in_row = 160000 + 2048;                 // (room for the last frame's read)
    float* in[nbuf];
    for (int b = 0; b < nbuf; ++b) {
        CHECK(hipMalloc(&in[b], rows * in_row * 4));
        CHECK(hipMemset(in[b], 0, rows * in_row * 4));
    }
    char* out;
    CHECK(hipMalloc(&out, (size_t)rows * frames * 8208 + 256));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int row_bytes[] = {8200, 8192, 4100, 4096};
    for (int rb : row_bytes) {
        const double bytes = (double)rows * frames * (2048.0 + ((rb - 15) / 16) * 16.0);
        auto run = [&](auto kern, int waves, const char* what) {
            float t = time_ms([&](int i) { hipLaunchKernelGGL(kern, dim3(cus), dim3(waves * 64), 0, 0, in[i % nbuf], out, frames, rows, rb, in_row); }, 40);
            printf("row %4d B  %-34s %.4f ms = %.2f TB/s\n", rb, what, t, bytes / t / 1e9);
        };
        run(rows_k<12, true, false>, 12, "12 waves, nt stores, frame reads");
        run(rows_k<12, false, false>, 12, "12 waves, plain stores, frame reads");
        run(rows_k<16, true, false>, 16, "16 waves, nt stores, frame reads");
        run(rows_k<16, true, true>, 16, "16 waves, nt stores, new-hop reads");
        run(rows_k<12, true, true>, 12, "12 waves, nt stores, new-hop reads");
        run(rows_k<12, true, true, 1>, 12, "12 waves, nt, new-hop, global order");
        run(rows_k<8, true, true, 1>, 8, " 8 waves, nt, new-hop, global order");
        run(rows_k<4, true, true, 1>, 4, " 4 waves, nt, new-hop, global order");
        run(rows_k<16, true, true, 1>, 16, "16 waves, nt, new-hop, global order");
        run(rows_k<12, true, true, 2>, 12, "12 waves, nt, new-hop, global order by XCD");
        run(rows_k<8, true, true, 2>, 8, " 8 waves, nt, new-hop, global order by XCD");
        run(rows_k<16, true, true, 2>, 16, "16 waves, nt, new-hop, global order by XCD");
        run(rows_k<12, false, true, 1>, 12, "12 waves, plain, new-hop, global order");
        run(rows_k<8, true, true, 0>, 8, " 8 waves, nt, new-hop reads");
        run(rows_k<4, true, true, 0>, 4, " 4 waves, nt, new-hop reads");
    }
    return 0;
}
