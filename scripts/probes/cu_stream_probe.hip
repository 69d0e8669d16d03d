// cu_stream_probe.hip - how fast ONE CU pulls HBM-resident data, by how many CUs pull at once (gfx950).
// Each workgroup (512 threads, 70 KB of LDS requested so that a CU holds at most two) reads ITEMS windows of
// 128 KB (16 rows of 8 KB, 16 bytes per lane - k_g2_mac's window fill), every window a different 128 KB of a 4 GB
// buffer (beyond the 256 MB Infinity Cache), all 16 loads of a thread in flight before the first use.
//   hipcc -O3 --offload-arch=gfx950 -o cu_stream_probe cu_stream_probe.hip && ./cu_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITEMS 64
__global__ __launch_bounds__(512) void k_read(const float4* __restrict__ src, size_t nwin, float* out, int stride_w) {
    extern __shared__ float lds[];
    float acc = 0.f;
    for (int it = 0; it < ITEMS; it++) {
        const size_t w = ((size_t)blockIdx.x * ITEMS + it) * (size_t)stride_w % nwin;
        const float4* p = src + w * 8192;
        float4 x[16];
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = p[threadIdx.x + 512 * r];
#pragma unroll
        for (int r = 0; r < 16; r++) acc += x[r].x + x[r].y + x[r].z + x[r].w;
    }
    if (acc == 1234.5f) out[0] = acc + lds[0];
}
__global__ __launch_bounds__(512) void k_write(float4* __restrict__ dst, size_t nwin, int stride_w) {
    for (int it = 0; it < ITEMS; it++) {
        const size_t w = ((size_t)blockIdx.x * ITEMS + it) * (size_t)stride_w % nwin;
        float4* p = dst + w * 8192;
#pragma unroll
        for (int r = 0; r < 16; r++) p[threadIdx.x + 512 * r] = make_float4((float)it, (float)r, 1.f, 2.f);
    }
}

int main() {
    const size_t bytes = (size_t)4 << 30, nwin = bytes / (8192 * 16);
    float4* buf;
    float* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int lds : {70 * 1024, 100 * 1024}) {
        printf("LDS request %d KB (=> at most %d workgroup(s) per CU)\n", lds / 1024, lds > 80 * 1024 ? 1 : 2);
        for (int mode = 0; mode < 2; mode++)
            for (int grid : {8, 32, 64, 128, 256, 512, 1024}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    hipEventRecord(e0, 0);
                    if (mode == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(512), lds, 0, buf, nwin, out, 7);
                    else hipLaunchKernelGGL(k_write, dim3(grid), dim3(512), lds, 0, buf, nwin, 7);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                const double total = (double)grid * ITEMS * 131072.0;
                const int cus = grid < 256 ? grid : 256;
                printf("  %s grid %4d: %7.3f ms  %7.2f TB/s chip  %6.1f GB/s per busy CU\n", mode ? "write" : "read ", grid, best, total / best / 1e9,
                       total / best / 1e6 / cus);
            }
    }
    return 0;
}
