// How fast does ONE wavefront per SIMD issue independent packed multiply-adds?  (round 4, the JACK tail's direct convolution)
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/pkfma_probe scripts/probes/pkfma_probe.hip ; prints shader clocks per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(unsigned long long* out, float seed) {
    v2f acc[8], h[4], x[4];
    for (int i = 0; i < 8; i++) acc[i] = v2f{seed * i, seed};
    for (int i = 0; i < 4; i++) h[i] = v2f{seed + i, seed - i}, x[i] = v2f{seed * 2 + i, seed * 3 - i};
    float s[16];
    for (int i = 0; i < 16; i++) s[i] = seed * i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 32; r++) {
#pragma unroll
        for (int o = 0; o < 8; o++) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[o]) : "v"(h[r & 3]), "v"(x[o & 3]));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[o]) : "v"(h[r & 3]), "v"(x[o & 3]));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[o]) : "v"(h[r & 3]), "v"(x[o & 3]));
            if (MODE == 3) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[2 * o]) : "v"(h[r & 3].x), "v"(x[o & 3].x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[2 * o + 1]) : "v"(h[r & 3].y), "v"(x[o & 3].x));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float z = 0.f;
    for (int i = 0; i < 8; i++) z += acc[i].x + acc[i].y;
    for (int i = 0; i < 16; i++) z += s[i];
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (z == 12345.678f) out[1] = 1;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    const char* names[4] = {"v_pk_fma_f32 plain", "v_pk_fma_f32 low half broadcast", "v_pk_fma_f32 high half broadcast", "2 x v_fma_f32"};
    for (int threads : {256, 512, 1024})
        for (int mode = 0; mode < 4; mode++) {
            unsigned long long best = ~0ull;
            for (int rep = 0; rep < 5; rep++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, d, 1.0f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, d, 1.0f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, d, 1.0f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, d, 1.0f);
                unsigned long long h[2];
                hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                if (h[0] < best) best = h[0];
            }
            printf("%4d threads (%d waves per SIMD), %-34s: %5llu clocks for 256 instruction slots = %.2f per slot\n", threads, threads / 256, names[mode], best, best / 256.0);
        }
    return 0;
}
