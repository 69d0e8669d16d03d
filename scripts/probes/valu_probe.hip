// valu_probe.hip - issue rate of packed and scalar fp32 VALU instructions on gfx950, by waves per SIMD.
// Each wave runs REPS x 64 independent instructions of one kind on 16 register pairs; cycles per instruction and
// per SIMD = (s_memtime ticks of the slowest wave) x waves per SIMD ... reported as cycles per wave-instruction
// when W waves share the SIMD (throughput), and the time of one wave alone (issue cost).
//   hipcc -O3 --offload-arch=gfx950 -o valu_probe valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REPS 256

template <int KIND>
__global__ void k(unsigned long long* out, float seed) {
    v2f a[16], b = v2f{seed, seed * 0.5f}, c = v2f{0.25f, -0.125f};
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = v2f{seed + i, seed - i};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 5) {  // two scalar fmas in place of one packed (same flops)
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.y), "v"(c.y));
                }
                if (KIND == 6)  // the kernels' complex product: pk_mul then dependent pk_fma with op_sel swizzles
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                                 : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 7) {  // the same product from four scalar instructions
                    float re, im;
                    asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %2, %5\n\tv_fma_f32 %0, -%3, %5, %0\n\tv_fma_f32 %1, %3, %4, %1"
                                 : "=&v"(re), "=&v"(im) : "v"(a[i].x), "v"(a[i].y), "v"(b.x), "v"(b.y));
                    a[i] = v2f{re, im};
                }
                if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(c.x));
                if (KIND == 9) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                if (KIND == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (s == 12345.678f) out[0] = 0;
}

template <int KIND>
void run(const char* name, int insts_per_iter) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 4096);
    printf("%-34s", name);
    for (int wps : {1, 2, 4}) {  // waves per SIMD: block = 256 * wps threads, one block per CU
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double ticks = (double)h[h.size() / 2];
        const double n = (double)REPS * 64 * insts_per_iter;
        // cycles of SIMD time per wave-instruction when wps waves share the SIMD
        printf("  wps %d: %6.2f cyc/inst/wave -> %5.2f cyc/inst/SIMD", wps, ticks / n, ticks / n / wps);
    }
    printf("\n");
    hipFree(d);
}

int main() {
    run<0>("v_pk_fma_f32", 1);
    run<1>("v_pk_mul_f32", 1);
    run<2>("v_pk_add_f32", 1);
    run<9>("v_pk_add_f32 op_sel+neg (+-j rot)", 1);
    run<3>("v_fma_f32", 1);
    run<4>("v_add_f32", 1);
    run<5>("2 x v_fma_f32 (= 1 pk)", 2);
    run<6>("cmul: pk_mul + pk_fma (per pair)", 2);
    run<7>("cmul: 2 v_mul + 2 v_fma (per 4)", 4);
    run<8>("v_mov_b32", 1);
    run<10>("v_add_u32", 1);
    return 0;
}
