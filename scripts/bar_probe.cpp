#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <immintrin.h>
__global__ void k_wait(volatile unsigned* bell, unsigned want, unsigned* out, unsigned long long* ticks) {
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned v;
    do {
        v = __hip_atomic_load((unsigned*)bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } while (v != want && __builtin_amdgcn_s_memrealtime() - t0 < 100000000ull);
    *ticks = __builtin_amdgcn_s_memrealtime() - t0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
    unsigned* d_bell = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&d_bell, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags finegrained: %s ptr %p\n", hipGetErrorString(e), (void*)d_bell);
    hipPointerAttribute_t at;
    if (e == hipSuccess && hipPointerGetAttributes(&at, d_bell) == hipSuccess) printf("type %d host %p dev %p managed %d\n", (int)at.type, at.hostPointer, at.devicePointer, at.isManaged);
    unsigned *h_out, *hd_out; unsigned long long* d_ticks;
    hipHostMalloc((void**)&h_out, 64, hipHostMallocMapped); hipHostGetDevicePointer((void**)&hd_out, h_out, 0);
    hipMalloc(&d_ticks, 8);
    if (e != hipSuccess) return 1;
    hipMemset(d_bell, 0, 4096); hipDeviceSynchronize();
    // can the CPU store to it directly?
    for (int trial = 1; trial <= 5; trial++) {
        *h_out = 0;
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, 0, d_bell, (unsigned)trial, hd_out, d_ticks);
        // give the kernel time to start
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(200)) {}
        auto t1 = std::chrono::steady_clock::now();
        __atomic_store_n(d_bell, (unsigned)trial, __ATOMIC_RELEASE);  // CPU store straight into device memory
        _mm_sfence();                                                 // (write-combining mapping: push the store out)
        while (__atomic_load_n(h_out, __ATOMIC_ACQUIRE) != (unsigned)trial) {
            if (std::chrono::steady_clock::now() - t1 > std::chrono::seconds(2)) { printf("timeout\n"); break; }
        }
        auto t2 = std::chrono::steady_clock::now();
        printf("trial %d: CPU store -> GPU saw it -> host saw the answer: %.2f us\n", trial, std::chrono::duration<double, std::micro>(t2 - t1).count());
        hipDeviceSynchronize();
    }
    // same with the bell in mapped host memory
    unsigned *h_bell, *hd_bell;
    hipHostMalloc((void**)&h_bell, 64, hipHostMallocMapped); hipHostGetDevicePointer((void**)&hd_bell, h_bell, 0);
    *h_bell = 0;
    for (int trial = 1; trial <= 5; trial++) {
        *h_out = 0;
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, 0, hd_bell, (unsigned)trial, hd_out, d_ticks);
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(200)) {}
        auto t1 = std::chrono::steady_clock::now();
        __atomic_store_n(h_bell, (unsigned)trial, __ATOMIC_RELEASE);
        while (__atomic_load_n(h_out, __ATOMIC_ACQUIRE) != (unsigned)trial) {
            if (std::chrono::steady_clock::now() - t1 > std::chrono::seconds(2)) { printf("timeout\n"); break; }
        }
        auto t2 = std::chrono::steady_clock::now();
        printf("host-memory bell trial %d: %.2f us\n", trial, std::chrono::duration<double, std::micro>(t2 - t1).count());
        hipDeviceSynchronize();
    }
    return 0;
}
