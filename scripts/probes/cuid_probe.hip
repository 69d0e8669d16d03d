// Which (XCC, SE, SH, CU) does a workgroup run on, and how many workgroups of a 512-thread, 69-KB-LDS launch share one?  (k_g2_mac's shape)
// hipcc --offload-arch=gfx950 -O2 -o cuid_probe cuid_probe.hip && ./cuid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out, unsigned long long* t) {
    __shared__ float pad[69 * 256];
    pad[threadIdx.x] = 0.f;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000) {}
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        t[blockIdx.x] = t0;
    }
    if (pad[threadIdx.x] != 0.f) out[0] = 0;
}
int main() {
    const int n = 1024;
    unsigned* d;
    unsigned long long* dt;
    hipMalloc(&d, 8 * n);
    hipMalloc(&dt, 8 * n);
    hipLaunchKernelGGL(k, dim3(n), dim3(512), 0, 0, d, dt);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    std::map<unsigned, std::vector<unsigned>> slots;
    for (int i = 0; i < n; i++) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned key = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        cnt[key]++;
        slots[key].push_back(hw & 0xf);
    }
    printf("%d workgroups on %zu distinct (xcc, se, sh, cu) keys\n", n, cnt.size());
    int shown = 0;
    for (auto& kv : cnt) {
        if (shown++ < 6) {
            printf("  key 0x%03x: %d workgroups, wave slots of their first waves:", kv.first, kv.second);
            for (unsigned s : slots[kv.first]) printf(" %u", s);
            printf("\n");
        }
    }
    std::map<int, int> hist;
    for (auto& kv : cnt) hist[kv.second]++;
    for (auto& kv : hist) printf("  %d keys with %d workgroups\n", kv.second, kv.first);
    return 0;
}
