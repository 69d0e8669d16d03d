#include "gpu.h"

#include <hip/hip_runtime_api.h>

#include <cassert>
#include <cstdint>
#include <cstdlib>

#include "log.h"

int selectGpu() {
    int count = 0;
    hipError_t rc = hipGetDeviceCount(&count);
    assert(rc == hipSuccess && count > 0);
    (void)rc;
    int best = -1;
    uint64_t bestScore = 0;
    if (const char* env = std::getenv("MCCONV_DEVICE")) {
        const int want = std::atoi(env);
        if (want >= 0 && want < count) best = want;
    }
    hipDeviceProp_t prop;
    if (best < 0) {
        for (int id = 0; id < count; id++) {
            if (hipGetDeviceProperties(&prop, id) != hipSuccess) continue;
            Log::info("gpu", "GPU %d", id);
            Log::newline(ESC(1) "%s" ESC(0) " (%s)", prop.name, prop.gcnArchName);
            Log::newline("Compute units:      " ESC(1) "%d" ESC(0), prop.multiProcessorCount);
            Log::newline("Clock rate (kHz):   " ESC(1) "%d" ESC(0), prop.clockRate);
            Log::newline("Memory (GiB):       " ESC(1) "%.0f" ESC(0), prop.totalGlobalMem / 1073741824.0);
            const uint64_t score = (uint64_t)prop.multiProcessorCount * (uint64_t)prop.clockRate;
            if (score > bestScore) {
                bestScore = score;
                best = id;
            }
        }
    }
    assert(best >= 0);
    rc = hipSetDevice(best);
    assert(rc == hipSuccess);
    if (hipGetDeviceProperties(&prop, best) == hipSuccess)
        Log::info("gpu", ESC(32;1) "Selected GPU %d: \"%s\" (%s)", best, prop.name, prop.gcnArchName);
    return best;
}
