// stub_engine.cpp - test infrastructure: a stand-in for libmcconv.so's C ABI (include/mcconv.h) and for selectGpu(),
// so that the UNMODIFIED host (cuda_audio_amd/host/main.cpp, conv.cpp, wav.cpp, settings.cpp, jackclient.cpp) can be
// linked and run without a GPU.  Every call that reaches the engine is logged to $MCSTUB_LOG; mc_process* write
// silence.  tests/test_reference_pins.py uses it to walk the reference's own settings.txt and ir/all.index through
// main()'s loader loop (reference src/main.cu:39-80) and to check the calls the engine would receive.
// It computes nothing: the product path has no CPU route (the real library fails loudly without its HIP code).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/mcconv.h"

struct mc_engine {
    mc_config cfg;
    mc_cc_value cc[2];
    uint64_t nirs = 0, processed = 0;
    uint32_t period = 256;
};

namespace {
FILE* logf() {
    static FILE* f = nullptr;
    if (!f) {
        const char* p = std::getenv("MCSTUB_LOG");
        f = p ? fopen(p, "w") : stderr;
    }
    return f;
}
}  // namespace

int selectGpu() {
    fprintf(logf(), "selectGpu\n");
    return 0;
}

extern "C" {
uint32_t mc_abi_version(void) { return MC_ABI_VERSION; }
const char* mc_last_error(void) { return "stub engine"; }
void mc_default_config(mc_config* cfg) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(*cfg);
    cfg->device = -1;
    cfg->n_ref = 131072;
    cfg->max_batch = 2048;
    cfg->compat = 1;
}
void mc_default_params(mc_cc_value* v) {
    std::memset(v, 0, sizeof(*v));
    v->speed = 100;
    v->dry = v->wet = 0.5f;
    v->level = 1.0f;
}
int mc_create(const mc_config* cfg, mc_engine** out) {
    auto* e = new mc_engine();
    e->cfg = *cfg;
    mc_default_params(&e->cc[0]);
    mc_default_params(&e->cc[1]);
    fprintf(logf(), "create n_ref=%llu max_batch=%u compat=%u\n", (unsigned long long)cfg->n_ref, cfg->max_batch, cfg->compat);
    *out = e;
    return MC_OK;
}
void mc_destroy(mc_engine* e) {
    fprintf(logf(), "destroy processed=%llu\n", (unsigned long long)(e ? e->processed : 0));
    fflush(logf());
    delete e;
}
int mc_set_period(mc_engine* e, uint32_t nframes) {
    e->period = nframes;
    fprintf(logf(), "set_period %u\n", nframes);
    return MC_OK;
}
int mc_load_ir(mc_engine* e, uint64_t idx, const float* lr, uint64_t frames, uint64_t nframes) {
    // what the engine would keep: min(frames, n_ref - nframes) frames (conv.cu:239); a checksum of the frames handed over
    double sum = 0.0, asum = 0.0;
    for (uint64_t i = 0; i < 2 * frames; i++) {
        sum += lr[i];
        asum += lr[i] < 0 ? -lr[i] : lr[i];
    }
    const uint64_t keep = frames < e->cfg.n_ref - nframes ? frames : e->cfg.n_ref - nframes;
    fprintf(logf(), "load_ir idx=%llu frames=%llu nframes=%llu keep=%llu sum=%.9e asum=%.9e first=%.9e,%.9e\n", (unsigned long long)idx,
            (unsigned long long)frames, (unsigned long long)nframes, (unsigned long long)keep, sum, asum, frames ? lr[0] : 0.f,
            frames ? lr[1] : 0.f);
    if (idx + 1 > e->nirs) e->nirs = idx + 1;
    return MC_OK;
}
int mc_set_params(mc_engine* e, int half, const mc_cc_value* v) {
    if (std::memcmp(&e->cc[half], v, sizeof(*v)) != 0)
        fprintf(logf(), "set_params half=%d select=%llu predelay=%llu speed=%llu vsteps=%llu dry=%g wet=%g panDry=%g panWet=%g level=%g\n", half,
                (unsigned long long)v->select, (unsigned long long)v->predelay, (unsigned long long)v->speed, (unsigned long long)v->vsteps, v->dry,
                v->wet, v->panDry, v->panWet, v->level);
    e->cc[half] = *v;
    return MC_OK;
}
int mc_get_params(const mc_engine* e, int half, mc_cc_value* v) {
    *v = e->cc[half];
    return MC_OK;
}
int mc_process(mc_engine* e, const float*, const float*, float* outL, float* outR, uint64_t nframes) {
    std::memset(outL, 0, sizeof(float) * nframes);
    std::memset(outR, 0, sizeof(float) * nframes);
    e->processed += nframes;
    return MC_OK;
}
int mc_process_batch(mc_engine* e, const float*, const float*, float* outL, float* outR, uint64_t nblocks) {
    std::memset(outL, 0, sizeof(float) * nblocks * MC_BLOCK);
    std::memset(outR, 0, sizeof(float) * nblocks * MC_BLOCK);
    e->processed += nblocks * MC_BLOCK;
    return MC_OK;
}
double mc_avg_runtime_ms(const mc_engine*) { return 0.0; }
}
