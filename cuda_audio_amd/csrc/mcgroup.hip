// mcgroup.hip - the multi-GPU driver behind include/mcconv_group.h: IR partitions sharded over the devices of one node,
// one host thread per device, RCCL reduce-scatter (or reduce) of the partial wet blocks on the engines' streams.
// Everything that touches samples is libmcconv.so's (mc_partial_batch_device / mc_finish_batch_*_device); this file
// owns the device staging buffers, the communicators and the per-call choreography.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcconv_group.h"

namespace {

thread_local char g_gerr[512] = "";

int gfail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_gerr, sizeof(g_gerr), fmt, ap);
    va_end(ap);
    return code;
}

struct Rank {
    int device = 0;
    mc_engine* eng = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t pb = 0, pe = 0;
    float *d_in = nullptr, *d_part = nullptr, *d_sum = nullptr, *d_out = nullptr;  // [2][cap], [2][cap], [2][cap], [2][cap] frames
    hipEvent_t ev_part = nullptr, ev_done = nullptr;
    int rc = 0;
    std::string err;
};

// virtual ranks on one device: sum[c][i] = sum_q part_q[c][first + i]   (what the reduce-scatter leaves on this rank)
struct SumArgs {
    const float* part[16];
    int n;
};
__global__ __launch_bounds__(256) void k_group_sum(SumArgs A, float* __restrict__ sum, int64_t frames_total, int64_t first, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * count) return;
    const int64_t c = i / count, j = i - c * count;
    float acc = 0.f;
    for (int q = 0; q < A.n; q++) acc += A.part[q][c * frames_total + first + j];
    sum[c * count + j] = acc;
}

}  // namespace

struct mc_group {
    std::vector<Rank> r;
    mc_config cfg;
    uint64_t cap = 0;  // frames per channel the staging buffers hold
    bool rccl = false, dup = false;
    bool solo_exchange = false;  // one rank, but through partial -> RCCL -> finish (mc_config.reserved = 1: the collective's code on a one-GPU box)
    int pm = 1;
};

namespace {

#define G_HIP(expr)                                                                                        \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) return gfail(MC_ERR_HIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

int ensure_buffers(mc_group* g, uint64_t frames) {
    if (g->cap >= frames) return MC_OK;
    for (Rank& k : g->r) {
        G_HIP(hipSetDevice(k.device));
        G_HIP(hipStreamSynchronize(k.stream));
        for (float** p : {&k.d_in, &k.d_part, &k.d_sum, &k.d_out}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
            G_HIP(hipMalloc(p, sizeof(float) * 2 * frames));
        }
    }
    g->cap = frames;
    return MC_OK;
}

// one rank's share of a batch; `phase` barriers are the caller's (threads meet between partial and exchange only where
// ranks share a device)
int rank_front(mc_group* g, int q, const float* in1, const float* in2, uint64_t T) {
    Rank& k = g->r[q];
    const uint64_t F = T * MC_BLOCK;
    G_HIP(hipSetDevice(k.device));
    G_HIP(hipMemcpyAsync(k.d_in, in1, sizeof(float) * F, hipMemcpyHostToDevice, k.stream));
    G_HIP(hipMemcpyAsync(k.d_in + F, in2, sizeof(float) * F, hipMemcpyHostToDevice, k.stream));
    if (g->r.size() == 1 && !g->solo_exchange) return MC_OK;
    int rc = mc_partial_batch_device(k.eng, k.d_in, k.d_in + F, k.d_part, T);
    if (rc) return gfail(rc, "rank %d: %s", q, mc_last_error());
    if (g->dup) G_HIP(hipEventRecord(k.ev_part, k.stream));
    return MC_OK;
}

int rank_back(mc_group* g, int q, float* outL, float* outR, uint64_t T, bool scatter) {
    Rank& k = g->r[q];
    const int n = (int)g->r.size();
    const uint64_t F = T * MC_BLOCK;
    G_HIP(hipSetDevice(k.device));
    if (n == 1 && !g->solo_exchange) {
        int rc = mc_process_batch_device(k.eng, k.d_in, k.d_in + F, k.d_out, k.d_out + F, T);
        if (rc) return gfail(rc, "rank 0: %s", mc_last_error());
        G_HIP(hipMemcpyAsync(outL, k.d_out, sizeof(float) * F, hipMemcpyDeviceToHost, k.stream));
        G_HIP(hipMemcpyAsync(outR, k.d_out + F, sizeof(float) * F, hipMemcpyDeviceToHost, k.stream));
        G_HIP(hipStreamSynchronize(k.stream));
        return MC_OK;
    }
    const uint64_t Ts = scatter ? T / (uint64_t)n : T, first = scatter ? (uint64_t)q * Ts : 0, Fs = Ts * MC_BLOCK;
    const bool finishes = scatter || q == 0;
    if (g->rccl) {
        for (int c = 0; c < 2; c++) {
            ncclResult_t nr = scatter ? ncclReduceScatter(k.d_part + (size_t)c * F, k.d_sum + (size_t)c * Fs, Fs, ncclFloat, ncclSum, k.comm, k.stream)
                                      : ncclReduce(k.d_part + (size_t)c * F, k.d_sum + (size_t)c * F, F, ncclFloat, ncclSum, 0, k.comm, k.stream);
            if (nr != ncclSuccess) return gfail(MC_ERR_HIP, "rank %d: RCCL %s -> %s", q, scatter ? "reduce-scatter" : "reduce", ncclGetErrorString(nr));
        }
    } else if (finishes) {  // virtual ranks on one device: wait for every rank's partial, then sum this rank's run
        SumArgs A;
        std::memset(&A, 0, sizeof(A));
        A.n = n;
        for (int p = 0; p < n; p++) {
            A.part[p] = g->r[p].d_part;
            if (p != q) G_HIP(hipStreamWaitEvent(k.stream, g->r[p].ev_part, 0));
        }
        hipLaunchKernelGGL(k_group_sum, dim3((unsigned)((2 * Fs + 255) / 256)), dim3(256), 0, k.stream, A, k.d_sum, (int64_t)F, (int64_t)(first * MC_BLOCK), (int64_t)Fs);
        G_HIP(hipGetLastError());
    }
    int rc;
    if (scatter)
        rc = mc_finish_batch_slice_device(k.eng, k.d_in, k.d_in + F, k.d_sum, k.d_out, k.d_out + Fs, T, first, Ts);
    else if (q == 0)
        rc = mc_finish_batch_device(k.eng, k.d_in, k.d_in + F, k.d_sum, k.d_out, k.d_out + F, T);
    else
        rc = mc_finish_batch_device(k.eng, nullptr, nullptr, nullptr, nullptr, nullptr, T);
    if (rc) return gfail(rc, "rank %d: %s", q, mc_last_error());
    if (finishes) {
        G_HIP(hipMemcpyAsync(outL + first * MC_BLOCK, k.d_out, sizeof(float) * Fs, hipMemcpyDeviceToHost, k.stream));
        G_HIP(hipMemcpyAsync(outR + first * MC_BLOCK, k.d_out + Fs, sizeof(float) * Fs, hipMemcpyDeviceToHost, k.stream));
    }
    G_HIP(hipEventRecord(k.ev_done, k.stream));
    G_HIP(hipStreamSynchronize(k.stream));
    return MC_OK;
}

}  // namespace

extern "C" {

const char* mc_group_last_error(void) { return g_gerr; }

int mc_group_create(const mc_config* cfg, const int32_t* devices, uint32_t ndev, mc_group** out) {
    if (!cfg || !devices || !out || ndev < 1 || ndev > 16) return gfail(MC_ERR_ARG, "bad argument (1..16 devices)");
    if (cfg->struct_size != sizeof(mc_config)) return gfail(MC_ERR_ARG, "mc_config size mismatch");
    if (cfg->form != 0 || cfg->pipeline != 0) return gfail(MC_ERR_ARG, "a group runs partitioned, unpipelined engines");
    mc_group* g = new (std::nothrow) mc_group();
    if (!g) return gfail(MC_ERR_NOMEM, "out of host memory");
    g->cfg = *cfg;
    g->pm = (int)((cfg->period ? cfg->period : MC_BLOCK) / MC_BLOCK);
    g->r.resize(ndev);
    for (uint32_t a = 0; a < ndev; a++)
        for (uint32_t b = a + 1; b < ndev; b++)
            if (devices[a] == devices[b]) g->dup = true;
    // partition runs in multiples of 16, as cuda_audio_amd/sharded.py shard_bounds
    const uint32_t P = cfg->max_partitions ? cfg->max_partitions : (uint32_t)((cfg->n_ref - 1024 + MC_BLOCK - 1) / MC_BLOCK);
    const uint32_t total = (P + 15) / 16 * 16, per = ((total / 16 + ndev - 1) / ndev) * 16;
    for (uint32_t q = 0; q < ndev; q++) {
        Rank& k = g->r[q];
        k.device = devices[q];
        k.pb = std::min(q * per, total);
        k.pe = std::min(k.pb + per, total);
        if (ndev > 1 && k.pe <= k.pb) {
            mc_group_destroy(g);
            return gfail(MC_ERR_ARG, "rank %u would own no partitions (%u partitions over %u devices)", q, P, ndev);
        }
        mc_config c = *cfg;
        c.device = k.device;
        c.part_begin = ndev > 1 ? k.pb : 0;
        c.part_end = ndev > 1 ? k.pe : 0;
        int rc = mc_create(&c, &k.eng);
        if (rc) {
            gfail(rc, "rank %u (device %d): %s", q, k.device, mc_last_error());
            mc_group_destroy(g);
            return rc;
        }
        k.stream = (hipStream_t)mc_get_stream(k.eng);
        if (hipSetDevice(k.device) != hipSuccess || hipEventCreateWithFlags(&k.ev_part, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&k.ev_done, hipEventDisableTiming) != hipSuccess) {
            mc_group_destroy(g);
            return gfail(MC_ERR_HIP, "rank %u: event creation failed", q);
        }
    }
    g->solo_exchange = ndev == 1 && cfg->reserved == 1;
    if ((ndev > 1 && !g->dup) || g->solo_exchange) {
        std::vector<ncclComm_t> comms(ndev);
        std::vector<int> devs(devices, devices + ndev);
        ncclResult_t nr = ncclCommInitAll(comms.data(), (int)ndev, devs.data());
        if (nr != ncclSuccess) {
            mc_group_destroy(g);
            return gfail(MC_ERR_HIP, "ncclCommInitAll over %u devices -> %s", ndev, ncclGetErrorString(nr));
        }
        for (uint32_t q = 0; q < ndev; q++) g->r[q].comm = comms[q];
        g->rccl = true;
    }
    *out = g;
    return MC_OK;
}

void mc_group_destroy(mc_group* g) {
    if (!g) return;
    for (Rank& k : g->r) {
        (void)hipSetDevice(k.device);
        if (k.stream) (void)hipStreamSynchronize(k.stream);
        if (k.comm) (void)ncclCommDestroy(k.comm);
        for (float* p : {k.d_in, k.d_part, k.d_sum, k.d_out})
            if (p) (void)hipFree(p);
        if (k.ev_part) (void)hipEventDestroy(k.ev_part);
        if (k.ev_done) (void)hipEventDestroy(k.ev_done);
        if (k.eng) mc_destroy(k.eng);
    }
    delete g;
}

uint32_t mc_group_size(const mc_group* g) { return g ? (uint32_t)g->r.size() : 0; }

mc_engine* mc_group_engine(mc_group* g, uint32_t rank) { return (g && rank < g->r.size()) ? g->r[rank].eng : nullptr; }

int mc_group_shard(const mc_group* g, uint32_t rank, uint32_t* begin, uint32_t* end) {
    if (!g || rank >= g->r.size() || !begin || !end) return gfail(MC_ERR_ARG, "bad argument");
    *begin = g->r[rank].pb;
    *end = g->r[rank].pe;
    return MC_OK;
}

const char* mc_group_exchange(const mc_group* g) {
    if (!g || (g->r.size() < 2 && !g->solo_exchange)) return "none";
    return g->rccl ? "rccl" : "device-sum";
}

int mc_group_load_ir(mc_group* g, uint64_t idx, const float* lr, uint64_t frames, uint64_t nframes) {
    if (!g) return gfail(MC_ERR_ARG, "null group");
    for (size_t q = 0; q < g->r.size(); q++) {
        int rc = mc_load_ir(g->r[q].eng, idx, lr, frames, nframes);
        if (rc) return gfail(rc, "rank %zu: %s", q, mc_last_error());
    }
    return MC_OK;
}

int mc_group_set_params(mc_group* g, int half, const mc_cc_value* v) {
    if (!g) return gfail(MC_ERR_ARG, "null group");
    for (size_t q = 0; q < g->r.size(); q++) {
        int rc = mc_set_params(g->r[q].eng, half, v);
        if (rc) return gfail(rc, "rank %zu: %s", q, mc_last_error());
    }
    return MC_OK;
}

int mc_group_process_batch(mc_group* g, const float* in1, const float* in2, float* outL, float* outR, uint64_t nblocks) {
    if (!g || !in1 || !in2 || !outL || !outR) return gfail(MC_ERR_ARG, "null argument");
    if (nblocks < 1 || nblocks > g->cfg.max_batch) return gfail(MC_ERR_ARG, "nblocks %llu outside [1, %u]", (unsigned long long)nblocks, g->cfg.max_batch);
    if (nblocks % (uint64_t)g->pm) return gfail(MC_ERR_ARG, "nblocks is not a multiple of the period");
    int rc = ensure_buffers(g, nblocks * MC_BLOCK);
    if (rc) return rc;
    const int n = (int)g->r.size();
    // equal runs of whole periods -> every rank finishes its run after a reduce-scatter; else rank 0 finishes all after a reduce
    const bool scatter = (n > 1 || g->solo_exchange) && nblocks % ((uint64_t)n * g->pm) == 0;
    auto run = [&](int phase) {
        std::vector<std::thread> th;
        for (int q = 0; q < n; q++)
            th.emplace_back([&, q]() {
                Rank& k = g->r[q];
                if (k.rc) return;  // (a rank that failed in the first phase sits the second one out)
                k.rc = phase == 0 ? rank_front(g, q, in1, in2, nblocks) : rank_back(g, q, outL, outR, nblocks, scatter);
                if (k.rc) k.err = g_gerr;
            });
        for (auto& t : th) t.join();
    };
    for (Rank& k : g->r) k.rc = 0;
    run(0);  // copies in and partial sums queued on every rank (the ranks of one device have recorded their partials' events)
    bool failed = false;
    for (Rank& k : g->r) failed = failed || k.rc;
    if (!failed) run(1);  // exchange, finish, copy out, wait
    for (Rank& k : g->r)
        if (k.rc) {
            // (after a failure between partial and finish the engines hold a half-done batch: the group is unusable until re-created)
            std::snprintf(g_gerr, sizeof(g_gerr), "%s", k.err.c_str());
            return k.rc;
        }
    return MC_OK;
}

}  // extern "C"
