// singlefft_host.hip.h — host side of the single-transform form (kernels and the algorithm: singlefft.hip.h).
// Included by mcconv.hip after its helpers (fail, HIP_TRY, pan_l ...); SfState itself is declared next to mc_engine.
inline void sf_free(mc_engine* e) {
    SfState* s = e->sf;
    if (!s) return;
    (void)hipFree(s->d_live);
    (void)hipFree(s->d_W);
    (void)hipFree(s->d_T);
    (void)hipFree(s->d_Z);
    (void)hipFree(s->d_acc);
    (void)hipFree(s->d_ctr);
    for (int i = 0; i < 4; i++) (void)hipFree(s->d_io[i]);
    for (int i = 0; i < kMaxIrs; i++)
        if (e->irs[i].d_S) (void)hipFree(e->irs[i].d_S);
    delete s;
    e->sf = nullptr;
}

inline int sf_zero(mc_engine* e) {
    SfState* s = e->sf;
    HIP_TRY(hipMemsetAsync(s->d_live, 0, sizeof(float2) * 4 * (size_t)(s->N / 2), e->stream));
    HIP_TRY(hipMemsetAsync(s->d_acc, 0, sizeof(float) * 2 * (size_t)s->N, e->stream));
    HIP_TRY(hipMemsetAsync(s->d_ctr, 0, sizeof(unsigned), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    s->base = 0;
    e->t_abs = 0;
    return MC_OK;
}

// buffers of the form; the caller has set device, stream, pm, the twiddle table and the mapped period buffer
inline int sf_create(mc_engine* e) {
    const uint64_t N = e->cfg.n_ref;
    if (N > (1ull << 20)) return fail(MC_ERR_ARG, "the single-transform form takes n_ref up to 1048576");
    if (e->cfg.part_begin || e->cfg.part_end || e->cfg.precision || e->cfg.pipeline)
        return fail(MC_ERR_ARG, "the single-transform form has no partitions to shard, no fp16 storage and no pipelined batches");
    if (!e->cfg.compat) return fail(MC_ERR_ARG, "the single-transform form is the reference's algorithm: compat = 1 only");
    SfState* s = new (std::nothrow) SfState();
    if (!s) return fail(MC_ERR_NOMEM, "out of host memory");
    e->sf = s;
    s->N = (int)N;
    s->M = (int)(N / FFT_N);
    s->AT = std::max(1, std::min(8, 2048 / s->M));
    s->lds_bytes = sizeof(float2) * (size_t)(2 * s->AT * s->M + s->M / 2);
    s->stockham = LAB_ENV("MCCONV_SF_STOCKHAM") != nullptr;  // (tests: the LDS transform of the long sizes at a short one)
    HIP_TRY(hipMalloc(&s->d_live, sizeof(float2) * 4 * (size_t)(N / 2)));
    HIP_TRY(hipMalloc(&s->d_W, sizeof(float2) * N));
    HIP_TRY(hipMalloc(&s->d_T, sizeof(float2) * N));
    HIP_TRY(hipMalloc(&s->d_Z, sizeof(float2) * N));
    HIP_TRY(hipMalloc(&s->d_acc, sizeof(float) * 2 * N));
    HIP_TRY(hipMalloc(&s->d_ctr, sizeof(unsigned)));
    s->io_cap = (size_t)e->Tmax * MC_B;
    for (int i = 0; i < 4; i++) HIP_TRY(hipMalloc(&s->d_io[i], sizeof(float) * s->io_cap));
    return sf_zero(e);
}

// Convolution::prepare, conv.cu:207-253
inline int sf_load_ir(mc_engine* e, uint64_t idx, const float* lr, uint64_t frames, uint64_t nframes) {
    SfState* s = e->sf;
    const uint64_t N = (uint64_t)s->N;
    const uint64_t n = std::min<uint64_t>(frames, N - nframes);  // conv.cu:239
    IrEntry& ir = e->irs[idx];
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<float2> z(N, make_float2(0.f, 0.f));  // conv.cu:223-227, 240: L -> re, R -> im, zero padded
    double sm[4] = {0, 0, 0, 0};
    for (uint64_t m = 0; m < n; m++) {
        z[m] = make_float2(lr[2 * m], lr[2 * m + 1]);
        const double sg = (m & 1) ? -1.0 : 1.0;
        sm[0] += lr[2 * m];
        sm[1] += lr[2 * m + 1];
        sm[2] += sg * lr[2 * m];
        sm[3] += sg * lr[2 * m + 1];
    }
    if (!ir.d_S) HIP_TRY(hipMalloc(&ir.d_S, sizeof(float2) * N));  // [H_L | H_R], N/2 bins each, bin d + M c at [d][c]
    HIP_TRY(hipMemcpy(s->d_W, z.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sf_ir_cols, dim3(FFT_N / s->AT), dim3(256), s->lds_bytes, e->stream, s->N, s->M, s->AT, s->d_W, s->d_T);
    hipLaunchKernelGGL(k_sf_ir_rows, dim3(s->M / SF_ROWS), dim3(64 * SF_ROWS), 0, e->stream, s->M, s->d_T, s->d_Z, e->d_tw);
    hipLaunchKernelGGL(k_sf_ir_unpack, dim3((s->N / 2 + 255) / 256), dim3(256), 0, e->stream, s->N, s->M, s->d_Z, ir.d_S);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::memcpy(ir.sums, sm, sizeof(sm));
    ir.taps = n;
    ir.P = (int)((n + MC_B - 1) / MC_B);
    if ((int)idx + 1 > e->nirs) e->nirs = (int)idx + 1;
    return MC_OK;
}

// One call of the reference's onProcess on buffers the device can read and write.  The parameters are sampled and the
// cross-fade counters stepped (compare-exchange, no lock), as onProcess does with its public members (conv.cu:339-353).
inline int sf_call(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR, int nframes, bool publish = false) {
    SfState* s = e->sf;
    mc_cc_value cc[2];
    e->last_gen = e->ph.sample(cc);  // (lock-free: params_handoff.h)
    for (int i = 0; i < 2; i++) e->ph.count_down(i, cc[i].vsteps, cc[i].vsteps > 0 ? 1 : 0);
    for (int i = 0; i < 2; i++)
        if (cc[i].select >= (uint64_t)kMaxIrs || !e->irs[cc[i].select].d_S)
            return fail(MC_ERR_STATE, "half %d selects IR %llu which is not loaded", i, (unsigned long long)cc[i].select);
    if (cc[0].predelay > MC_MAX_PREDELAY) return fail(MC_ERR_ARG, "predelay %llu > %d", (unsigned long long)cc[0].predelay, MC_MAX_PREDELAY);
    SfCall C;
    std::memset(&C, 0, sizeof(C));
    C.in1 = in1;
    C.in2 = in2;
    C.outL = outL;
    C.outR = outR;
    C.nframes = nframes;
    C.pd = (int)cc[0].predelay;
    C.base = s->base;
    if (publish) {
        C.done_flag = e->hd_flag;
        C.seq = ++e->flag_seq;
    }
    for (int i = 0; i < 2; i++) {
        C.wet[i] = cc[i].wet;
        C.div[i] = (float)(cc[i].vsteps + 5);
        C.b[i] = e->irs[cc[i].select].d_S;
        const double pw[2] = {pan_l(cc[i].panWet), pan_r(cc[i].panWet)}, pdry[2] = {pan_l(cc[i].panDry), pan_r(cc[i].panDry)};
        for (int c = 0; c < 2; c++) {
            C.sc[c][i] = (float)(pw[c] * (double)cc[i].level / (double)s->N);
            C.dry[c][i] = (float)((double)cc[i].dry * pdry[c] * (double)cc[i].level);
        }
    }
    hipLaunchKernelGGL(k_sf_fwdmac, dim3(2 * s->M / SF_ROWS), dim3(64 * SF_ROWS), 0, e->stream, C, s->N, s->M, s->d_live, s->d_W, e->d_tw);
    hipLaunchKernelGGL(k_sf_inv1, dim3(s->M / SF_ROWS), dim3(64 * SF_ROWS), 0, e->stream, s->N, s->M, s->d_W, s->d_T, e->d_tw);
    if (s->M <= FFT_N && !s->stockham)
        hipLaunchKernelGGL(k_sf_inv2w, dim3(FFT_N / SF_ROWS2), dim3(64 * SF_ROWS2), 0, e->stream, C, s->N, s->M, s->d_T, s->d_acc, s->d_ctr, e->d_tw);
    else
        hipLaunchKernelGGL(k_sf_inv2, dim3(FFT_N / s->AT), dim3(256), s->lds_bytes, e->stream, C, s->N, s->M, s->AT, s->d_T, s->d_acc, s->d_ctr);
    s->base = (s->base + (unsigned)nframes) & (unsigned)(s->N - 1);
    e->t_abs += (uint64_t)e->pm;
    return MC_OK;
}

// T blocks of 256 frames = T / pm calls, device buffers
inline int sf_batch_device(mc_engine* e, const float* d_in1, const float* d_in2, float* d_outL, float* d_outR, int T) {
    if (T <= 0 || T > e->Tmax) return fail(MC_ERR_ARG, "nblocks %d outside [1, %d]", T, e->Tmax);
    if (T % e->pm) return fail(MC_ERR_ARG, "nblocks %d is not a multiple of the period (%d blocks)", T, e->pm);
    const int nf = e->pm * MC_B;
    for (int q = 0; q < T / e->pm; q++) {
        int rc = sf_call(e, d_in1 + (size_t)q * nf, d_in2 + (size_t)q * nf, d_outL + (size_t)q * nf, d_outR + (size_t)q * nf, nf);
        if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    return MC_OK;
}

// host buffers, any length: chunks of max_batch blocks through the staging buffers
inline int sf_batch_host(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR, int T) {
    SfState* s = e->sf;
    if (T <= 0) return fail(MC_ERR_ARG, "nblocks must be positive");
    if (T % e->pm) return fail(MC_ERR_ARG, "nblocks %d is not a multiple of the period (%d blocks)", T, e->pm);
    for (int o = 0; o < T;) {
        const int n = std::min(T - o, e->Tmax);
        const size_t off = (size_t)o * MC_B, bytes = sizeof(float) * (size_t)n * MC_B;
        HIP_TRY(hipMemcpyAsync(s->d_io[0], in1 + off, bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(s->d_io[1], in2 + off, bytes, hipMemcpyHostToDevice, e->stream));
        int rc = sf_batch_device(e, s->d_io[0], s->d_io[1], s->d_io[2], s->d_io[3], n);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(outL + off, s->d_io[2], bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(outR + off, s->d_io[3], bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        o += n;
    }
    return MC_OK;
}

// one JACK period, host buffers: through the mapped period buffer (no copies are enqueued), then wait (the reference
// blocks in cudaEventSynchronize, conv.cu:455)
inline int sf_process(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR) {
    const int nf = e->pm * MC_B;
    const size_t cap = (size_t)e->Thost * MC_B;
    const float *pin1 = e->hd_io + 0 * cap, *pin2 = e->hd_io + 1 * cap;
    if (e->bar_io) {  // straight into device memory through the BAR (write-combined: fenced before the launches)
        std::memcpy(e->d_bar + 16, in1, sizeof(float) * nf);
        std::memcpy(e->d_bar + 16 + 4 * MC_B, in2, sizeof(float) * nf);
        _mm_sfence();
        pin1 = e->d_bar + 16;
        pin2 = e->d_bar + 16 + 4 * MC_B;
    } else {
        std::memcpy(e->h_io + 0 * cap, in1, sizeof(float) * nf);
        std::memcpy(e->h_io + 1 * cap, in2, sizeof(float) * nf);
    }
    int rc = sf_call(e, pin1, pin2, e->hd_io + 2 * cap, e->hd_io + 3 * cap, nf, e->spin_wait);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    if (e->spin_wait) {  // the last workgroup publishes the sequence number once the period is in h_io
        const unsigned seq = e->flag_seq;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(e->h_flag, __ATOMIC_ACQUIRE) != seq)
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(300)) {
                HIP_TRY(hipStreamSynchronize(e->stream));
                if (__atomic_load_n(e->h_flag, __ATOMIC_ACQUIRE) != seq) return fail(MC_ERR_HIP, "period did not complete");
            }
    } else {
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    std::memcpy(outL, e->h_io + 2 * cap, sizeof(float) * nf);
    std::memcpy(outR, e->h_io + 3 * cap, sizeof(float) * nf);
    return MC_OK;
}
