// lab_kernels.hip.h - measured alternatives of k_g2_mac that lost (round 1's one-workgroup-per-CU form, round 3's lockstep form).
// Part of the translation unit only under -DMCCONV_LAB (scripts/build_variant.sh lab -DMCCONV_LAB): the default library
// carries neither the kernels nor the switches that select them (MCCONV_G2_WIDE, MCCONV_G2_DUO*).
#pragma once

// k_g2_mac_wide: the one-workgroup-per-CU form (MCCONV_G2_WIDE=1; kept as the measured alternative of k_g2_mac below).
// grid = 256 bins x chunks, block = 1024 (two halves: input 1 / input 2, later Y_L / Y_R).
// chunk_t + taps - 1 <= G2_N.
// Bounds of every global access (host checks: nitems = 256 * ceil(T / chunk_t), T <= ycap, ring a power of two,
// grid <= nitems; any grid >= 1 is correct, the loop strides over the items):
//   items    the loop runs item = blockIdx.x (< gridDim.x <= nitems), + gridDim.x while < nitems; the look-ahead for
//            item + gridDim.x is issued only under the same `< nitems` test, so window_row never sees an item
//            >= nitems.  item -> xq = item >> 3 < 32 nch, bin = (xq / nch) * 8 + (item & 7) < 256, chunk = xq % nch.
//   window   fdl[bin * ring + ((sb + n) & (ring - 1))]: the mask keeps the slot in [0, ring) for any sb (negative
//            at the start of the stream, wrapping later), bin < 256: inside fdl's 256 * ring entries.  Rows with
//            n >= L are not loaded.
//   spectra  float4 index j = tid + 1024 r < 4096 into a row of G2_N float2 = 4096 float4; rows (c * 257 + row),
//            c < 2, row <= 256: inside the IR's 2 * 257 * G2_N entries.  j passes through an empty asm only to stop
//            the compiler hoisting the address arithmetic above the transforms; its value is unchanged.
//   sums     Yc[bin * ycap + t_c0 + t], t < nout = min(chunk_t, T - t_c0): t_c0 + t < T <= ycap.
// LDS: G2_P(n) <= G2_P(8191) = 8446 < G2_LDS; the mirrored positions of bin 0 are permutations of [0, G2_N).
__global__ __launch_bounds__(G2_THREADS) void k_g2_mac_wide(const float4* __restrict__ fdl, int ring, int slot0, int T, int chunk_t,
                                                            int taps, Fft2Voices vv, float4* __restrict__ Yc, int ycap, int nitems) {
    __shared__ float2 s[2][G2_LDS];
    __shared__ float2 t_lo[128], t_hi[64];
    // block ids 8 apart run on one XCD: there the chunks of a bin follow each other, so that the bin's second-level
    // spectra (262 KB for four paths) and the overlap of adjacent windows are read once into that XCD's L2
    const int nch = nitems >> 8;
    const int half = threadIdx.x >> 9, tt = threadIdx.x & 511;
    g2_tables(t_lo, t_hi);
    // the window of an item, once: 16 bytes per slot carry both inputs.  G2_PW of its G2_N / G2_THREADS rows are requested
    // one item ahead (before the inverse transform of the current item) and wait in registers.
    float4 xw[G2_PW];
    auto window_row = [&](int it, int r) -> float4 {
        const int xq_ = it >> 3, bin_ = (xq_ / nch) * 8 + (it & 7), chunk_ = xq_ % nch;
        const int t0_ = chunk_ * chunk_t, L = min(chunk_t, T - t0_) + taps - 1;
        const float4* fk = fdl + (size_t)bin_ * ring;
        const int sb = slot0 + t0_ - (taps - 1);
        const int n = threadIdx.x + G2_THREADS * r;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < L) x = fk[(sb + n) & (ring - 1)];
        return x;
    };
#pragma unroll
    for (int r = 0; r < G2_PW; r++) xw[r] = window_row(blockIdx.x, r);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int xq = item >> 3;
    const int bin = (xq / nch) * 8 + (item & 7), chunk = xq % nch;
    const int t_c0 = chunk * chunk_t, nout = min(chunk_t, T - t_c0);
    {
        // element n = tid + 1024 r sits at G2_P(tid) + r (1024 + 32): one address, immediate offsets
        int t0 = threadIdx.x;
        asm volatile("" : "+v"(t0));
        float2* w0 = &s[0][G2_P(t0)];
#pragma unroll
        for (int r = 0; r < G2_N / G2_THREADS; r++) {
            const float4 x = r < G2_PW ? xw[r] : window_row(item, r);
            w0[r * (G2_THREADS + G2_THREADS / 32)] = make_float2(x.x, x.y);
            w0[G2_LDS + r * (G2_THREADS + G2_THREADS / 32)] = make_float2(x.z, x.w);
        }
    }
    __syncthreads();
    // The spectra of the first voice, G2_PF groups of product entries (4 G2_PF loads of 16 bytes per thread) at a time.
    // 32-bit byte offsets from the (wave-uniform) row bases: scalar base + vector offset addressing, no 64-bit
    // vector arithmetic.
    auto load_spectra = [&](int rp, float4 (&HLp)[G2_PF][2], float4 (&HRp)[G2_PF][2]) {
#pragma unroll
        for (int r2 = 0; r2 < G2_PF; r2++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                unsigned off = (threadIdx.x + G2_THREADS * (unsigned)(rp + r2)) * 16u;
                asm volatile("" : "+v"(off));  // addresses of this group are formed here, not ahead of the transforms
                const char* h = reinterpret_cast<const char*>((i == 0 ? vv.h0[0] : vv.h1[0]) + (size_t)bin * G2_N);
                HLp[r2][i] = *reinterpret_cast<const float4*>(h + off);
                HRp[r2][i] = *reinterpret_cast<const float4*>(h + (size_t)257 * G2_N * sizeof(float2) + off);
            }
    };
    // products in place: thread owns entries 2j, 2j + 1 (j = tid + 1024 r) of both buffers - the pairs of the
    // transforms' radix-2 stage, which is applied here on the way in and on the way out
    v2f yl[8], yr[8];
    auto products = [&](int rp, const float4 (&HLp)[G2_PF][2], const float4 (&HRp)[G2_PF][2]) {
#pragma unroll
        for (int r2 = 0; r2 < G2_PF; r2++) {
            const int r = rp + r2;
            int t0 = threadIdx.x;
            asm volatile("" : "+v"(t0));
            const int j = t0 + G2_THREADS * r;
            const unsigned off = (unsigned)j * 16u;
            const int idx = 2 * j;
            v2f aL0 = v2f{0.f, 0.f}, aL1 = aL0, aR0 = aL0, aR1 = aL0;
            for (int q = 0; q < (bin == 0 ? 4 : 2); q++) {
                const int i = q & 1, var = q >> 1;  // var 1 (bin 0 only): the spectrum of conj(x_i) against h2
                v2f S0, S1;
                if (var) {
                    // -f and -(f + N/2) are a pair again (usually in the other order)
                    const int m0 = g2_mirror(idx), m1 = g2_mirror(idx + 1);
                    const v2f a = vx_ld(&s[i][G2_P(m0 & ~1)]), b = vx_ld(&s[i][G2_P(m0 | 1)]);
                    const v2f sum = a + b, dif = a - b;
                    S0 = (m0 & 1) ? dif : sum;
                    S1 = (m1 & 1) ? dif : sum;
                    S0.y = -S0.y;
                    S1.y = -S1.y;
                } else {
                    // entries 2 j, 2 j + 1 (j = tid + 1024 r) share a pad group: G2_P(2 tid) + r (2048 + 64), and + 1
                    const float2* pp = &s[i][G2_P(2 * t0)] + r * (2 * G2_THREADS + 2 * G2_THREADS / 32);
                    const v2f a = vx_ld(pp), b = vx_ld(pp + 1);
                    S0 = a + b;
                    S1 = a - b;
                }
                const size_t row = (size_t)(var ? 256 : bin) * G2_N;
#pragma unroll
                for (int vi = 0; vi < MC_MAXV; vi++) {
                    if (vi >= vv.n) break;
                    const float2* h = i == 0 ? vv.h0[vi] : vv.h1[vi];
                    const float gl = i == 0 ? vv.g[vi].x : vv.g[vi].y, gr = i == 0 ? vv.g[vi].z : vv.g[vi].w;
                    float4 HL, HR;
                    if (vi == 0 && !var) {
                        HL = i == 0 ? HLp[r2][0] : HLp[r2][1];
                        HR = i == 0 ? HRp[r2][0] : HRp[r2][1];
                    } else {
                        const char* hb = reinterpret_cast<const char*>(h + row);
                        HL = *reinterpret_cast<const float4*>(hb + off);
                        HR = *reinterpret_cast<const float4*>(hb + (size_t)257 * G2_N * sizeof(float2) + off);
                    }
                    aL0 += gl * vx_mul(S0, v2f{HL.x, HL.y});
                    aL1 += gl * vx_mul(S1, v2f{HL.z, HL.w});
                    aR0 += gr * vx_mul(S0, v2f{HR.x, HR.y});
                    aR1 += gr * vx_mul(S1, v2f{HR.z, HR.w});
                }
            }
            yl[2 * r] = aL0 + aL1;
            yl[2 * r + 1] = aL0 - aL1;
            yr[2 * r] = aR0 + aR1;
            yr[2 * r + 1] = aR0 - aR1;
        }
    };
#if G2_AHEAD
    // the first G2_PF groups are requested before the forward transforms and arrive under them; the rest at the start
    // of the products, so that all of an item's spectra are in flight or in registers before the first product
    static_assert(2 * G2_PF == 4, "G2_AHEAD splits the four groups in two batches");
    float4 HLa[G2_PF][2], HRa[G2_PF][2], HLb[G2_PF][2], HRb[G2_PF][2];
    load_spectra(0, HLa, HRa);
    g2_forward<false>(s[half], t_lo, t_hi, tt);
    load_spectra(G2_PF, HLb, HRb);
    products(0, HLa, HRa);
    products(G2_PF, HLb, HRb);
#else
    g2_forward<false>(s[half], t_lo, t_hi, tt);
#pragma unroll
    for (int rp = 0; rp < 4; rp += G2_PF) {
        float4 HLp[G2_PF][2], HRp[G2_PF][2];
        load_spectra(rp, HLp, HRp);
        products(rp, HLp, HRp);
    }
#endif
    __syncthreads();  // bin 0 reads mirrored entries that other threads own: every read before any write
    {
        int t0 = threadIdx.x;
        asm volatile("" : "+v"(t0));
        float2* y0 = &s[0][G2_P(2 * t0)];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            constexpr int RS = 2 * G2_THREADS + 2 * G2_THREADS / 32;
            vx_st(y0 + r * RS, yl[2 * r]);
            vx_st(y0 + r * RS + 1, yl[2 * r + 1]);
            vx_st(y0 + G2_LDS + r * RS, yr[2 * r]);
            vx_st(y0 + G2_LDS + r * RS + 1, yr[2 * r + 1]);
        }
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) {
#pragma unroll
        for (int r = 0; r < G2_PW; r++) xw[r] = window_row(item + (int)gridDim.x, r);
    }
    g2_inverse<false>(s[half], t_lo, t_hi, tt);
    const float sc = 1.0f / (float)G2_N;
    float4* dst = Yc + (size_t)bin * ycap + t_c0;
    for (int t = threadIdx.x; t < nout; t += G2_THREADS) {
        const float2 a = s[0][G2_P(t + taps - 1)], b = s[1][G2_P(t + taps - 1)];
        dst[t] = make_float4(a.x * sc, a.y * sc, b.x * sc, b.y * sc);
    }
    __syncthreads();  // the buffers are free for the next item
  }
}


// ---------------------------------------------------------------------------
// k_g2_duo: the fused second-level transform as ONE workgroup of 1024 threads per CU whose two halves ("groups") run
// k_g2_mac's per-item sequence on a buffer each, ONE PHASE APART, in barrier lockstep (round 3).
//
// Why.  Measured on k_g2_mac (profiles/r3_g2_ablation.md): with its butterflies and LDS passes removed the launch still
// takes 238 of 364 us - the memory phases alone - and with two independent workgroups per CU the transforms' time is
// simply added on top: a workgroup that is alone in a memory phase does not fill the CU's memory pipe, two workgroups in
// memory phases at once share it, and nothing makes the two alternate (a second co-resident workgroup buys 14 %, where
// perfect alternation would buy ~50 %).  A CU pulls HBM-class data at the chip's rate / 256 whatever is in flight, so
// the HBM side is only busy while EVERY CU has requests outstanding.
// Here the alternation is built in.  An item is four phases of G2D_NB barriers each -
//   F    window loads issued at once, four idle barriers, wait + unpack                      (memory: HBM read)
//   C12  forward transforms of x1 and x2                                                     (VALU / LDS)
//   P    products against the four paths' spectra, streamed one entry pair ahead             (memory: L2 read)
//   C34  inverse transforms of Y_L and Y_R, the stores issued behind the last pass           (VALU / LDS; HBM write drains under F)
// - every barrier is the whole workgroup's s_barrier, and group 1 starts one phase late: group 0's C12 runs beside
// group 1's F, its P beside group 1's C12, its C34 beside group 1's P, its next F beside group 1's C34.  A memory
// phase always has a compute phase of the other group beside it, by construction instead of by chance.  Barriers
// inside a memory phase cost the waiting group nothing (loads and stores stay in flight across s_barrier: it is issued
// bare, behind s_waitcnt lgkmcnt(0) only); a group without an item in a round, and the partner of a bin-0 item's second
// run, execute the same number of barriers empty.
// Persistent: grid = 8 k workgroups (<= CUs); the 2 * grid / 8 workers of an XCD lane (blockIdx & 7, speed only) walk
// the lane's items in order - chunk after chunk of a bin, bin after bin - so that a bin's second-level spectra and the
// overlap of adjacent windows are read into that XCD's L2 once, as in k_g2_mac.
// Same arithmetic in the same order per item as k_g2_mac: bit-identical sums.
// Bounds: as at k_g2_mac_wide (items via xq < nitems / 8, window slots masked into the ring, spectrum rows, t_c0 + t <
// T <= ycap); LDS: G2_P(8191) < G2_LDS per group.
// ---------------------------------------------------------------------------
#define G2D_THREADS 1024
#define G2D_NB 5
__device__ __forceinline__ void g2d_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void g2d_bars(int n) {
    for (int i = 0; i < n; i++) g2d_bar();
}

__global__ __launch_bounds__(G2D_THREADS, 4) void k_g2_duo(const float4* __restrict__ fdl, int ring, int slot0, int T, int chunk_t,
                                                            int taps, Fft2Voices vv, float4* __restrict__ Yc, int ycap, int nitems,
                                                            int nrounds, int solo) {
    __shared__ float2 s2[2][G2_LDS];
    __shared__ float2 t_lo[128], t_hi[64];
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 9));
    float2* s = s2[grp];
    const int nch = nitems >> 8, nxq = nitems >> 3;
    g2_tables(t_lo, t_hi);
    __syncthreads();
    if (grp) g2d_bars(G2D_NB);  // group 1 runs one phase behind group 0
    // solo (measurement, MCCONV_G2_DUO_SOLO=1): group 0 takes every item, group 1 only keeps the barriers company
    const int lane8 = blockIdx.x & 7, wpl = (int)(gridDim.x >> 3) * (solo ? 1 : 2), jw = (int)(blockIdx.x >> 3) * (solo ? 1 : 2);
#if G2_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = 0, st_now;
    int st_items = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define G2D_STAMP(k)                                                                   \
    do {                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if ((k) >= 0) st_acc[(k) < 0 ? 0 : (k)] += st_now - st_prev;                   \
        st_prev = st_now;                                                              \
    } while (0)
#else
#define G2D_STAMP(k) do { } while (0)
#endif
    constexpr int ROWS = G2_N / G2B_THREADS;                    // 16 window entries per thread and sequence
    constexpr int PS = 2 * G2B_THREADS + 2 * G2B_THREADS / 32;  // ... of entry pairs 2 j, 2 (j + 512)
    for (int rnd = 0; rnd < nrounds; rnd++) {
        const int xq0 = rnd * wpl + jw, xq = xq0 + (solo ? 0 : grp);
        const bool valid = xq < nxq && !(solo && grp);
        // a bin-0 item runs twice (z against h1, conj z against h2): its partner keeps it company with empty barriers
        const bool twice = lane8 == 0 && xq0 < nch;  // (xq0 < nch: group 0's item, and with it or without it group 1's, is bin 0's)
        const int bin = (xq / nch) * 8 + lane8, chunk = xq % nch;
        const int t_c0 = chunk * chunk_t, nout = min(chunk_t, T - t_c0), L = nout + taps - 1;
        const float4* fk = fdl + (size_t)bin * ring;
        const int sb = slot0 + t_c0 - (taps - 1);
        for (int pass = 0; pass < (twice ? 2 : 1); pass++) {
            if (!valid || (pass && bin != 0)) {
                g2d_bars(4 * G2D_NB);
                continue;
            }
            const float cj = pass ? -1.0f : 1.0f;  // second run of bin 0: conj(z)
            const int row = pass ? 256 : bin;
            int tt = threadIdx.x & (G2B_THREADS - 1);
            asm volatile("" : "+v"(tt));
            G2D_STAMP(-1);
            // ---- F: the window (16 bytes per slot carry both inputs), all rows requested at once
            v2f x1[ROWS], x2[ROWS];
            {
                float4 x[ROWS];
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    const int n = tt + G2B_THREADS * r;
                    x[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n < L) x[r] = fk[(sb + n) & (ring - 1)];
                }
                g2d_bars(G2D_NB - 1);  // (the other group's inverse transforms run meanwhile)
                G2D_STAMP(4);  // F, part 1: the loads issued + the other group's first four segments
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    x1[r] = v2f{x[r].x, cj * x[r].y};
                    x2[r] = v2f{x[r].z, cj * x[r].w};
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                G2D_STAMP(5);  // F, part 2: what the loads still needed after that
            }
            g2d_bar();
            G2D_STAMP(0);
            // ---- C12: forward transforms; the first pass on the registers just loaded (see k_g2_mac)
            g2_pair<false, 9, false, true>(s, t_lo, t_hi, tt, tt, x1);  // quarter lengths 2048, 512
            g2d_bar();
            g2_pair<false, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);  // 128, 32
            G2B_WAVE_SYNC();
            g2_pair<false, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);  // 8, 2
            g2d_bar();
            v2f X1[ROWS];
            {
                asm volatile("" : "+v"(tt));
                const float2* pp = &s[G2_P(2 * tt)];
#pragma unroll
                for (int r = 0; r < ROWS / 2; r++) {
                    const v2f a = vx_ld(pp + r * PS), b = vx_ld(pp + r * PS + 1);
                    X1[2 * r] = a + b;
                    X1[2 * r + 1] = a - b;
                }
            }
            g2d_bar();  // every thread has its X1 entries: the buffer is free for x2
            g2_pair<false, 9, false, true>(s, t_lo, t_hi, tt, tt, x2);
            g2d_bar();
            g2_pair<false, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
            G2B_WAVE_SYNC();
            g2_pair<false, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
            g2d_bar();
            G2D_STAMP(1);
            // ---- P: products (as in k_g2_mac); barriers after entry pairs 0, 3, 4, 5 and 7 pair its pieces with the other
            // group's forward passes (short, long, the register hand-off, short, long)
            {
                asm volatile("" : "+v"(tt));
                float2* pp = &s[G2_P(2 * tt)];
                constexpr int NP = ROWS / 2, RING = G2B_AHEAD + 1;
                float4 HLq[RING][2], HRq[RING][2];
                const char* hrow[2] = {reinterpret_cast<const char*>(vv.h0[0] + (size_t)row * G2_N),
                                       reinterpret_cast<const char*>(vv.h1[0] + (size_t)row * G2_N)};
                auto request = [&](int r) {
                    const unsigned off = ((unsigned)tt + G2B_THREADS * (unsigned)r) * 16u;
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        HLq[r % RING][i] = *reinterpret_cast<const float4*>(hrow[i] + off);
                        HRq[r % RING][i] = *reinterpret_cast<const float4*>(hrow[i] + (size_t)257 * G2_N * sizeof(float2) + off);
                    }
                };
#pragma unroll
                for (int r = 0; r < G2B_AHEAD; r++) request(r);
#pragma unroll
                for (int r = 0; r < NP; r++) {
                    if (r + G2B_AHEAD < NP) request(r + G2B_AHEAD);
                    const unsigned off = ((unsigned)tt + G2B_THREADS * (unsigned)r) * 16u;
                    const v2f a = vx_ld(pp + r * PS), b = vx_ld(pp + r * PS + 1);
                    const v2f S[2][2] = {{X1[2 * r], X1[2 * r + 1]}, {a + b, a - b}};
                    v2f aL0 = v2f{0.f, 0.f}, aL1 = aL0, aR0 = aL0, aR1 = aL0;
#pragma unroll
                    for (int i = 0; i < 2; i++) {
#pragma unroll
                        for (int vi = 0; vi < MC_MAXV; vi++) {
                            if (vi >= vv.n) break;
                            const float gl = i == 0 ? vv.g[vi].x : vv.g[vi].y, gr = i == 0 ? vv.g[vi].z : vv.g[vi].w;
                            float4 HL, HR;
                            if (vi == 0) {
                                HL = HLq[r % RING][i];
                                HR = HRq[r % RING][i];
                            } else {
                                const char* hb = reinterpret_cast<const char*>((i == 0 ? vv.h0[vi] : vv.h1[vi]) + (size_t)row * G2_N);
                                HL = *reinterpret_cast<const float4*>(hb + off);
                                HR = *reinterpret_cast<const float4*>(hb + (size_t)257 * G2_N * sizeof(float2) + off);
                            }
                            aL0 += gl * vx_mul(S[i][0], v2f{HL.x, HL.y});
                            aL1 += gl * vx_mul(S[i][1], v2f{HL.z, HL.w});
                            aR0 += gr * vx_mul(S[i][0], v2f{HR.x, HR.y});
                            aR1 += gr * vx_mul(S[i][1], v2f{HR.z, HR.w});
                        }
                    }
                    vx_st(pp + r * PS, aL0 + aL1);  // the inverse transform's radix-2 stage on the way out
                    vx_st(pp + r * PS + 1, aL0 - aL1);
                    X1[2 * r] = aR0 + aR1;
                    X1[2 * r + 1] = aR0 - aR1;
                    __builtin_amdgcn_sched_barrier(0);
                    if (r == 0 || r == 3 || r == 4 || r == 5 || r == NP - 1) g2d_bar();
                }
            }
            G2D_STAMP(2);
            // ---- C34: inverse transforms; the stores are issued behind the last pass and drain under the next phase
            v2f yl[ROWS];
            g2_pair<true, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
            G2B_WAVE_SYNC();
            g2_pair<true, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
            g2d_bar();
            g2_pair<true, 9, true, false>(s, t_lo, t_hi, tt, tt, yl);
            g2d_bar();
            {
                asm volatile("" : "+v"(tt));
                float2* pp = &s[G2_P(2 * tt)];
#pragma unroll
                for (int r = 0; r < ROWS / 2; r++) {
                    vx_st(pp + r * PS, X1[2 * r]);
                    vx_st(pp + r * PS + 1, X1[2 * r + 1]);
                }
            }
            g2d_bar();
            {
                v2f yr[ROWS];
                g2_pair<true, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
                G2B_WAVE_SYNC();
                g2_pair<true, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
                g2d_bar();
                g2_pair<true, 9, true, false>(s, t_lo, t_hi, tt, tt, yr);
                asm volatile("" : "+v"(tt));
                const float sc = 1.0f / (float)G2_N;
                float4* dst = Yc + (size_t)bin * ycap + t_c0;
#pragma unroll
                for (int m = 0; m < ROWS; m++) {
                    const int t = tt + G2B_THREADS * m - (taps - 1);
                    if (t >= 0 && t < nout) {
                        float4 y = make_float4(yl[m].x * sc, yl[m].y * sc, yr[m].x * sc, yr[m].y * sc);
                        if (pass) {
                            const float4 o = dst[t];
                            y = make_float4(o.x + y.x, o.y + y.y, o.z + y.z, o.w + y.w);
                        }
                        dst[t] = y;
                    }
                }
            }
            g2d_bar();  // the buffer is free for the next run
            G2D_STAMP(3);
#if G2_STAMPS
            st_items++;
#endif
        }
    }
    if (!grp) g2d_bars(G2D_NB);
#if G2_STAMPS
    if ((threadIdx.x & 511) == 0 && (blockIdx.x == 3 || blockIdx.x == 137 || blockIdx.x == 200))
    {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - st_t0, dr = __builtin_amdgcn_s_memrealtime() - st_r0;
        printf("g2 wg %d grp %d items %d: F %llu (= to the 4th barrier %llu + loads still out %llu + last barrier %llu) C12 %llu P %llu C34 %llu (cycles per item); workgroup %llu cycles in %llu x 10 ns = %.0f MHz\n",
               (int)blockIdx.x, grp, st_items, (st_acc[0] + st_acc[4] + st_acc[5]) / st_items, st_acc[4] / st_items, st_acc[5] / st_items, st_acc[0] / st_items,
               st_acc[1] / st_items, st_acc[2] / st_items, st_acc[3] / st_items, dt, dr, (double)dt / (double)dr * 100.0);
    }
#endif
}

