// The JACK path's kernels (one period per call): the tail of a 256-frame period (k_tail1; tail1_body, tail1_helper), the same fused with the
// sweep of the period after next (k_jack), the tail of 512- and 1024-frame periods (k_tailp).  Included by kernels.hip.h, behind the
// kernels and helpers they build on (fft512_wave, cmac, corr_terms, retired_at, write_history, mac_stream_body, drop_period_fft_body) -
// not a translation unit of its own.  Reference: Convolution::onProcess, conv.cu:287-466.
#pragma once

// ---------------------------------------------------------------------------
// Single-block (JACK) path.  k_mac_stream sums partitions p >= 2, which depend only on blocks at least two periods old; the tail
// (tail1_body below, one workgroup) does everything that needs the new block or the previous one: the p = 1 term against the previous
// block's spectrum, the p = 0 term against the new block - in the time domain when the tail had to wait for the period, in the
// frequency domain when the period was already there -, overlap-add with the previous segment, Q1/Q2 prefix update, predelay, Q8,
// clamp, dry mix, the output as tagged granules to mapped host memory; then the block's spectrum for the delay line.
// Replaces, for nframes = 256, the whole body of onProcess (conv.cu:321-451).
// ---------------------------------------------------------------------------
struct VoiceSet {
    int n;                      // voices with a partition-0 term
    int vid[MC_MAXV];           // their gain-table rows
    const float4* H0[MC_MAXV];  // spectra of the voice's IR for input 1 / input 2
    const float4* H1[MC_MAXV];
};

// Everything one period's tail needs (kernel arguments of k_tail1 and of the fused k_jack).
struct TailArgs {
    const float *in1, *in2;  // the period, mapped host memory
    VoiceSet vset;
    int pstride_ir;
    float4 *fdl, *slotgain;
    int ring, slot0;
    const float4* part;
    int nsum;
    const BlockParams* ptab;
    float* seg;
    int sr, seg0;
    float* wet;
    int wr;
    double* cring;
    int rc;
    VoiceSums vs;
    double inv_n;
    int compat;
    int64_t tabs0, predelay, n_ref;
    float *outL, *outR;  // mapped host memory
    const float2* g_tw;
    TailDrop td;
    uint2* fdl16;
    unsigned* done_flag;  // mapped: the completion word the host spins on
    unsigned seq;
    Retired ret;
    // A tail launched one call ahead parks here until the host rings: *bell == {seq, command 0} go, {seq, 1} give
    // up without touching any state; after park_ticks (100 MHz) without either it gives up on its own and says so
    // in *exited.  bell == null: not parked.
    const unsigned long long* bell;
    unsigned* exited;
    unsigned long long park_ticks;
    const float* drop;  // Q8 terms of the period's samples [2][256] (k_drop_period), or null
    // Tagged I/O (round 3, MCCONV_TAGGED_IO): the period and the output travel as 8-byte granules {value, sequence number}.  A
    // parked tail polls the period's own granules (every lane its two) instead of a doorbell followed by a second round trip
    // for the 2 KB it announces, and the host polls the output's granules instead of a completion word that has to wait behind
    // a system-scope release of every store of the kernel.  in_gran: [2][256] in device memory the CPU writes through the BAR
    // (used only when `bell` is set); out_gran: [2][256] in mapped host memory (null: plain outL / outR + completion word).
    const unsigned long long* in_gran;
    unsigned long long* out_gran;
    // 256-frame tail: which form a period took, counted in device memory {frequency domain, time domain} (mc_debug_read item 16), and
    // form != 0 (lab build, MCCONV_TAIL_FORM=td|fd): 1 = always the time-domain form (with its dry run), 2 = always the frequency-domain one
    unsigned* formcount;
    int form;
};

// ---------------------------------------------------------------------------
// The tail of one 256-frame JACK period (round 4 form).  Of the period's segment seg_t = sum_p (x_{t-p} (*) h_p) only the p = 0
// term depends on the period itself.  Everything else - the sweep's partial sums (p >= 2), partition 1 against the previous
// block, and the INVERSE TRANSFORM of their sum - is finished before the period arrives (a parked tail does it while it waits);
// after the arrival the p = 0 term is a direct 256 x 256 convolution of the period with the gain-weighted first 256 taps of the
// sounding IRs, added to the finished rest.  Two single-wavefront 512-point transforms and two workgroup barriers leave the
// critical path (the forward transform of the period, which later periods need in the delay line, and the segment's second half
// follow the output).  tail1_body_fft0 (lab build, -DMC_TAIL_FFT0) is the round-3 form: forward transform, partition 0 in the
// frequency domain, inverse transform - all behind the arrival.
//
// The direct convolution, register-tiled (a thread per output frame and tap reads 24 bytes of LDS per four multiply-adds: 3 us):
// a unit is NO consecutive output frames x NT consecutive taps; the taps {L<-in1, R<-in1, L<-in2, R<-in2} sit in registers before
// the period arrives, the unit's NT + NO - 1 input frames come from LDS (zeros in front of the period take care of the triangle),
// each (frame, tap) is two packed multiply-adds (v_pk_fma_f32 with the input sample broadcast), and the partial sums of the units
// of a frame meet in LDS in a fixed order.  One wavefront issues a packed multiply-add every 4.8 clocks however many share its
// SIMD (scripts/probes/pkfma_probe.hip), so the workgroup has EIGHT wavefronts: threads 256..511 (tail1_helper) only convolve.
// First half of the segment (frames 0..255: taps j <= m): 4 frames x 18 taps, 491 units on 512 threads, 144 packed instructions
// each.  Second half (frames 256 + r: taps j > r), behind the output: the same triangle with both sequences reversed, 8 frames x
// 18 taps, 249 units on the helper threads while wave 0 transforms the period for the delay line.
// ---------------------------------------------------------------------------
#define TD_PAD 32  // zeros in front of the period in LDS (a unit's window starts up to NT + NO - 2 frames before frame 0)
__device__ __forceinline__ void td_fma_in1(v2f& acc, v2f h, v2f x) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(h), "v"(x)); }
__device__ __forceinline__ void td_fma_in2(v2f& acc, v2f h, v2f x) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(h), "v"(x)); }
// unit u of the triangle's tiling: output frames [NO a, NO a + NO) need taps 0 .. NO a + NO - 1, i.e. ceil((NO a + NO) / NT) units of NT taps
template <int NT, int NO>
__device__ __forceinline__ bool td_unit(int u, int& a, int& c) {
    for (a = 0; a < MC_B / NO; a++) {
        const int n = (NO * a + NO + NT - 1) / NT;
        if (u < n) break;
        u -= n;
    }
    c = u;
    return a < MC_B / NO;
}
// taps of a unit (REV: of the reversed tap sequence) from LDS into registers: h1 = {L<-in1, R<-in1}, h2 = {L<-in2, R<-in2}
template <int NT, bool REV>
__device__ __forceinline__ void td_taps(const float4* s_hc, int c, bool on, v2f (&h1)[NT], v2f (&h2)[NT]) {
#pragma unroll
    for (int jj = 0; jj < NT; jj++) {
        const int j = NT * c + jj;
        float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on && j < MC_B) h = s_hc[REV ? MC_B - 1 - j : j];
        h1[jj] = v2f{h.x, h.y};
        h2[jj] = v2f{h.z, h.w};
    }
}
// the unit's window of the period: xpad = the (possibly reversed) period with TD_PAD zeros in front
template <int NT, int NO>
__device__ __forceinline__ void td_window(const float2* xpad, int a, int c, v2f (&w)[NT + NO - 1]) {
    const float2* w0 = xpad + TD_PAD + NO * a - NT * c - (NT - 1);  // frame of window entry 0 (>= -TD_PAD)
#pragma unroll
    for (int q = NT + NO - 2; q >= 0; q--) w[q] = vx_of(w0[q]);  // (in the order of use: tap 0 takes the last NO entries)
}
// the unit's NO partial sums {L, R}.  A packed multiply-add can follow one on the same accumulator after eight others (measured:
// 5.8 clocks per instruction with eight chains, 9.1 with four - scripts/probes/pkfma_tile_probe.hip), so four outputs keep
// the two inputs' sums apart until the end
template <int NT, int NO>
__device__ __forceinline__ void td_tile(const v2f (&w)[NT + NO - 1], const v2f (&h1)[NT], const v2f (&h2)[NT], v2f (&acc)[NO]) {
    v2f acc2[NO];
#pragma unroll
    for (int o = 0; o < NO; o++) acc[o] = acc2[o] = v2f{0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < NT; jj++) {
#pragma unroll
        for (int o = 0; o < NO; o++) td_fma_in1(acc[o], h1[jj], w[NT - 1 + o - jj]);
#pragma unroll
        for (int o = 0; o < NO; o++) td_fma_in2(NO < 8 ? acc2[o] : acc[o], h2[jj], w[NT - 1 + o - jj]);
    }
    if (NO < 8) {
#pragma unroll
        for (int o = 0; o < NO; o++) acc[o] += acc2[o];
    }
}
// a unit's sums into the frame's slots
template <int NO>
__device__ __forceinline__ void td_store(float2 (*s_pc)[MC_B], int a, int c, const v2f (&acc)[NO]) {
    float4* dst = reinterpret_cast<float4*>(&s_pc[c][NO * a]);
#pragma unroll
    for (int o = 0; o < NO / 2; o++) dst[o] = make_float4(acc[2 * o].x, acc[2 * o].y, acc[2 * o + 1].x, acc[2 * o + 1].y);
}
// sum over the wavefront, in every lane: two quad exchanges and two mirrors inside the rows of 16 lanes (DPP: no LDS crossbar
// round trips as __shfl_xor takes), then the four row sums
__device__ __forceinline__ float td_wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
#define TD_NT1 18  // first half: 4 frames x 18 taps, 491 units on the workgroup's 512 threads
#define TD_NO1 4
#define TD_NT2 18  // second half: 8 frames x 18 taps, 249 units on the 256 helper threads
#define TD_NO2 8
#define TAIL1_THREADS 512
#define TD_SLOTS ((MC_B + TD_NT1 - 1) / TD_NT1)  // units a frame's sum can have (15 in either half)

// the first look's verdict (workgroup-uniform: s_here is read behind a barrier)
__device__ __forceinline__ bool tail1_here(const TailArgs& A, const int* s_here) {
    if (A.form) return A.form == 2;
    return !A.bell || (s_here[0] && s_here[1] && s_here[2] && s_here[3]);  // (not parked: the period is in place)
}
// Threads 256..511 of the tail's workgroup: a unit of each half's direct convolution, and every barrier the others pass until then
// (a wavefront that has ended no longer counts at a barrier).
__device__ __forceinline__ void tail1_helper(const TailArgs& A, const float4* s_hc, const float2* s_xp, const float2* s_xr, float2 (*s_pc)[MC_B],
                                             const int* s_here, const int* s_abort, const int* s_go) {
    const int tid = threadIdx.x;
    __syncthreads();  // the taps are in LDS, the first look has been taken
    if (tail1_here(A, s_here)) return;  // the period is in place: the frequency-domain form
    int ua, uc;
    const bool uon = td_unit<TD_NT1, TD_NO1>(tid, ua, uc);
    v2f h1r[TD_NT1], h2r[TD_NT1];
    td_taps<TD_NT1, false>(s_hc, uc, uon, h1r, h2r);
    __syncthreads();  // the rest's inverse transform
    const bool inblock = A.predelay > 0 && A.predelay < MC_B;
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        const bool live = pass == 1;
        if (live && A.bell && !A.in_gran) {
            __syncthreads();  // lane 0 has heard the doorbell
            if (!*s_go) return;
        }
        __syncthreads();  // the period is in LDS
        if (live && *s_abort) return;
        v2f acc[TD_NO1], win[TD_NT1 + TD_NO1 - 1];
        td_window<TD_NT1, TD_NO1>(s_xp, uon ? ua : 0, uon ? uc : 0, win);
        td_tile<TD_NT1, TD_NO1>(win, h1r, h2r, acc);
        if (uon) td_store<TD_NO1>(s_pc, ua, uc, acc);
        __syncthreads();  // the units' sums are in LDS
        if (inblock) __syncthreads();
    }
    __syncthreads();  // everyone has read the first half's sums
    {
        int a2, c2;
        const bool on2 = td_unit<TD_NT2, TD_NO2>(tid - MC_B, a2, c2);
        v2f g1[TD_NT2], g2[TD_NT2], acc[TD_NO2], win[TD_NT2 + TD_NO2 - 1];
        td_taps<TD_NT2, true>(s_hc, c2, on2, g1, g2);
        td_window<TD_NT2, TD_NO2>(s_xr, on2 ? a2 : 0, on2 ? c2 : 0, win);
        td_tile<TD_NT2, TD_NO2>(win, g1, g2, acc);
        if (on2) td_store<TD_NO2>(s_pc, a2, c2, acc);
    }
    __syncthreads();  // the second half's sums are in LDS
}

__device__ __forceinline__ void tail1_body(const TailArgs& A) {
    const float* in1 = A.in1;
    const float* in2 = A.in2;
    const VoiceSet& vset = A.vset;
    const int pstride_ir = A.pstride_ir;
    float4* __restrict__ fdl = A.fdl;
    float4* __restrict__ slotgain = A.slotgain;
    const int ring = A.ring, slot0 = A.slot0;
    const float4* __restrict__ part = A.part;
    const int nsum = A.nsum;
    const BlockParams* __restrict__ ptab = A.ptab;
    float* __restrict__ seg = A.seg;
    const int sr = A.sr, seg0 = A.seg0;
    float* __restrict__ wet = A.wet;
    const int wr = A.wr;
    double* __restrict__ cring = A.cring;
    const int rc = A.rc;
    const VoiceSums& vs = A.vs;
    const double inv_n = A.inv_n;
    const int compat = A.compat;
    const int64_t tabs0 = A.tabs0, predelay = A.predelay, n_ref = A.n_ref;
    float* __restrict__ outL = A.outL;
    float* __restrict__ outR = A.outR;
    const float2* __restrict__ g_tw = A.g_tw;
    const TailDrop& td = A.td;
    uint2* __restrict__ fdl16 = A.fdl16;
    unsigned* __restrict__ done_flag = A.done_flag;
    const unsigned seq = A.seq;
    const Retired& ret = A.ret;
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[FFT_WAVE_LDS];
    __shared__ float4 s_y[MC_NB];                        // Y_L, Y_R of everything but partition 0
    __shared__ __align__(16) float4 s_hc[MC_B];          // gain-weighted first 256 taps {L<-in1, R<-in1, L<-in2, R<-in2}
    __shared__ __align__(16) float2 s_xp[TD_PAD + MC_B]; // the period {in1, in2} behind TD_PAD zeros
    __shared__ __align__(16) float2 s_xr[TD_PAD + MC_B]; // the same, reversed in time
    __shared__ __align__(16) float2 s_pc[TD_SLOTS][MC_B];  // the units' partial sums of a frame
    __shared__ float s_wet[2][MC_B];
    __shared__ float4 s_red[4];  // per wave {S1, S2, A1, A2}
    __shared__ int s_abort;      // a parked tail gives up (told to, or the host stayed away)
    __shared__ int s_here[4];    // the period was there at the first look (per wave)
    __shared__ int s_go;         // doorbell path: the period is there (0: give up)
    __shared__ __align__(16) double s_kq[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid >= MC_B) {  // the second four wavefronts only convolve
        tail1_helper(A, s_hc, s_xp, s_xr, s_pc, s_here, &s_abort, &s_go);
        return;
    }
    const BlockParams& bp = ptab[0];
    const int m = tid;
    const int64_t tau0 = tabs0 * MC_B, tau = tau0 + m, u = tau - predelay;

    // ---- before the period: every load whose address is known now, in one round of memory latency
    const float2 tw0 = g_tw[tid], tw1 = g_tw[tid + 256];
    float4 ysum = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float4* src = part + (size_t)tid * nsum;
        for (int c = 0; c < nsum; c++) {
            const float4 a = src[c];
            ysum.x += a.x;
            ysum.y += a.y;
            ysum.z += a.z;
            ysum.w += a.w;
        }
    }
    // partition 1 pairs with the previous block: its spectrum and its slot's gains come from the delay line
    float4 h0w[MC_MAXV], h1w[MC_MAXV], g1w[MC_MAXV];
    float4 h0v[MC_MAXV], h1v[MC_MAXV];  // partition 0 as spectra: for a period that is already there (below)
    const int slot1 = (slot0 + ring - 1) & (ring - 1);
    const float4 xprev = fdl[(size_t)tid * ring + slot1];
    float4 hc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int vi = 0; vi < MC_MAXV; vi++) {  // constant indices: runtime-indexed kernel-argument arrays go to scratch
        h0w[vi] = h1w[vi] = g1w[vi] = h0v[vi] = h1v[vi] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vi < vset.n) {
            h0v[vi] = vset.H0[vi][(size_t)tid * pstride_ir];
            h1v[vi] = vset.H1[vi][(size_t)tid * pstride_ir];
            h0w[vi] = vset.H0[vi][(size_t)tid * pstride_ir + 1];
            h1w[vi] = vset.H1[vi][(size_t)tid * pstride_ir + 1];
            g1w[vi] = slotgain[(size_t)vset.vid[vi] * ring + slot1];
            // tap `tid` of the voice's two IRs (the taps the Q8 pass keeps on the device), weighted with the gains this block carries
            const float* g = ptab->g[vset.vid[vi]];
            float2 t0 = make_float2(0.f, 0.f), t1 = t0;
#pragma unroll
            for (int v = 0; v < MC_MAXV; v++)
                if (v == vset.vid[vi]) {
                    if (tid < td.L0[v]) t0 = td.h0[v][tid];
                    if (tid < td.L1[v]) t1 = td.h1[v][tid];
                }
            hc.x += g[0] * t0.x;
            hc.y += g[2] * t0.y;
            hc.z += g[1] * t1.x;
            hc.w += g[3] * t1.y;
        }
    }
    const float* prv = seg + (size_t)((seg0 + sr - 1) & (sr - 1)) * 2 * FFT_N;
    const float prvL = prv[MC_B + m], prvR = prv[FFT_N + MC_B + m];
    float dwl = 0.f, dwr = 0.f;  // delayed wet sample when it predates this period
    if (u >= 0 && u < tau0) {
        dwl = wet[(size_t)(u & (wr - 1))];
        dwr = wet[(size_t)wr + (u & (wr - 1))];
    }
    float2 ra = make_float2(0.f, 0.f), rb = ra;
    if (tau < ret.end) {  // what blocks played under an earlier predelay still owe
        ra = retired_at(ret.mac, ret.rr, tau);
        rb = retired_at(ret.fix, ret.rr, tau);
    }
    const int64_t thi = u >> 8;
    int64_t tlo = tau - n_ref >= 0 ? ((tau - n_ref) >> 8) : -1;
    if (tlo < ret.b0 - 1) tlo = ret.b0 - 1;
    const bool corr_on = compat && u >= 0 && thi > tlo;
    double ca[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0}, cprev[4] = {0, 0, 0, 0};
    if (corr_on && thi != tabs0) {
        const double* pa = cring + (size_t)(thi & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) ca[c] = pa[c];
    }
    if (corr_on && tlo >= 0) {
        const double* pb = cring + (size_t)(tlo & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) cb[c] = pb[c];
    }
    if (tabs0 > 0) {  // the prefix sums up to the previous block (every thread extends them by this block's terms itself)
        const double* pp = cring + (size_t)((tabs0 + rc - 1) & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) cprev[c] = pp[c];
    }
    // this block's Q1/Q2 terms are linear in its sums {S1, S2, A1, A2} (corr_terms with unit sums gives the coefficients); through
    // LDS, so that they are computed before the period arrives and not where the compiler finds their first use
    if (tid == 64) {
        double d[4];
        corr_terms(make_float4(1.f, 0.f, 0.f, 0.f), bp, vs, inv_n, d);
        s_kq[0] = d[1];  // S1 -> D_R
        corr_terms(make_float4(0.f, 1.f, 0.f, 0.f), bp, vs, inv_n, d);
        s_kq[1] = d[0], s_kq[2] = d[1];  // S2 -> D_L, D_R
        corr_terms(make_float4(0.f, 0.f, 1.f, 0.f), bp, vs, inv_n, d);
        s_kq[3] = d[2], s_kq[4] = d[3];  // A1 -> Q_L, Q_R
        corr_terms(make_float4(0.f, 0.f, 0.f, 1.f), bp, vs, inv_n, d);
        s_kq[5] = d[2], s_kq[6] = d[3];  // A2 -> Q_L, Q_R
        s_kq[7] = 0.0;
    }
    // the dry mix of the block: read here - where it is used, behind the last barrier, the load's round trip through memory
    // (2000+ clocks) would sit between the period and its output
    const float4 dmix = make_float4(bp.d[0], bp.d[1], bp.d[2], bp.d[3]);
    float td_l = 0.f, td_r = 0.f;  // Q8: the cut terms of this period's samples come from blocks at least n_ref frames old (summed ahead)
    if (td.on) {
        td_l = A.drop[m];
        td_r = A.drop[MC_B + m];
    }
    s_tw[tid] = tw0;
    s_tw[tid + 256] = tw1;
    s_hc[tid] = hc;
    if (tid < TD_PAD) s_xp[tid] = s_xr[tid] = make_float2(0.f, 0.f);
    if (tid == 0) s_abort = 0;
#pragma unroll
    for (int c = 0; c < TD_SLOTS; c++) s_pc[c][tid] = make_float2(0.f, 0.f);  // (a frame's sum takes all slots; its units fill the first few)
    float4 yrest;
    {  // bin tid of everything but partition 0: the sweep's partial sums + partition 1 against the previous block
        const int k = tid;
        float4 y = ysum;
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++) {
            if (vi >= vset.n) break;
            const float4 p0 = h0w[vi], p1 = h1w[vi], gq = g1w[vi];
            float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            if (k == 0) {
                cmac<true>(a0, p0.x, p0.y, xprev.x, xprev.y);
                cmac<true>(a1, p1.x, p1.y, xprev.z, xprev.w);
                cmac<true>(a2, p0.z, p0.w, xprev.x, xprev.y);
                cmac<true>(a3, p1.z, p1.w, xprev.z, xprev.w);
            } else {
                cmac<false>(a0, p0.x, p0.y, xprev.x, xprev.y);
                cmac<false>(a1, p1.x, p1.y, xprev.z, xprev.w);
                cmac<false>(a2, p0.z, p0.w, xprev.x, xprev.y);
                cmac<false>(a3, p1.z, p1.w, xprev.z, xprev.w);
            }
            y.x += gq.x * a0.x + gq.y * a1.x;
            y.y += gq.x * a0.y + gq.y * a1.y;
            y.z += gq.z * a2.x + gq.w * a3.x;
            y.w += gq.z * a2.y + gq.w * a3.y;
        }
        s_y[k] = y;
        yrest = y;
    }
    // A first look for the period decides the form of the rest.
    // NOT THERE (the host idles between periods, as under jackd): there is time - the rest's inverse transform is done now, and then
    // a DRY RUN of everything between the period and its output.  That stretch is straight-line code executed once per launch, and
    // from an instruction cache that every launch starts cold it runs at 8-12 clocks per instruction (s_memtime stamps, four wavefronts: 288 packed
    // multiply-adds 2400 clocks, the whole stretch 4500); the second time, 1.4 us sooner, it comes out of the cache.
    // ALREADY THERE (calls back to back, or a period launched on arrival): nothing can be prepared, every instruction is fetched
    // cold, and what counts is how many there are until the kernel ends: the frequency-domain form below (partition 0 as one more
    // product in front of a single inverse transform, round 3's) has 3.6 us of them, the time-domain form 6.5 (its second half and
    // the period's transform follow the output).
    float xin1 = 0.f, xin2 = 0.f;
    if (A.bell && A.in_gran) {
        const unsigned long long g1 = __hip_atomic_load(A.in_gran + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long g2 = __hip_atomic_load(A.in_gran + MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool here = __all((unsigned)(g1 >> 32) == seq && (unsigned)(g2 >> 32) == seq);
        if (here) {
            xin1 = __uint_as_float((unsigned)g1);
            xin2 = __uint_as_float((unsigned)g2);
        }
        if (lane == 0) s_here[wave] = here;
    } else if (A.bell && tid == 0) {
        const unsigned long long v = __hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_here[0] = s_here[1] = s_here[2] = s_here[3] = (unsigned)v == seq && (v >> 32) == 0;
    }
    __syncthreads();
    const bool here = tail1_here(A, s_here);
    if (here) {
        // ---- the period is in place: forward transform, partition 0 in the frequency domain, one inverse transform
        float4* s_x = reinterpret_cast<float4*>(&s_pc[0][0]);  // raw spectra of the new block {X1, X2} (the time-domain form's partial sums are not in use)
        __shared__ double s_c[4];
        __shared__ float4 s_sa;
#ifdef MC_JACK_TRACE
        unsigned long long t_start = 0, c_start = 0;
#endif
        if (A.form == 2 && A.bell) {  // (lab build: this form forced on a parked tail - wait here)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            if (A.in_gran) {
                for (;;) {
                    const unsigned long long g1 = __hip_atomic_load(A.in_gran + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned long long g2 = __hip_atomic_load(A.in_gran + MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (__all((unsigned)(g1 >> 32) == seq && (unsigned)(g2 >> 32) == seq)) {
                        xin1 = __uint_as_float((unsigned)g1);
                        xin2 = __uint_as_float((unsigned)g2);
                        break;
                    }
                    if ((unsigned)__hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == seq) return;  // told to give up
                    if (__builtin_amdgcn_s_memrealtime() - t0 > A.park_ticks) {  // (every wave comes here within a poll of the others)
                        if (tid == 0) __hip_atomic_store(A.exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            } else {
                for (;;) {
                    const unsigned long long v = __hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned)v == seq && (v >> 32) != 0) return;  // told to give up
                    if ((unsigned)v == seq) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > A.park_ticks) {
                        if (tid == 0) __hip_atomic_store(A.exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        return;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            }
        }
        if (!(A.bell && A.in_gran)) {  // (tagged granules were read by the look)
            if (A.bell) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the period was written before the doorbell
            // (system scope: the period may sit in device memory the CPU wrote through the BAR - not to be served from a cache)
            xin1 = __hip_atomic_load(in1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            xin2 = __hip_atomic_load(in2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        float4 xs_keep[4];  // wave 0: the block's spectra, stored to the delay line at the end of the kernel
#ifdef MC_FD_WARM  // (measurement build: what the frequency-domain form would take out of a warm instruction cache - a dry pass first)
#pragma nounroll
        for (int fdpass = 0; fdpass < 2; fdpass++) {
#endif
#ifdef MC_JACK_TRACE
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start), "=s"(c_start) : "v"(xin1));  // (in program order: per trip of the loop)
#endif
        s_xp[TD_PAD + tid] = make_float2(xin1, xin2);
        __syncthreads();
        if (wave == 0) {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = s_xp[TD_PAD + lane + 64 * r];
#pragma unroll
            for (int r = 4; r < 8; r++) v[r] = make_float2(0.f, 0.f);
            fft512_wave<-1, false>(v, s_fft, s_tw, lane);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = lane + 64 * j;
                const float2 za = s_fft[k], zb = s_fft[(FFT_N - k) & (FFT_N - 1)];
                float2 x1, x2;
                if (k == 0) {
                    const float2 zn = s_fft[MC_B];
                    x1 = make_float2(za.x, zn.x);
                    x2 = make_float2(za.y, zn.y);
                    s_sa = make_float4(za.x, za.y, zn.x, zn.y);
                } else {
                    x1 = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y));
                    x2 = make_float2(0.5f * (za.y + zb.y), -0.5f * (za.x - zb.x));
                }
                const float4 xs = make_float4(x1.x, x1.y, x2.x, x2.y);
                s_x[k] = xs;
                xs_keep[j] = xs;
            }
        }
        __syncthreads();
        {  // bin tid: the rest + partition 0 of every voice's IRs against the new block
            const int k = tid;
            const float4 x = s_x[k];
            float4 y = yrest;
#pragma unroll
            for (int vi = 0; vi < MC_MAXV; vi++) {
                if (vi >= vset.n) break;
                const float* g = ptab->g[vset.vid[vi]];
                const float4 h0 = h0v[vi], h1 = h1v[vi];
                float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
                if (k == 0) {
                    cmac<true>(a0, h0.x, h0.y, x.x, x.y);
                    cmac<true>(a1, h1.x, h1.y, x.z, x.w);
                    cmac<true>(a2, h0.z, h0.w, x.x, x.y);
                    cmac<true>(a3, h1.z, h1.w, x.z, x.w);
                } else {
                    cmac<false>(a0, h0.x, h0.y, x.x, x.y);
                    cmac<false>(a1, h1.x, h1.y, x.z, x.w);
                    cmac<false>(a2, h0.z, h0.w, x.x, x.y);
                    cmac<false>(a3, h1.z, h1.w, x.z, x.w);
                }
                y.x += g[0] * a0.x + g[1] * a1.x;
                y.y += g[0] * a0.y + g[1] * a1.y;
                y.z += g[2] * a2.x + g[3] * a3.x;
                y.w += g[2] * a2.y + g[3] * a3.y;
            }
            s_y[k] = y;
        }
        __syncthreads();
        if (wave == 0) {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int n = lane + 64 * r;
                float2 w;
                if (n == 0) {
                    const float4 y = s_y[0];
                    w = make_float2(y.x, y.z);
                } else if (n == MC_B) {
                    const float4 y = s_y[0];
                    w = make_float2(y.y, y.w);
                } else if (n < MC_B) {
                    const float4 y = s_y[n];
                    w = make_float2(y.x - y.w, y.y + y.z);
                } else {
                    const float4 y = s_y[FFT_N - n];
                    w = make_float2(y.x + y.w, -y.y + y.z);
                }
                v[r] = w;
            }
            fft512_wave<+1, false>(v, s_fft, s_tw, lane);
        } else if (tid == 64) {
            // meanwhile: this block's Q1/Q2 terms and the new prefix entry (float64, serial)
            double d[4] = {0, 0, 0, 0};
            if (compat) corr_terms(s_sa, bp, vs, inv_n, d);
            for (int c = 0; c < 4; c++) s_c[c] = cprev[c] + d[c];
        }
        __syncthreads();
#ifdef MC_FD_WARM
        }
#endif
        // No global store is issued before the output has left: a barrier drains the vector-memory counter, so every
        // store ahead of it would put its acknowledgement latency on the critical path.
        float seg_lo[2], seg_hi[2], own_wet[2];
        {
            // overlap-add with the previous block's tail; this block's segments go to the ring
            const float sc = 1.0f / FFT_N;
            const float2 lo = s_fft[m], hi = s_fft[MC_B + m];
            seg_lo[0] = lo.x * sc;
            seg_lo[1] = lo.y * sc;
            seg_hi[0] = hi.x * sc;
            seg_hi[1] = hi.y * sc;
            own_wet[0] = seg_lo[0] + prvL;
            own_wet[1] = seg_lo[1] + prvR;
            s_wet[0][m] = own_wet[0];
            s_wet[1][m] = own_wet[1];
        }
        __syncthreads();
        {
            float wl = dwl, wr_ = dwr;
            if (u >= tau0) {  // inside this block: not yet visible through global memory
                wl = s_wet[0][u - tau0];
                wr_ = s_wet[1][u - tau0];
            }
            wl += ra.x + rb.x;
            wr_ += ra.y + rb.y;
            double cl = 0.0, cr = 0.0;
            if (corr_on) {
                if (thi == tabs0)
                    for (int c = 0; c < 4; c++) ca[c] = s_c[c];
                const double sg = (u & 1) ? -1.0 : 1.0;
                cl = (ca[0] - cb[0]) + sg * (ca[2] - cb[2]);
                cr = (ca[1] - cb[1]) + sg * (ca[3] - cb[3]);
            }
            if (td.on) {
                wl -= td_l;
                wr_ -= td_r;
            }
            const float vl = fminf(fmaxf((float)((double)wl + cl), -1.f), 1.f);
            const float vr = fminf(fmaxf((float)((double)wr_ + cr), -1.f), 1.f);
            const float yl = vl + xin1 * dmix.x + xin2 * dmix.y, yr = vr + xin1 * dmix.z + xin2 * dmix.w;
            if (A.out_gran) {  // the output as granules {value, sequence number}: on the host as soon as the posted writes land
                __hip_atomic_store(A.out_gran + m, ((unsigned long long)seq << 32) | __float_as_uint(yl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(A.out_gran + MC_B + m, ((unsigned long long)seq << 32) | __float_as_uint(yr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                outL[m] = yl;
                outR[m] = yr;
            }
            write_history(td, tau, tabs0, m, xin1, xin2, bp, rc);
        }
#ifdef MC_JACK_TRACE
        const unsigned long long t_out = __builtin_amdgcn_s_memrealtime();  // (behind the stores' acknowledgements: late by a round trip)
        const unsigned long long c_out = __builtin_amdgcn_s_memtime();
#endif
        // the state later periods need: delay-line slot, slot gains, segments, wet ring, Q1/Q2 prefix entry
        {
            float* cur = seg + (size_t)seg0 * 2 * FFT_N;
            cur[m] = seg_lo[0];
            cur[MC_B + m] = seg_hi[0];
            cur[FFT_N + m] = seg_lo[1];
            cur[FFT_N + MC_B + m] = seg_hi[1];
            wet[(size_t)(tau & (wr - 1))] = own_wet[0];
            wet[(size_t)wr + (tau & (wr - 1))] = own_wet[1];
            if (wave == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = lane + 64 * j;
                    fdl[(size_t)k * ring + slot0] = xs_keep[j];
                    if (fdl16) fdl16[(size_t)k * ring + slot0] = pack_half4(xs_keep[j], FDL16_SCALE);
                }
                if (lane < MC_MAXV)
                    slotgain[(size_t)lane * ring + slot0] = make_float4(bp.g[lane][0], bp.g[lane][1], bp.g[lane][2], bp.g[lane][3]);
            }
            if (m == 0) {
                double* o = cring + (size_t)(tabs0 & (rc - 1)) * 4;
                for (int c = 0; c < 4; c++) o[c] = s_c[c];
                if (A.formcount) A.formcount[0] += 1;  // (one tail at a time per engine)
            }
        }
        // publish completion to the host (see the end of the function)
#ifndef MC_JACK_TRACE
        if (A.out_gran) return;
#endif
        __syncthreads();
        if (tid == 0) {
#ifdef MC_JACK_TRACE
            reinterpret_cast<unsigned long long*>(done_flag)[1] = t_start;
            reinterpret_cast<unsigned long long*>(done_flag)[2] = __builtin_amdgcn_s_memrealtime();
            reinterpret_cast<unsigned long long*>(done_flag)[3] = t_out;
            reinterpret_cast<unsigned long long*>(done_flag)[4] = c_out - c_start;
            reinterpret_cast<unsigned long long*>(done_flag)[5] = reinterpret_cast<unsigned long long*>(done_flag)[6] = reinterpret_cast<unsigned long long*>(done_flag)[7] = 0;
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    // ---- the period is still to come
    int ua, uc;  // this thread's unit of the first half
    const bool uon = td_unit<TD_NT1, TD_NO1>(tid, ua, uc);
    v2f h1r[TD_NT1], h2r[TD_NT1];
    td_taps<TD_NT1, false>(s_hc, uc, uon, h1r, h2r);
    if (wave == 0) {  // the inverse transform of the rest: the segment without the period's own term
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = lane + 64 * r;
            float2 w;
            if (n == 0) {
                const float4 y = s_y[0];
                w = make_float2(y.x, y.z);
            } else if (n == MC_B) {
                const float4 y = s_y[0];
                w = make_float2(y.y, y.w);
            } else if (n < MC_B) {
                const float4 y = s_y[n];
                w = make_float2(y.x - y.w, y.y + y.z);
            } else {
                const float4 y = s_y[FFT_N - n];
                w = make_float2(y.x + y.w, -y.y + y.z);
            }
            v[r] = w;
        }
        fft512_wave<+1, false>(v, s_fft, s_tw, lane);
    }
    __syncthreads();
    const float sc = 1.0f / FFT_N;
    float pre_lo[2];
    float2* s_hi = reinterpret_cast<float2*>(&s_y[0]);  // the rest's second half, until the period's own term joins it behind the output (s_y has been read)
    {
        const float2 lo = s_fft[m], hi = s_fft[MC_B + m];
        pre_lo[0] = lo.x * sc, pre_lo[1] = lo.y * sc;
        s_hi[m] = make_float2(hi.x * sc, hi.y * sc);
    }

    // the Q1/Q2 terms of this thread's output sample, as far as they are known: everything but this block's own share (which
    // only a predelay below one block lets in); what the retired epochs still owe, what the cut at n_ref takes away and the
    // delayed wet sample itself where it predates this period: one addend per channel
    const float addl = (u >= tau0 ? 0.f : dwl) + ((ra.x + rb.x) - (td.on ? td_l : 0.f)), addr = (u >= tau0 ? 0.f : dwr) + ((ra.y + rb.y) - (td.on ? td_r : 0.f));
    const double csg = (u & 1) ? -1.0 : 1.0;
    const bool cown = corr_on && thi == tabs0;
    double cbase_l = 0.0, cbase_r = 0.0;
    if (corr_on) {
        const double* src = cown ? cprev : ca;
        cbase_l = (src[0] - cb[0]) + csg * (src[2] - cb[2]);
        cbase_r = (src[1] - cb[1]) + csg * (src[3] - cb[3]);
    }
    // ---- the period: pass 0 is the dry run (nothing is waited for, nothing is stored), pass 1 the period's
    float seg_lo[2] = {0.f, 0.f}, own_wet[2] = {0.f, 0.f};
    double cnow[4] = {0, 0, 0, 0};  // the Q1/Q2 prefix sums including this block
#ifdef MC_JACK_TRACE
    unsigned long long t_start = 0, c_start = 0, c_b1 = 0, c_fma = 0, c_b2 = 0, t_out = 0, c_out = 0;
#endif
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        const bool live = pass == 1;
        if (live && A.bell) {  // (not parked - this form forced in a lab build: the period is in place)
            if (A.in_gran) {
                // Parked, tagged input: every lane looks at its own two granules of the period; lane 0 also watches the doorbell word for
                // the "give up" command and the park time.  THREE looks are in flight, a third of a round trip through memory apart
                // (a look whose request passes the memory side before the period lands comes back empty: with one look at a time
                // the period is seen half a round trip + half a look-to-look distance after it lands, 0.7 us; with three, 0.45).
                // A wave leaves the loop when all its lanes hold this period's samples, or when wave 0 has said to leave (an LDS
                // word); the barrier behind the loop makes the decision the workgroup's.
                const unsigned long long* q1 = A.in_gran + tid;
                const unsigned long long* q2 = A.in_gran + MC_B + tid;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned long long g1[3], g2[3], gb = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    g1[k] = __hip_atomic_load(q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    g2[k] = __hip_atomic_load(q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (k == 0 && tid == 0) gb = __hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __builtin_amdgcn_s_sleep(8);
                }
#define TD_LOOK_HIT(k) __all((unsigned)(g1[k] >> 32) == seq && (unsigned)(g2[k] >> 32) == seq)
#define TD_LOOK_TAKE(k) xin1 = __uint_as_float((unsigned)g1[k]), xin2 = __uint_as_float((unsigned)g2[k])
#define TD_LOOK_AGAIN(k) \
    g1[k] = __hip_atomic_load(q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), g2[k] = __hip_atomic_load(q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                // (six rounds per trip of the loop: at the loop's head the compiler waits for everything in flight)
                for (;;) {
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        if (TD_LOOK_HIT(0)) {
                            TD_LOOK_TAKE(0);
                            goto td_period_seen;
                        }
                        TD_LOOK_AGAIN(0);
                        if (tid == 0) {
                            if ((unsigned)gb == seq && (gb >> 32) != 0) {
                                s_abort = 1;  // told to give up
                            } else if (__builtin_amdgcn_s_memrealtime() - t0 > A.park_ticks) {
                                __hip_atomic_store(A.exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                s_abort = 1;  // the host has been away for longer than the park time
                            }
                            gb = __hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        asm volatile("" ::: "memory");  // (the word is read from LDS every time; not through a generic pointer: a flat load would wait for the looks in flight)
                        if (s_abort) goto td_period_seen;
                        if (TD_LOOK_HIT(1)) {
                            TD_LOOK_TAKE(1);
                            goto td_period_seen;
                        }
                        TD_LOOK_AGAIN(1);
                        if (TD_LOOK_HIT(2)) {
                            TD_LOOK_TAKE(2);
                            goto td_period_seen;
                        }
                        TD_LOOK_AGAIN(2);
                    }
                }
            td_period_seen:;
#undef TD_LOOK_HIT
#undef TD_LOOK_TAKE
#undef TD_LOOK_AGAIN
                // (no barrier here: a wave that holds its samples stores them and meets the others at the barrier below; s_abort is
                // final for a wave only behind that barrier)
            } else {
                // Parked: one lane polls the mapped doorbell (a PCIe read per poll), the others wait at the barrier.
                if (tid == 0) {
                    int go = 1;
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        const unsigned long long v = __hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if ((unsigned)v == seq) {
                            go = (v >> 32) == 0;
                            break;
                        }
                        if (__builtin_amdgcn_s_memrealtime() - t0 > A.park_ticks) {
                            go = 0;
                            __hip_atomic_store(A.exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    s_go = go;
                }
                __syncthreads();
                if (!s_go) return;  // nothing has been written: the host launches this period again
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the period was written before the doorbell
            }
        }
#ifdef MC_JACK_TRACE  // diagnostic build: when the period's work started and ended (100 MHz), next to the completion word
        t_start = __builtin_amdgcn_s_memrealtime();
        c_start = __builtin_amdgcn_s_memtime();
#endif
        // (system scope: the period may sit in device memory the CPU wrote through the BAR - not to be served from a cache)
        if (live && !(A.bell && A.in_gran)) {  // (tagged granules are read by the looks above)
            xin1 = __hip_atomic_load(in1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            xin2 = __hip_atomic_load(in2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s_xp[TD_PAD + tid] = make_float2(xin1, xin2);
        s_xr[TD_PAD + MC_B - 1 - tid] = make_float2(xin1, xin2);
        __syncthreads();
        if (live && s_abort) return;  // nothing has been written: the host launches this period again
#ifdef MC_JACK_TRACE
        c_b1 = __builtin_amdgcn_s_memtime();
#endif
        // ---- partition 0 in the time domain, first half of the segment
#ifdef MC_JACK_TRACE
        c_fma = 0;
#endif
        {
            v2f acc[TD_NO1], win[TD_NT1 + TD_NO1 - 1];
            td_window<TD_NT1, TD_NO1>(s_xp, uon ? ua : 0, uon ? uc : 0, win);
            // the block's sums {S1, S2, A1, A2} (conv.cu:55-71: what the DC / Nyquist bins of its transform hold), per wave, while
            // the window is on its way from LDS
            {
                const float sg = (tid & 1) ? -1.f : 1.f;
                const float r0 = td_wave_sum(xin1), r1 = td_wave_sum(xin2), r2 = td_wave_sum(sg * xin1), r3 = td_wave_sum(sg * xin2);
                if (lane == 0) s_red[wave] = make_float4(r0, r1, r2, r3);
            }
            td_tile<TD_NT1, TD_NO1>(win, h1r, h2r, acc);
#ifdef MC_JACK_TRACE
            asm volatile("" ::"v"(acc[0]), "v"(acc[TD_NO1 - 1]));
            c_fma = __builtin_amdgcn_s_memtime();
#endif
            if (uon) td_store<TD_NO1>(s_pc, ua, uc, acc);
        }
        __syncthreads();
#ifdef MC_JACK_TRACE
        c_b2 = __builtin_amdgcn_s_memtime();
#endif
        {
            v2f sum = vx_of(s_pc[0][m]);
#pragma unroll
            for (int c = 1; c < TD_SLOTS; c++) sum += vx_of(s_pc[c][m]);
            seg_lo[0] = pre_lo[0] + sum.x;
            seg_lo[1] = pre_lo[1] + sum.y;
            own_wet[0] = seg_lo[0] + prvL;
            own_wet[1] = seg_lo[1] + prvR;
            const float4 q0 = s_red[0], q1 = s_red[1], q2 = s_red[2], q3 = s_red[3];
            const double S1 = (q0.x + q1.x) + (q2.x + q3.x), S2 = (q0.y + q1.y) + (q2.y + q3.y), A1 = (q0.z + q1.z) + (q2.z + q3.z),
                         A2 = (q0.w + q1.w) + (q2.w + q3.w);
            cnow[0] = cnow[1] = cnow[2] = cnow[3] = 0.0;  // (this block's terms; the prefix sums are added behind the output)
            if (compat) {
                cnow[0] = S2 * s_kq[1];
                cnow[1] = S1 * s_kq[0] + S2 * s_kq[2];
                cnow[2] = A1 * s_kq[3] + A2 * s_kq[5];
                cnow[3] = A1 * s_kq[4] + A2 * s_kq[6];
            }
        }
        // a predelay inside the block: the delayed sample is another thread's (not yet visible through global memory)
        const bool inblock = predelay > 0 && predelay < MC_B;
        if (inblock) {
            s_wet[0][m] = own_wet[0];
            s_wet[1][m] = own_wet[1];
            __syncthreads();
        }
        // No global store is issued before the output has left: a barrier drains the vector-memory counter, so every
        // store ahead of it would put its acknowledgement latency on the critical path.
        {
            float wl = 0.f, wr_ = 0.f;
            if (u >= tau0) {
                wl = inblock ? s_wet[0][u - tau0] : own_wet[0];
                wr_ = inblock ? s_wet[1][u - tau0] : own_wet[1];
            }
            wl += addl;
            wr_ += addr;
            double cl = cbase_l, cr = cbase_r;
            if (cown) {
                cl += cnow[0] + csg * cnow[2];
                cr += cnow[1] + csg * cnow[3];
            }
            const float x1 = xin1, x2 = xin2;
            const float vl = fminf(fmaxf((float)((double)wl + cl), -1.f), 1.f);
            const float vr = fminf(fmaxf((float)((double)wr_ + cr), -1.f), 1.f);
            const float yl = vl + x1 * dmix.x + x2 * dmix.y, yr = vr + x1 * dmix.z + x2 * dmix.w;
#ifdef MC_JACK_TRACE
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_out) : "v"(yl), "v"(yr));  // (the output is ready to be stored)
#endif
            if (!live) {
                asm volatile("" ::"v"(yl), "v"(yr));  // (the dry run: computed, not stored)
            } else if (A.out_gran) {  // the output as granules {value, sequence number}: on the host as soon as the posted writes land
                __hip_atomic_store(A.out_gran + m, ((unsigned long long)seq << 32) | __float_as_uint(yl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(A.out_gran + MC_B + m, ((unsigned long long)seq << 32) | __float_as_uint(yr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                outL[m] = yl;
                outR[m] = yr;
            }
        }
#ifdef MC_JACK_TRACE
        t_out = __builtin_amdgcn_s_memrealtime();  // the output's stores are issued
#endif
    }
    write_history(td, tau, tabs0, m, xin1, xin2, bp, rc);
    // ---- behind the output: what later periods need.  Wave 0 transforms the period for the delay line; the helper threads sum the
    // period's own term of the segment's second half: frame 256 + r takes taps j > r, which is the first half's triangle for the
    // reversed taps and the reversed period (frame 510 - m' of the segment is output m' of that convolution; frame 511 has no term)
    __syncthreads();  // (everyone has read the first half's partial sums)
    float4 xs_keep[4];
    if (wave == 0) {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = s_xp[TD_PAD + lane + 64 * r];
#pragma unroll
        for (int r = 4; r < 8; r++) v[r] = make_float2(0.f, 0.f);
        fft512_wave<-1, false>(v, s_fft, s_tw, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = lane + 64 * j;
            const float2 za = s_fft[k], zb = s_fft[(FFT_N - k) & (FFT_N - 1)];
            float2 x1, x2;
            if (k == 0) {
                const float2 zn = s_fft[MC_B];
                x1 = make_float2(za.x, zn.x);
                x2 = make_float2(za.y, zn.y);
            } else {
                x1 = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y));
                x2 = make_float2(0.5f * (za.y + zb.y), -0.5f * (za.x - zb.x));
            }
            xs_keep[j] = make_float4(x1.x, x1.y, x2.x, x2.y);
        }
    }
    __syncthreads();
    float seg_hi[2];
    {
        v2f sum = v2f{0.f, 0.f};
        const int mr = MC_B - 2 - m;  // output of the reversed convolution that is frame 256 + m (m = 255: none)
        const int nu = mr >= 0 ? (8 * (mr >> 3) + 8 + TD_NT2 - 1) / TD_NT2 : 0;  // (the slots behind them still hold the first half's sums)
#pragma unroll
        for (int c = 0; c < (MC_B + TD_NT2 - 1) / TD_NT2; c++) {
            const v2f v = vx_of(s_pc[c][mr < 0 ? 0 : mr]);
            const float keep = c < nu ? 1.f : 0.f;
            sum += v * keep;
        }
        int mh = m;
        asm volatile("" : "+v"(mh));  // (the address is formed here: kept from in front of the loop it costs the register allocator a spill)
        const float2 pre_hi = s_hi[mh];
        seg_hi[0] = pre_hi.x + sum.x;
        seg_hi[1] = pre_hi.y + sum.y;
    }
    {
        float* cur = seg + (size_t)seg0 * 2 * FFT_N;
        cur[m] = seg_lo[0];
        cur[MC_B + m] = seg_hi[0];
        cur[FFT_N + m] = seg_lo[1];
        cur[FFT_N + MC_B + m] = seg_hi[1];
        wet[(size_t)(tau & (wr - 1))] = own_wet[0];
        wet[(size_t)wr + (tau & (wr - 1))] = own_wet[1];
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = lane + 64 * j;
                fdl[(size_t)k * ring + slot0] = xs_keep[j];
                if (fdl16) fdl16[(size_t)k * ring + slot0] = pack_half4(xs_keep[j], FDL16_SCALE);
            }
            if (lane < MC_MAXV)
                slotgain[(size_t)lane * ring + slot0] = make_float4(bp.g[lane][0], bp.g[lane][1], bp.g[lane][2], bp.g[lane][3]);
        }
        if (m == 0) {
            double* o = cring + (size_t)(tabs0 & (rc - 1)) * 4;
            for (int c = 0; c < 4; c++) o[c] = cprev[c] + cnow[c];
            if (A.formcount) A.formcount[1] += 1;
        }
    }
    // publish completion to the host (mapped pinned memory): all waves drain their stores at the barrier
    // (__syncthreads waits vmcnt(0)), then ONE lane issues the system-scope release and the sequence number.
    // With tagged output the host does not look at the word (the granules are the completion): no release.
#ifndef MC_JACK_TRACE
    if (A.out_gran) return;
#endif
    __syncthreads();
    if (tid == 0) {
#ifdef MC_JACK_TRACE
        reinterpret_cast<unsigned long long*>(done_flag)[1] = t_start;
        reinterpret_cast<unsigned long long*>(done_flag)[2] = __builtin_amdgcn_s_memrealtime();
        reinterpret_cast<unsigned long long*>(done_flag)[3] = t_out;
        reinterpret_cast<unsigned long long*>(done_flag)[4] = c_out - c_start;
        reinterpret_cast<unsigned long long*>(done_flag)[5] = c_b1 - c_start;
        reinterpret_cast<unsigned long long*>(done_flag)[6] = c_fma - c_start;
        reinterpret_cast<unsigned long long*>(done_flag)[7] = c_b2 - c_start;
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#if defined(MCCONV_LAB) && defined(MC_TAIL_FFT0)  // the round-3 form of the tail (measurement build)
__device__ __forceinline__ void tail1_body_fft0(const TailArgs& A) {
    const float* in1 = A.in1;
    const float* in2 = A.in2;
    const VoiceSet& vset = A.vset;
    const int pstride_ir = A.pstride_ir;
    float4* __restrict__ fdl = A.fdl;
    float4* __restrict__ slotgain = A.slotgain;
    const int ring = A.ring, slot0 = A.slot0;
    const float4* __restrict__ part = A.part;
    const int nsum = A.nsum;
    const BlockParams* __restrict__ ptab = A.ptab;
    float* __restrict__ seg = A.seg;
    const int sr = A.sr, seg0 = A.seg0;
    float* __restrict__ wet = A.wet;
    const int wr = A.wr;
    double* __restrict__ cring = A.cring;
    const int rc = A.rc;
    const VoiceSums& vs = A.vs;
    const double inv_n = A.inv_n;
    const int compat = A.compat;
    const int64_t tabs0 = A.tabs0, predelay = A.predelay, n_ref = A.n_ref;
    float* __restrict__ outL = A.outL;
    float* __restrict__ outR = A.outR;
    const float2* __restrict__ g_tw = A.g_tw;
    const TailDrop& td = A.td;
    uint2* __restrict__ fdl16 = A.fdl16;
    unsigned* __restrict__ done_flag = A.done_flag;
    const unsigned seq = A.seq;
    const Retired& ret = A.ret;
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[FFT_WAVE_LDS];
    __shared__ float4 s_x[MC_NB];  // raw spectra of the new block {X1, X2}
    __shared__ float4 s_y[MC_NB];  // Y_L, Y_R
    __shared__ float s_wet[2][MC_B];
    __shared__ float s_in[2][MC_B];
    __shared__ double s_c[4];
    __shared__ float4 s_sa;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockParams& bp = ptab[0];  // read through global memory: a local copy indexed at run time would live in scratch
    const float4 dmix = make_float4(bp.d[0], bp.d[1], bp.d[2], bp.d[3]);  // (read here: at its use the load's round trip would sit between the period and its output)
    const int m = tid;
    const int64_t tau0 = tabs0 * MC_B, tau = tau0 + m, u = tau - predelay;

    // ---- every load whose address is known now is issued here, in one round of memory latency: the period
    // (PCIe), twiddles, this bin's chunk partials and partition-0 spectra, the previous tail, the delayed wet
    // samples, the Q1/Q2 prefix entries and the retired-epoch residuals.  (The kernel is one workgroup on the
    // critical path of a JACK period: six dependent round trips cost more than everything it computes.)
    const float2 tw0 = g_tw[tid], tw1 = g_tw[tid + 256];
    float4 ysum = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const float4* src = part + (size_t)tid * nsum;
        for (int c = 0; c < nsum; c++) {
            const float4 a = src[c];
            ysum.x += a.x;
            ysum.y += a.y;
            ysum.z += a.z;
            ysum.w += a.w;
        }
    }
    float4 h0v[MC_MAXV], h1v[MC_MAXV];
    // partition 1 pairs with the previous block: its spectrum and its slot's gains come from the delay line.  The
    // sweep over partitions >= 2 of THIS block needed nothing of the previous period, so it ran beside that period's
    // tail on a second stream (mcconv.hip, process_one).
    float4 h0w[MC_MAXV], h1w[MC_MAXV], g1w[MC_MAXV];
    const int slot1 = (slot0 + ring - 1) & (ring - 1);
    const float4 xprev = fdl[(size_t)tid * ring + slot1];
#pragma unroll
    for (int vi = 0; vi < MC_MAXV; vi++) {  // constant indices: runtime-indexed kernel-argument arrays go to scratch
        h0v[vi] = h1v[vi] = h0w[vi] = h1w[vi] = g1w[vi] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vi < vset.n) {
            h0v[vi] = vset.H0[vi][(size_t)tid * pstride_ir];
            h1v[vi] = vset.H1[vi][(size_t)tid * pstride_ir];
            h0w[vi] = vset.H0[vi][(size_t)tid * pstride_ir + 1];
            h1w[vi] = vset.H1[vi][(size_t)tid * pstride_ir + 1];
            g1w[vi] = slotgain[(size_t)vset.vid[vi] * ring + slot1];
        }
    }
    const float* prv = seg + (size_t)((seg0 + sr - 1) & (sr - 1)) * 2 * FFT_N;
    const float prvL = prv[MC_B + m], prvR = prv[FFT_N + MC_B + m];
    float dwl = 0.f, dwr = 0.f;  // delayed wet sample when it predates this period
    if (u >= 0 && u < tau0) {
        dwl = wet[(size_t)(u & (wr - 1))];
        dwr = wet[(size_t)wr + (u & (wr - 1))];
    }
    float2 ra = make_float2(0.f, 0.f), rb = ra;
    if (tau < ret.end) {  // what blocks played under an earlier predelay still owe
        ra = retired_at(ret.mac, ret.rr, tau);
        rb = retired_at(ret.fix, ret.rr, tau);
    }
    const int64_t thi = u >> 8;
    int64_t tlo = tau - n_ref >= 0 ? ((tau - n_ref) >> 8) : -1;
    if (tlo < ret.b0 - 1) tlo = ret.b0 - 1;
    const bool corr_on = compat && u >= 0 && thi > tlo;
    double ca[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0}, cprev[4] = {0, 0, 0, 0};
    if (corr_on && thi != tabs0) {
        const double* pa = cring + (size_t)(thi & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) ca[c] = pa[c];
    }
    if (corr_on && tlo >= 0) {
        const double* pb = cring + (size_t)(tlo & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) cb[c] = pb[c];
    }
    if (tid == 64 && tabs0 > 0) {  // the thread that will extend the prefix sums
        const double* pp = cring + (size_t)((tabs0 + rc - 1) & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) cprev[c] = pp[c];
    }

    s_tw[tid] = tw0;
    s_tw[tid + 256] = tw1;
    // Q8: what the reference's cut at n_ref takes away from this period's samples comes from blocks at least n_ref frames
    // old - nothing of it depends on the period itself: k_drop_period, launched ahead of this kernel, has summed it
    float td_l[1] = {0.f}, td_r[1] = {0.f};
    if (td.on) {
        td_l[0] = A.drop[m];
        td_r[0] = A.drop[MC_B + m];
    }
    float xin1 = 0.f, xin2 = 0.f;
    bool have_in = false;
    if (A.bell && A.in_gran) {
        // Parked, tagged input: every lane polls its own two granules of the period; wave 0 also watches the doorbell word for
        // the "give up" command and the park time.  A wave leaves the loop when all its lanes hold this period's samples, or
        // when wave 0 has said to leave (an LDS word); the barrier behind the loop makes the decision the workgroup's.
        __shared__ int s_abort;
        if (tid == 0) s_abort = 0;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned long long g1 = __hip_atomic_load(A.in_gran + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long g2 = __hip_atomic_load(A.in_gran + MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (__all((unsigned)(g1 >> 32) == seq && (unsigned)(g2 >> 32) == seq)) {
                xin1 = __uint_as_float((unsigned)g1);
                xin2 = __uint_as_float((unsigned)g2);
                break;
            }
            if (tid == 0) {
                const unsigned long long v = __hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)v == seq && (v >> 32) != 0) {
                    *(volatile int*)&s_abort = 1;  // told to give up
                } else if (__builtin_amdgcn_s_memrealtime() - t0 > A.park_ticks) {
                    __hip_atomic_store(A.exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    *(volatile int*)&s_abort = 1;  // the host has been away for longer than the park time
                }
            }
            if (*(volatile int*)&s_abort) break;
            __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        if (*(volatile int*)&s_abort) return;  // nothing has been written: the host launches this period again
        have_in = true;
    } else if (A.bell) {
        // Parked: everything above was requested without the period; only its 2 KB are still missing.  One lane
        // polls the mapped doorbell (a PCIe read per poll), the others wait at the barrier.
        __shared__ int s_go;
        if (tid == 0) {
            int go = 1;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned long long v = __hip_atomic_load(A.bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)v == seq) {
                    go = (v >> 32) == 0;
                    break;
                }
                if (__builtin_amdgcn_s_memrealtime() - t0 > A.park_ticks) {
                    go = 0;
                    __hip_atomic_store(A.exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            s_go = go;
        }
        __syncthreads();
        if (!s_go) return;  // nothing has been written: the host launches this period again
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the period was written before the doorbell
    }
#ifdef MC_JACK_TRACE  // diagnostic build: when the period's work started and ended (100 MHz), next to the completion word
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c_start = __builtin_amdgcn_s_memtime();
#endif
    // (system scope: the period may sit in device memory the CPU wrote through the BAR - not to be served from a cache)
    if (!have_in) {
        xin1 = __hip_atomic_load(in1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        xin2 = __hip_atomic_load(in2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    s_in[0][tid] = xin1;
    s_in[1][tid] = xin2;
    __syncthreads();
    float4 xs_keep[4];  // wave 0: the block's spectra, stored to the delay line at the end of the kernel
    if (wave == 0) {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = make_float2(s_in[0][lane + 64 * r], s_in[1][lane + 64 * r]);
#pragma unroll
        for (int r = 4; r < 8; r++) v[r] = make_float2(0.f, 0.f);
        fft512_wave<-1, false>(v, s_fft, s_tw, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = lane + 64 * j;
            const float2 za = s_fft[k], zb = s_fft[(FFT_N - k) & (FFT_N - 1)];
            float2 x1, x2;
            if (k == 0) {
                const float2 zn = s_fft[MC_B];
                x1 = make_float2(za.x, zn.x);
                x2 = make_float2(za.y, zn.y);
                s_sa = make_float4(za.x, za.y, zn.x, zn.y);
            } else {
                x1 = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y));
                x2 = make_float2(0.5f * (za.y + zb.y), -0.5f * (za.x - zb.x));
            }
            const float4 xs = make_float4(x1.x, x1.y, x2.x, x2.y);
            s_x[k] = xs;
            xs_keep[j] = xs;
        }
    }
    __syncthreads();
    {  // bin tid: chunk partials (partitions >= 1) + partition 0 of every voice's IRs against the new block
        const int k = tid;
        const float4 x = s_x[k];
        float4 y = ysum;
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++) {
            if (vi >= vset.n) break;
            const float* g = ptab->g[vset.vid[vi]];
            const float4 h0 = h0v[vi], h1 = h1v[vi];
            float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            if (k == 0) {
                cmac<true>(a0, h0.x, h0.y, x.x, x.y);
                cmac<true>(a1, h1.x, h1.y, x.z, x.w);
                cmac<true>(a2, h0.z, h0.w, x.x, x.y);
                cmac<true>(a3, h1.z, h1.w, x.z, x.w);
            } else {
                cmac<false>(a0, h0.x, h0.y, x.x, x.y);
                cmac<false>(a1, h1.x, h1.y, x.z, x.w);
                cmac<false>(a2, h0.z, h0.w, x.x, x.y);
                cmac<false>(a3, h1.z, h1.w, x.z, x.w);
            }
            y.x += g[0] * a0.x + g[1] * a1.x;
            y.y += g[0] * a0.y + g[1] * a1.y;
            y.z += g[2] * a2.x + g[3] * a3.x;
            y.w += g[2] * a2.y + g[3] * a3.y;
            // partition 1 against the previous block, with the gains that block carries
            const float4 p0 = h0w[vi], p1 = h1w[vi], gq = g1w[vi];
            a0 = a1 = a2 = a3 = make_float2(0.f, 0.f);
            if (k == 0) {
                cmac<true>(a0, p0.x, p0.y, xprev.x, xprev.y);
                cmac<true>(a1, p1.x, p1.y, xprev.z, xprev.w);
                cmac<true>(a2, p0.z, p0.w, xprev.x, xprev.y);
                cmac<true>(a3, p1.z, p1.w, xprev.z, xprev.w);
            } else {
                cmac<false>(a0, p0.x, p0.y, xprev.x, xprev.y);
                cmac<false>(a1, p1.x, p1.y, xprev.z, xprev.w);
                cmac<false>(a2, p0.z, p0.w, xprev.x, xprev.y);
                cmac<false>(a3, p1.z, p1.w, xprev.z, xprev.w);
            }
            y.x += gq.x * a0.x + gq.y * a1.x;
            y.y += gq.x * a0.y + gq.y * a1.y;
            y.z += gq.z * a2.x + gq.w * a3.x;
            y.w += gq.z * a2.y + gq.w * a3.y;
        }
        s_y[k] = y;
    }
    __syncthreads();
    if (wave == 0) {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = lane + 64 * r;
            float2 w;
            if (n == 0) {
                const float4 y = s_y[0];
                w = make_float2(y.x, y.z);
            } else if (n == MC_B) {
                const float4 y = s_y[0];
                w = make_float2(y.y, y.w);
            } else if (n < MC_B) {
                const float4 y = s_y[n];
                w = make_float2(y.x - y.w, y.y + y.z);
            } else {
                const float4 y = s_y[FFT_N - n];
                w = make_float2(y.x + y.w, -y.y + y.z);
            }
            v[r] = w;
        }
        fft512_wave<+1, false>(v, s_fft, s_tw, lane);
    } else if (tid == 64) {
        // meanwhile: this block's Q1/Q2 terms and the new prefix entry (float64, serial)
        double d[4] = {0, 0, 0, 0};
        if (compat) corr_terms(s_sa, bp, vs, inv_n, d);
        for (int c = 0; c < 4; c++) s_c[c] = cprev[c] + d[c];
    }
    __syncthreads();
    // No global store is issued before the output has left: a barrier drains the vector-memory counter, so every
    // store ahead of it would put its acknowledgement latency on the critical path.
    float seg_lo[2], seg_hi[2], own_wet[2];
    {
        // overlap-add with the previous block's tail; this block's segments go to the ring
        const float sc = 1.0f / FFT_N;
        const float2 lo = s_fft[m], hi = s_fft[MC_B + m];
        seg_lo[0] = lo.x * sc;
        seg_lo[1] = lo.y * sc;
        seg_hi[0] = hi.x * sc;
        seg_hi[1] = hi.y * sc;
        own_wet[0] = seg_lo[0] + prvL;
        own_wet[1] = seg_lo[1] + prvR;
        s_wet[0][m] = own_wet[0];
        s_wet[1][m] = own_wet[1];
    }
    __syncthreads();
    {
        float wl = dwl, wr_ = dwr;
        if (u >= tau0) {  // inside this block: not yet visible through global memory
            wl = s_wet[0][u - tau0];
            wr_ = s_wet[1][u - tau0];
        }
        wl += ra.x + rb.x;
        wr_ += ra.y + rb.y;
        double cl = 0.0, cr = 0.0;
        if (corr_on) {
            if (thi == tabs0)
                for (int c = 0; c < 4; c++) ca[c] = s_c[c];
            const double sg = (u & 1) ? -1.0 : 1.0;
            cl = (ca[0] - cb[0]) + sg * (ca[2] - cb[2]);
            cr = (ca[1] - cb[1]) + sg * (ca[3] - cb[3]);
        }
        const float x1 = xin1, x2 = xin2;
        if (td.on) {
            wl -= td_l[0];
            wr_ -= td_r[0];
        }
        const float vl = fminf(fmaxf((float)((double)wl + cl), -1.f), 1.f);
        const float vr = fminf(fmaxf((float)((double)wr_ + cr), -1.f), 1.f);
        const float yl = vl + x1 * dmix.x + x2 * dmix.y, yr = vr + x1 * dmix.z + x2 * dmix.w;
        if (A.out_gran) {  // the output as granules {value, sequence number}: on the host as soon as the posted writes land
            __hip_atomic_store(A.out_gran + m, ((unsigned long long)seq << 32) | __float_as_uint(yl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(A.out_gran + MC_B + m, ((unsigned long long)seq << 32) | __float_as_uint(yr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            outL[m] = yl;
            outR[m] = yr;
        }
        write_history(td, tau, tabs0, m, x1, x2, bp, rc);
    }
#ifdef MC_JACK_TRACE
    const unsigned long long t_out = __builtin_amdgcn_s_memrealtime();  // the output's stores are issued
    const unsigned long long c_out = __builtin_amdgcn_s_memtime();      // (shader clocks over the same stretch)
#endif
    // the state later periods need: delay-line slot, slot gains, segments, wet ring, Q1/Q2 prefix entry
    {
        float* cur = seg + (size_t)seg0 * 2 * FFT_N;
        cur[m] = seg_lo[0];
        cur[MC_B + m] = seg_hi[0];
        cur[FFT_N + m] = seg_lo[1];
        cur[FFT_N + MC_B + m] = seg_hi[1];
        wet[(size_t)(tau & (wr - 1))] = own_wet[0];
        wet[(size_t)wr + (tau & (wr - 1))] = own_wet[1];
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = lane + 64 * j;
                fdl[(size_t)k * ring + slot0] = xs_keep[j];
                if (fdl16) fdl16[(size_t)k * ring + slot0] = pack_half4(xs_keep[j], FDL16_SCALE);
            }
            if (lane < MC_MAXV)
                slotgain[(size_t)lane * ring + slot0] = make_float4(bp.g[lane][0], bp.g[lane][1], bp.g[lane][2], bp.g[lane][3]);
        }
        if (m == 0) {
            double* o = cring + (size_t)(tabs0 & (rc - 1)) * 4;
            for (int c = 0; c < 4; c++) o[c] = s_c[c];
        }
    }
    // publish completion to the host (mapped pinned memory): all waves drain their stores at the barrier
    // (__syncthreads waits vmcnt(0)), then ONE lane issues the system-scope release and the sequence number.
    // With tagged output the host does not look at the word (the granules are the completion): no release, the kernel ends
    // 0.6 us earlier and the next period's kernel starts that much sooner (back-to-back calls).
#ifndef MC_JACK_TRACE
    if (A.out_gran) return;
#endif
    __syncthreads();
    if (tid == 0) {
#ifdef MC_JACK_TRACE
        reinterpret_cast<unsigned long long*>(done_flag)[1] = t_start;
        reinterpret_cast<unsigned long long*>(done_flag)[2] = __builtin_amdgcn_s_memrealtime();
        reinterpret_cast<unsigned long long*>(done_flag)[3] = t_out;
        reinterpret_cast<unsigned long long*>(done_flag)[4] = c_out - c_start;
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#define tail1_body tail1_body_fft0
#endif

#if defined(MCCONV_LAB) && defined(MC_TAIL_FFT0)
#undef TAIL1_THREADS
#define TAIL1_THREADS 256
#endif
__global__ __launch_bounds__(TAIL1_THREADS) void k_tail1(TailArgs A) { tail1_body(A); }

// The tail of period t (workgroup 0, usually parked on its doorbell) and the streaming sweep over partitions >= 2 of
// period t + 1 (the other 256 bins x chunks workgroups, one voice) in ONE launch: the sweep pairs only with blocks at
// least two periods old, so it needs nothing the tail produces and runs while the tail waits for the period.
struct SweepArgs {
    const void *H0, *H1;
    int pstride_ir, p_begin, p_end, chunk;
    const void* fdl;
    const float4* slotgain;
    int ring, slot0;
    float4* part;
    int nsum, ch_off;
    float4 ugain;
    float2 inv;
    int nchunk;
    float* drop_next;  // Q8 regime: != null: the launch's LAST workgroup sums the cut terms of the period after the tail's (block A.tabs0 + 1, the
                       // tail's own TailDrop, predelay and epoch) into this buffer - like the sweep a speculation the next call checks: they
                       // depend on blocks at least n_ref frames old, so nothing of them waits for a period (k_drop_period_fft as a launch
                       // of its own cost the host a launch per period and, back to back, 5 us of stream time between two tails)
};
template <bool UNIFORM>
__global__ __launch_bounds__(TAIL1_THREADS) void k_jack(TailArgs A, SweepArgs S) {
    if (blockIdx.x == 0) {
        tail1_body(A);
    } else if (S.drop_next && blockIdx.x == gridDim.x - 1) {
        drop_period_fft_body<1>(A.td, S.drop_next, A.tabs0 + 1, A.predelay, A.n_ref, A.ret.b0);
    } else {
        const int w = (int)blockIdx.x - 1;
        mac_stream_body<UNIFORM, TAIL1_THREADS, false>(w & (MC_NB - 1), w >> 8, 0, S.H0, S.H1, S.pstride_ir, S.p_begin, S.p_end, S.chunk, S.fdl,
                                             S.slotgain, S.ring, S.slot0, S.part, S.nsum, S.ch_off, S.ugain, S.inv);
    }
}

// ---------------------------------------------------------------------------
// JACK periods of 512 / 1024 frames (PM = 2 / 4 blocks per call): the same
// idea as k_tail1 for PM blocks at once.  k_mac_stream has summed partitions
// p >= PM for the PM blocks of the call in the shadow of the previous period
// (they pair only with blocks already in the delay line); this kernel does the
// rest in one workgroup: wave w transforms block w, every thread adds the
// PM x PM low-partition products of its bin (new blocks from LDS, the PM - 1
// newest old blocks from the delay line), wave w inverts block w, then
// overlap-add, Q1/Q2 prefix, predelay, Q8, clamp, dry for PM x 256 frames.
// The blocks of a call share one parameter entry (the reference advances its
// cross-fade once per call) and the Q1/Q2/Q8 windows start at the call.
// ---------------------------------------------------------------------------
// SELF_DROP (a parked launch in the Q8 regime): the workgroup sums the period's own cut terms first, while the period has not arrived
// (they depend on blocks at least n_ref frames old) - no launch of their own ahead of this one.
template <int PM, bool SELF_DROP>
#ifdef TAILP_VGPR_CAP  // (measurement build: DESIGN section 4, JACK path, item 3)
__attribute__((amdgpu_num_vgpr(TAILP_VGPR_CAP)))
#endif
__global__ __launch_bounds__(256) void k_tailp(const float* __restrict__ in1, const float* __restrict__ in2, VoiceSet vset,
                                               int pstride_ir, float4* __restrict__ fdl, float4* __restrict__ slotgain, int ring,
                                               int slot0, const float4* __restrict__ part, int nsum,
                                               const BlockParams* __restrict__ ptab, float* __restrict__ seg, int sr,
                                               float* __restrict__ wet, int wr, double* __restrict__ cring, int rc, VoiceSums vs,
                                               double inv_n, int compat, int64_t tabs0, int64_t predelay, int64_t n_ref,
                                               float* __restrict__ outL, float* __restrict__ outR,
                                               const float2* __restrict__ g_tw, TailDrop td, uint2* __restrict__ fdl16,
                                               unsigned* __restrict__ done_flag, unsigned seq, Retired ret,
                                               const unsigned long long* bell, unsigned* exited, unsigned long long park_ticks, const float* __restrict__ drop,
                                               const unsigned long long* in_gran, unsigned long long* out_gran) {
    // bell != null: launched one call ahead, parks on its doorbell like the single-block tail (tail1_body)
    static_assert(PM == 2 || PM == 4, "one wave per block of the call");
    if (SELF_DROP) {
        drop_period_fft_body<PM>(td, const_cast<float*>(drop), tabs0, predelay, n_ref, ret.b0);
        __syncthreads();  // (the workgroup's own stores, read back below)
    }
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[PM][FFT_WAVE_LDS];
    __shared__ float4 s_xy[PM][MC_NB];  // spectra {X1, X2} of the new blocks, then {Y_L, Y_R}
    __shared__ float s_wet[2][PM * MC_B];
    __shared__ float s_in[2][PM * MC_B];
    __shared__ double s_d[PM][4];  // Q1/Q2 terms of the blocks, then their running prefix sums
    __shared__ float4 s_sa[PM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockParams& bp = ptab[0];
    const float4 dmix = make_float4(bp.d[0], bp.d[1], bp.d[2], bp.d[3]);  // (read here: at its use the load's round trip would sit between the period and its output)
    const int m = tid;
    const int64_t tau0 = tabs0 * MC_B;

    // ---- loads with addresses known at entry (one round of memory latency; see k_tail1) ----
    float xin[PM][2];
    if (!bell) {
#pragma unroll
        for (int j = 0; j < PM; j++) {
            // (system scope: the period may sit in device memory the CPU wrote through the BAR)
            xin[j][0] = __hip_atomic_load(in1 + j * MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            xin[j][1] = __hip_atomic_load(in2 + j * MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const float2 tw0 = g_tw[tid], tw1 = g_tw[tid + 256];
    float4 ysum[PM];
#pragma unroll
    for (int j = 0; j < PM; j++) {
        ysum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* src = part + ((size_t)j * MC_NB + tid) * nsum;
        for (int c = 0; c < nsum; c++) {
            const float4 a = src[c];
            ysum[j].x += a.x;
            ysum[j].y += a.y;
            ysum[j].z += a.z;
            ysum[j].w += a.w;
        }
    }
    float4 xold[PM];  // xold[q] = spectrum of the block q before the call, q = 1 .. PM-1
#pragma unroll
    for (int q = 1; q < PM; q++) xold[q] = fdl[(size_t)tid * ring + ((slot0 - q) & (ring - 1))];
    const float* prv = seg + (size_t)((tabs0 - 1) & (sr - 1)) * 2 * FFT_N;
    const float prvL = prv[MC_B + m], prvR = prv[FFT_N + MC_B + m];
    float dw[PM][2];
    float2 ra[PM], rb[PM];
#pragma unroll
    for (int j = 0; j < PM; j++) {
        const int64_t tau = tau0 + j * MC_B + m, u = tau - predelay;
        dw[j][0] = dw[j][1] = 0.f;
        if (u >= 0 && u < tau0) {
            dw[j][0] = wet[(size_t)(u & (wr - 1))];
            dw[j][1] = wet[(size_t)wr + (u & (wr - 1))];
        }
        ra[j] = rb[j] = make_float2(0.f, 0.f);
        if (tau < ret.end) {
            ra[j] = retired_at(ret.mac, ret.rr, tau);
            rb[j] = retired_at(ret.fix, ret.rr, tau);
        }
    }
    // 512-frame periods: the low partitions' spectra and the old blocks' slot gains are requested here, with everything else (behind the
    // forward transforms' barrier, where they are used, their round trip through memory would sit between the period and its output;
    // at 1024 frames the 176 registers they would hold across the wait are not there)
    constexpr bool EARLY_H = PM == 2;
    float4 h0e[EARLY_H ? MC_MAXV : 1][PM], h1e[EARLY_H ? MC_MAXV : 1][PM], golde[EARLY_H ? MC_MAXV : 1][PM];
    if constexpr (EARLY_H) {
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++) {
#pragma unroll
            for (int p = 0; p < PM; p++) h0e[vi][p] = h1e[vi][p] = golde[vi][p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vi < vset.n) {
#pragma unroll
                for (int p = 0; p < PM; p++) {
                    h0e[vi][p] = vset.H0[vi][(size_t)tid * pstride_ir + p];
                    h1e[vi][p] = vset.H1[vi][(size_t)tid * pstride_ir + p];
                }
#pragma unroll
                for (int q = 1; q < PM; q++) golde[vi][q] = slotgain[(size_t)vset.vid[vi] * ring + ((slot0 - q) & (ring - 1))];
            }
        }
    }
    double cprev[4] = {0, 0, 0, 0};
    if (tid == 0 && tabs0 > 0) {
        const double* pp = cring + (size_t)((tabs0 - 1) & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) cprev[c] = pp[c];
    }
    s_tw[tid] = tw0;
    s_tw[tid + 256] = tw1;
    // Q8 terms of the call's PM x 256 samples: from blocks at least n_ref frames old, summed by k_drop_period ahead of this kernel
    float td_l[PM], td_r[PM];
#pragma unroll
    for (int j = 0; j < PM; j++) {
        td_l[j] = td_r[j] = 0.f;
        if (td.on) {
            td_l[j] = drop[j * MC_B + m];
            td_r[j] = drop[PM * MC_B + j * MC_B + m];
        }
    }
    if (bell && in_gran) {
        // Parked, tagged input (see k_tail1): every lane polls its own 2 PM granules of the period, wave 0 watches the doorbell word
        // for the "give up" command and the park time
        __shared__ int s_abort;
        if (tid == 0) s_abort = 0;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < PM; j++) {
                const unsigned long long g1 = __hip_atomic_load(in_gran + j * MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long g2 = __hip_atomic_load(in_gran + (PM + j) * MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = ok && (unsigned)(g1 >> 32) == seq && (unsigned)(g2 >> 32) == seq;
                xin[j][0] = __uint_as_float((unsigned)g1);
                xin[j][1] = __uint_as_float((unsigned)g2);
            }
            if (__all(ok)) break;
            if (tid == 0) {
                const unsigned long long v = __hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)v == seq && (v >> 32) != 0) {
                    *(volatile int*)&s_abort = 1;
                } else if (__builtin_amdgcn_s_memrealtime() - t0 > park_ticks) {
                    __hip_atomic_store(exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    *(volatile int*)&s_abort = 1;
                }
            }
            if (*(volatile int*)&s_abort) break;
            __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        if (*(volatile int*)&s_abort) return;  // nothing has been written: the host launches this period again
    } else if (bell) {
        // Parked: everything above was requested without the period; only its samples are still missing.  One lane polls
        // the doorbell, the others wait at the barrier.
        __shared__ int s_go;
        if (tid == 0) {
            int go = 1;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned long long v = __hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)v == seq) {
                    go = (v >> 32) == 0;
                    break;
                }
                if (__builtin_amdgcn_s_memrealtime() - t0 > park_ticks) {
                    go = 0;
                    __hip_atomic_store(exited, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            s_go = go;
        }
        __syncthreads();
        if (!s_go) return;  // nothing has been written: the host launches this period again
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the period was written before the doorbell
#pragma unroll
        for (int j = 0; j < PM; j++) {
            xin[j][0] = __hip_atomic_load(in1 + j * MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            xin[j][1] = __hip_atomic_load(in2 + j * MC_B + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#pragma unroll
    for (int j = 0; j < PM; j++) {
        s_in[0][j * MC_B + tid] = xin[j][0];
        s_in[1][j * MC_B + tid] = xin[j][1];
    }
    __syncthreads();

    // ---- wave w: forward transform of block w ----
    float4 xs_keep[4];
    if (wave < PM) {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = make_float2(s_in[0][wave * MC_B + lane + 64 * r], s_in[1][wave * MC_B + lane + 64 * r]);
#pragma unroll
        for (int r = 4; r < 8; r++) v[r] = make_float2(0.f, 0.f);
        float2* lds = s_fft[wave];
        fft512_wave<-1, false>(v, lds, s_tw, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = lane + 64 * j;
            const float2 za = lds[k], zb = lds[(FFT_N - k) & (FFT_N - 1)];
            float2 x1, x2;
            if (k == 0) {
                const float2 zn = lds[MC_B];
                x1 = make_float2(za.x, zn.x);
                x2 = make_float2(za.y, zn.y);
                s_sa[wave] = make_float4(za.x, za.y, zn.x, zn.y);
            } else {
                x1 = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y));
                x2 = make_float2(0.5f * (za.y + zb.y), -0.5f * (za.x - zb.x));
            }
            const float4 xs = make_float4(x1.x, x1.y, x2.x, x2.y);
            s_xy[wave][k] = xs;
            xs_keep[j] = xs;
        }
    }
    __syncthreads();

    // ---- bin tid: partitions p < PM of every voice against blocks j - p (new: LDS, old: delay line) ----
    {
        const int k = tid;
        float4 xnew[PM], y[PM];
#pragma unroll
        for (int j = 0; j < PM; j++) {
            xnew[j] = s_xy[j][k];
            y[j] = ysum[j];
        }
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++) {
            if (vi >= vset.n) break;
            const int row = vset.vid[vi];
            const float* gn = ptab->g[row];
            float4 h0[PM], h1[PM], gold[PM];
            if constexpr (EARLY_H) {
#pragma unroll
                for (int p = 0; p < PM; p++) h0[p] = h0e[vi][p], h1[p] = h1e[vi][p], gold[p] = golde[vi][p];
            } else {
#pragma unroll
                for (int p = 0; p < PM; p++) {
                    h0[p] = vset.H0[vi][(size_t)k * pstride_ir + p];
                    h1[p] = vset.H1[vi][(size_t)k * pstride_ir + p];
                }
#pragma unroll
                for (int q = 1; q < PM; q++) gold[q] = slotgain[(size_t)row * ring + ((slot0 - q) & (ring - 1))];
            }
#pragma unroll
            for (int j = 0; j < PM; j++) {
#pragma unroll
                for (int p = 0; p < PM; p++) {
                    const float4 x = (p <= j) ? xnew[(p <= j) ? j - p : 0] : xold[(p > j) ? p - j : 1];
                    const float4 g = (p <= j) ? make_float4(gn[0], gn[1], gn[2], gn[3]) : gold[(p > j) ? p - j : 1];
                    float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
                    if (k == 0) {
                        cmac<true>(a0, h0[p].x, h0[p].y, x.x, x.y);
                        cmac<true>(a1, h1[p].x, h1[p].y, x.z, x.w);
                        cmac<true>(a2, h0[p].z, h0[p].w, x.x, x.y);
                        cmac<true>(a3, h1[p].z, h1[p].w, x.z, x.w);
                    } else {
                        cmac<false>(a0, h0[p].x, h0[p].y, x.x, x.y);
                        cmac<false>(a1, h1[p].x, h1[p].y, x.z, x.w);
                        cmac<false>(a2, h0[p].z, h0[p].w, x.x, x.y);
                        cmac<false>(a3, h1[p].z, h1[p].w, x.z, x.w);
                    }
                    y[j].x += g.x * a0.x + g.y * a1.x;
                    y[j].y += g.x * a0.y + g.y * a1.y;
                    y[j].z += g.z * a2.x + g.w * a3.x;
                    y[j].w += g.z * a2.y + g.w * a3.y;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PM; j++) s_xy[j][k] = y[j];
    }
    __syncthreads();

    // ---- wave w: inverse transform of block w, then (lane 0) the block's Q1/Q2 terms ----
    if (wave < PM) {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = lane + 64 * r;
            float2 w;
            if (n == 0) {
                const float4 y = s_xy[wave][0];
                w = make_float2(y.x, y.z);
            } else if (n == MC_B) {
                const float4 y = s_xy[wave][0];
                w = make_float2(y.y, y.w);
            } else if (n < MC_B) {
                const float4 y = s_xy[wave][n];
                w = make_float2(y.x - y.w, y.y + y.z);
            } else {
                const float4 y = s_xy[wave][FFT_N - n];
                w = make_float2(y.x + y.w, -y.y + y.z);
            }
            v[r] = w;
        }
        fft512_wave<+1, false>(v, s_fft[wave], s_tw, lane);
        if (lane == 0) {
            double d[4] = {0, 0, 0, 0};
            if (compat) corr_terms(s_sa[wave], bp, vs, inv_n, d);
            for (int c = 0; c < 4; c++) s_d[wave][c] = d[c];
        }
    }
    __syncthreads();

    // ---- overlap-add chain through the call; running Q1/Q2 prefix ----
    const float sc = 1.0f / FFT_N;
    float seg_lo[PM][2], seg_hi[PM][2], own[PM][2];
#pragma unroll
    for (int j = 0; j < PM; j++) {
        const float2 lo = s_fft[j][m], hi = s_fft[j][MC_B + m];
        seg_lo[j][0] = lo.x * sc;
        seg_lo[j][1] = lo.y * sc;
        seg_hi[j][0] = hi.x * sc;
        seg_hi[j][1] = hi.y * sc;
        own[j][0] = seg_lo[j][0] + (j == 0 ? prvL : seg_hi[j > 0 ? j - 1 : 0][0]);
        own[j][1] = seg_lo[j][1] + (j == 0 ? prvR : seg_hi[j > 0 ? j - 1 : 0][1]);
        s_wet[0][j * MC_B + m] = own[j][0];
        s_wet[1][j * MC_B + m] = own[j][1];
    }
    if (tid == 0) {
        double run[4] = {cprev[0], cprev[1], cprev[2], cprev[3]};
#pragma unroll
        for (int j = 0; j < PM; j++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                run[c] += s_d[j][c];
                s_d[j][c] = run[c];
            }
        }
    }
    __syncthreads();

    // ---- predelay, residuals, Q1/Q2 windows (from the call start), Q8, clamp, dry ----
#pragma unroll
    for (int j = 0; j < PM; j++) {
        const int64_t tau = tau0 + j * MC_B + m, u = tau - predelay;
        float wl = dw[j][0], wr_ = dw[j][1];
        if (u >= tau0) {
            wl = s_wet[0][u - tau0];
            wr_ = s_wet[1][u - tau0];
        }
        wl += ra[j].x + rb[j].x;
        wr_ += ra[j].y + rb[j].y;
        double cl = 0.0, cr = 0.0;
        if (compat && u >= 0) {
            const int64_t thi = ((u >> 8) / PM + 1) * PM - 1;  // last block of the call that holds sample u
            const int64_t v = tau - n_ref;
            int64_t tlo = v >= 0 ? ((v >> 8) / PM + 1) * PM - 1 : -1;
            if (tlo < ret.b0 - 1) tlo = ret.b0 - 1;
            if (thi > tlo) {
                double a[4], b[4] = {0, 0, 0, 0};
                if (thi >= tabs0) {
                    for (int c = 0; c < 4; c++) a[c] = s_d[thi - tabs0][c];
                } else {
                    const double* pa = cring + (size_t)(thi & (rc - 1)) * 4;
                    for (int c = 0; c < 4; c++) a[c] = pa[c];
                }
                if (tlo >= 0) {
                    const double* pb = cring + (size_t)(tlo & (rc - 1)) * 4;
                    for (int c = 0; c < 4; c++) b[c] = pb[c];
                }
                const double sg = (u & 1) ? -1.0 : 1.0;
                cl = (a[0] - b[0]) + sg * (a[2] - b[2]);
                cr = (a[1] - b[1]) + sg * (a[3] - b[3]);
            }
        }
        const float x1 = xin[j][0], x2 = xin[j][1];
        if (td.on) {
            wl -= td_l[j];
            wr_ -= td_r[j];
        }
        const float vl = fminf(fmaxf((float)((double)wl + cl), -1.f), 1.f);
        const float vr = fminf(fmaxf((float)((double)wr_ + cr), -1.f), 1.f);
        const float yl = vl + x1 * dmix.x + x2 * dmix.y, yr = vr + x1 * dmix.z + x2 * dmix.w;
        if (out_gran) {  // granules {value, sequence number}: on the host as soon as the posted writes land (see k_tail1)
            __hip_atomic_store(out_gran + j * MC_B + m, ((unsigned long long)seq << 32) | __float_as_uint(yl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(out_gran + (PM + j) * MC_B + m, ((unsigned long long)seq << 32) | __float_as_uint(yr), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            outL[j * MC_B + m] = yl;
            outR[j * MC_B + m] = yr;
        }
    }
    // ---- state for later periods (no global store before this point: see k_tail1) ----
#pragma unroll
    for (int j = 0; j < PM; j++) {
        const int64_t tau = tau0 + j * MC_B + m;
        float* cur = seg + (size_t)((tabs0 + j) & (sr - 1)) * 2 * FFT_N;
        cur[m] = seg_lo[j][0];
        cur[MC_B + m] = seg_hi[j][0];
        cur[FFT_N + m] = seg_lo[j][1];
        cur[FFT_N + MC_B + m] = seg_hi[j][1];
        wet[(size_t)(tau & (wr - 1))] = own[j][0];
        wet[(size_t)wr + (tau & (wr - 1))] = own[j][1];
        write_history(td, tau, tabs0 + j, m, xin[j][0], xin[j][1], bp, rc);
    }
    if (wave < PM) {
        const int slot = (slot0 + wave) & (ring - 1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = lane + 64 * j;
            fdl[(size_t)k * ring + slot] = xs_keep[j];
            if (fdl16) fdl16[(size_t)k * ring + slot] = pack_half4(xs_keep[j], FDL16_SCALE);
        }
        if (lane < MC_MAXV)
            slotgain[(size_t)lane * ring + slot] = make_float4(bp.g[lane][0], bp.g[lane][1], bp.g[lane][2], bp.g[lane][3]);
    }
    if (tid < PM) {
        double* o = cring + (size_t)((tabs0 + tid) & (rc - 1)) * 4;
        for (int c = 0; c < 4; c++) o[c] = s_d[tid][c];
    }
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
