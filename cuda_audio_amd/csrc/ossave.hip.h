// ossave.hip.h - long settled batches as ONE overlap-save convolution per segment of OS_N = 512 x 8192 frames
// (round 4; replaces k_fwd + k_g2_mac + k_inv_wet for the batches it applies to, run_front in mcconv.hip).
//
// What the reference computes per call is one n_ref-point product of whole-IR spectra (conv.cu:367-408).  The
// partitioned engine reaches the same samples through 256-frame blocks zero-padded to 512 (two spectrum entries per input
// frame) and a second transform along the block axis whose chunks lose a fifth of their length to the IR's partitions.
// For a batch of thousands of blocks whose window carries ONE set of gains neither is needed: the batch is cut into
// segments of OS_N - ovl new frames (ovl = the longest sounding IR, rounded up to blocks), each segment - ovl frames of
// history in front - goes through ONE OS_N-point complex transform of z = in1 + j in2, one product per bin
//     W[k] = A[k] Z[k] + B[k] conj(Z[-k]),   A = (C1 - j C2) / 2N,  B = (C1 + j C2) / 2N,
//     C1 = FFT(sum_v g_v (h_v,in1->L + j h_v,in1->R)),  C2 = the same for input 2          (f_pointwiseMultiplyAndScale,
// conv.cu:102-123, 392-401 with the two-for-one split of conv.cu:47-73 folded into A and B) and one inverse transform whose
// real / imaginary parts are the wet L / R frames of the segment (the first ovl are the circular wrap and are dropped).
// One spectrum entry per input frame instead of two, no halo inside a chunk: the row pass below has the shape of
// k_g2_mac's items but there are 256 of them per 16384 - P blocks instead of per 8192 - P.
//
// The OS_N-point transform is the four-step form, n = 8192 n1 + n2, k = k1 + 512 k2:
//   k_os_cols   per n2: 512-point transform over n1 (one wavefront, fft512_wave), times exp(-2 pi i n2 k1 / N)   -> T[k1][n2]
//   k_os_rows   per pair of rows (k1, 512 - k1): 8192-point transforms over n2 in LDS (k_g2_mac's passes), the product -
//               bin (k1, k2) pairs with (512 - k1, 8191 - k2), i.e. with the COMPLEMENT position of the other row in any
//               bit-permuted order - and the inverse row transforms, in place
//   k_os_out    per n2: times exp(+2 pi i n2 k1 / N), inverse 512-point transform over k1, and the output stage of
//               k_inv_wet<true> (Q1/Q2 window sums, Q8 cut terms, predelay, clamp, dry mix; wet ring for the frames later calls reach)
// Rows 0 and 256 pair with themselves (os_rows0_body).  T is kept as float4 [256 items][8192] = {row item, row 512 - item}
// (item 0: rows 0 and 256), so the row pass loads and stores 16 bytes per lane like k_g2_mac's window.
// The spectra A, B are built per (IR set, gains) by the same passes run on the gain-weighted taps (k_os_ir_mix,
// k_os_cols, k_os_rows_fwd, k_os_ir_combine) and cached by the host; a gain change costs one rebuild (~0.2 ms).
// The Q1/Q2 block sums {S1, S2, A1, A2} (conv.cu:55-71) that k_fwd reads off its DC / Nyquist bins come from the column
// pass as sixteen partial sums per block (a tile holds 16 frames of 512 different blocks), summed by the prefix kernels.
#pragma once

#define OS_N1 FFT_N
#define OS_N2 G2_N
#define OS_N (OS_N1 * OS_N2)
#define OS_LOG2_N2 13
#define OS_TILE 16                  // columns (n2) per workgroup of the column passes: one wavefront each
#define OS_THREADS (64 * OS_TILE)
#define OS_WSTR 580                 // float2 entries between the wave-private transform buffers (576 + 4: the output stage reads four of them per lane)
#define OS_TPS (OS_N2 / OS_TILE)    // tiles per segment
#define OS_ITEMS 256                // row pairs per segment
#define OS_ROWT 17                  // padded row length of the transposed tiles

static_assert(OS_N2 == (1 << OS_LOG2_N2), "row length");

struct OsGeo {
    int64_t hop;   // new frames per segment = OS_N - ovl
    int64_t ovl;   // frames of history in front of a segment (multiple of 256, >= taps - 1)
    int64_t n_in;  // frames of the batch (multiple of 256)
    int64_t tau0;  // absolute frame of the batch's frame 0
    int64_t base;  // frame of the batch at which segment 0's new frames begin (0; a block-sliced engine: where its window begins)
    int64_t wet_end;  // wet frames of the batch from here on are not produced (n_in; a block-sliced engine: the end of its slice)
    int seg0;      // first segment of this launch
};

#ifndef OS_ABL
#define OS_ABL 0  // timing-only ablations (wrong results), bit flags: 1 no inter-pass twiddles, 2 no Q1/Q2 window sums in the output pass, 4 no partial block sums, 8 no column transforms
#endif
#ifndef OS_XG
#define OS_XG 16  // consecutive tiles of a segment that run on ONE XCD back to back (their 64-byte pieces of a line meet in that L2)
#endif

// workgroup -> (segment of the launch, tile): workgroup ids 8 apart share an XCD; a group of OS_XG neighbouring tiles takes
// consecutive turns there, so the 64-byte pieces they read of the same 128-byte lines (and write of the same rows) meet in one L2
__device__ __forceinline__ void os_tile_of(int bid, int& segl, int& tile) {
    segl = bid / OS_TPS;
    const int r = bid % OS_TPS;
#if OS_XG > 1
    const int x = r & 7, i = r >> 3;              // XCD lane, turn on it
    const int grp = (i / OS_XG) * 8 + x;          // group of OS_XG tiles
    tile = grp * OS_XG + (i % OS_XG);
#else
    tile = r;
#endif
}

__device__ __forceinline__ v2f os_cis(float x) {  // exp(i pi x)
    float sn, cs;
    sincospif(x, &sn, &cs);
    return v2f{cs, sn};
}

// w[j] = exp(SIGN 2 pi i n2 (lane + 64 j) / OS_N): two evaluations and a power ladder three products deep
template <int SIGN>
__device__ __forceinline__ void os_twiddles(int n2, int lane, v2f (&w)[8]) {
    const float sc = (float)SIGN * 2.0f / (float)OS_N;
    const v2f base = os_cis((float)(n2 * lane) * sc);  // n2 lane < 2^19: exact
    const v2f s1 = os_cis((float)(n2 * 64) * sc);
    const v2f s2 = vx_mul(s1, s1), s3 = vx_mul(s2, s1), s4 = vx_mul(s2, s2);
    w[0] = base;
    w[1] = vx_mul(base, s1);
    w[2] = vx_mul(base, s2);
    w[3] = vx_mul(base, s3);
    w[4] = vx_mul(base, s4);
    w[5] = vx_mul(w[4], s1);
    w[6] = vx_mul(w[4], s2);
    w[7] = vx_mul(w[4], s3);
#if OS_ABL & 1
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = v2f{1.f, 0.f};
#endif
}

// ---------------------------------------------------------------------------
// Column pass.  grid = segments x 512 tiles, block = 1024 (16 wavefronts = 16 columns n2).
// A tile reads, for each n1, 16 consecutive frames of both inputs (64 bytes each) - from the batch's buffers, from the
// input-history ring where the segment reaches back before the batch (xhist != null), zero beyond the batch - and writes
// 256 bytes per item row.  part != null: the tile's sixteenth of the sums {S1, S2, A1, A2} of the block each row lies in,
// part[(seg 512 + tile) 512 + n1] (256 contiguous bytes per wave and store; CorrArgs::parts gathers a block's sixteen).
// Bounds: src frames are tested against [0, n_in); the ring index is masked; Tbuf[(seg 256 + item) 8192 + n2], seg <
// gridDim.x / 512, item < 256, n2 < 8192; part index < segments 512 512.
// LDS: the staging tile [512][17] float2, the 16 transform buffers and the transposed tile [256][17] float4 share one
// 74 KB region (two workgroups per CU).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(OS_THREADS) void k_os_cols(const float* __restrict__ in1, const float* __restrict__ in2,
                                                        const float* __restrict__ xhist, int xr, OsGeo G,
                                                        float4* __restrict__ Tbuf, float4* __restrict__ part,
                                                        const float2* __restrict__ g_tw) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ __align__(16) float2 s_mem[OS_TILE * OS_WSTR];
    static_assert(OS_TILE * OS_WSTR >= OS_N1 * OS_ROWT, "staging tile fits");
    static_assert(OS_TILE * OS_WSTR >= OS_ITEMS * OS_ROWT * 2, "transposed tile fits");
    load_twiddles(s_tw, g_tw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int segl, tile;
    os_tile_of((int)blockIdx.x, segl, tile);
    const int n2_0 = tile * OS_TILE;
    const int64_t seg = (int64_t)G.seg0 + segl;
    const int64_t sbase = G.base + seg * G.hop - G.ovl + n2_0;  // batch-relative frame of (n1 = 0, column 0)
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int e = tid + OS_THREADS * it, n1 = e >> 2, q = e & 3;
        const int64_t src = sbase + (int64_t)OS_N2 * n1 + 4 * q;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (src >= 0) {
            if (src < G.n_in) {
                a = *reinterpret_cast<const float4*>(in1 + src);
                b = *reinterpret_cast<const float4*>(in2 + src);
            }
        } else if (xhist && G.tau0 + src >= 0) {
            const size_t at = (size_t)((G.tau0 + src) & (int64_t)(xr - 1));
            a = *reinterpret_cast<const float4*>(xhist + at);
            b = *reinterpret_cast<const float4*>(xhist + xr + at);
        }
        float2* z = s_mem + n1 * OS_ROWT + 4 * q;
        z[0] = make_float2(a.x, b.x);
        z[1] = make_float2(a.y, b.y);
        z[2] = make_float2(a.z, b.z);
        z[3] = make_float2(a.w, b.w);
        if (part && !(OS_ABL & 4)) {  // (kernel-uniform) frames 4 q + k of a block-aligned run of 16: parity of the frame = parity of k
            float s1 = (a.x + a.y) + (a.z + a.w), s2 = (b.x + b.y) + (b.z + b.w);
            float d1 = (a.x - a.y) + (a.z - a.w), d2 = (b.x - b.y) + (b.z - b.w);
            s1 += __shfl_xor(s1, 1), s2 += __shfl_xor(s2, 1), d1 += __shfl_xor(d1, 1), d2 += __shfl_xor(d2, 1);
            s1 += __shfl_xor(s1, 2), s2 += __shfl_xor(s2, 2), d1 += __shfl_xor(d1, 2), d2 += __shfl_xor(d2, 2);
            // (every row is written; the prefix kernels read the rows of the blocks the segment owns)
            if (q == 0) part[((size_t)segl * OS_TPS + tile) * OS_N1 + n1] = make_float4(s1, s2, d1, d2);
        }
    }
    __syncthreads();
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = s_mem[(lane + 64 * r) * OS_ROWT + wave];
    __syncthreads();  // every column is in registers: the staging tile becomes the transform buffers
    float2* lds = s_mem + wave * OS_WSTR;
#if OS_ABL & 8
#pragma unroll
    for (int r = 0; r < 8; r++) lds[lane + 64 * r] = v[r];
#else
    fft512_wave<-1, false>(v, lds, s_tw, lane);
#endif
    v2f z[8];
    {
        v2f w[8];
        os_twiddles<-1>(n2_0 + wave, lane, w);
#pragma unroll
        for (int j = 0; j < 8; j++) z[j] = vx_mul(vx_of(lds[lane + 64 * j]), w[j]);
    }
    __syncthreads();  // every wave has read its transform: the buffers become the transposed tile
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k1 = lane + 64 * j;
        const int item = k1 < 256 ? k1 : ((OS_N1 - k1) & 255), half = k1 >= 256 ? 1 : 0;
        s_mem[(item * OS_ROWT + wave) * 2 + half] = vx_to(z[j]);
    }
    __syncthreads();
    float4* dst = Tbuf + ((size_t)segl * OS_ITEMS << OS_LOG2_N2) + n2_0;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int e = tid + OS_THREADS * it, item = e >> 4, c = e & 15;
        dst[((size_t)item << OS_LOG2_N2) + c] = *reinterpret_cast<const float4*>(&s_mem[(item * OS_ROWT + c) * 2]);
    }
}

// Rows 0 and 256 of a segment: bin (0, k2) pairs with (0, -k2), bin (256, k2) with (256, 8191 - k2).  The workgroup of
// k_os_rows whose item is 0 takes them one after the other through its one buffer.  SP0[row][p] = {A, B} at the
// transform's position p.
__device__ __forceinline__ void os_rows0_body(float2* s, const float2* t_lo, const float2* t_hi, float4* __restrict__ row,
                                              const float4* __restrict__ SP0) {
    const int tt = threadIdx.x;
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
        for (int i = tt; i < G2_N; i += G2B_THREADS) s[G2_P(i)] = reinterpret_cast<const float2*>(row + i)[c];
        __syncthreads();
        g2_forward(s, t_lo, t_hi, tt);
        float2 w[G2_N / G2B_THREADS];
#pragma unroll
        for (int m = 0; m < G2_N / G2B_THREADS; m++) {
            const int p = tt + G2B_THREADS * m, pm = c == 0 ? g2_mirror(p) : G2_N - 1 - p;
            const float2 X = s[G2_P(p)], Xm = s[G2_P(pm)];
            const float4 ab = SP0[(size_t)c * G2_N + p];
            w[m] = vx_to(vx_mul(vx_of(X), v2f{ab.x, ab.y}) + vx_mulc(v2f{ab.z, ab.w}, vx_of(Xm)));
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < G2_N / G2B_THREADS; m++) s[G2_P(tt + G2B_THREADS * m)] = w[m];
        __syncthreads();
        g2_inverse(s, t_lo, t_hi, tt);
        for (int i = tt; i < G2_N; i += G2B_THREADS) reinterpret_cast<float2*>(row + i)[c] = s[G2_P(i)];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Row pass: k_g2_mac's per-item sequence on the rows (item, 512 - item) of one segment, in place.
//   window (16 bytes per n2 carry both rows) -> forward(row A) -> own entries to registers -> forward(row B) ->
//   products against the item's spectra (row B's results replace it in LDS, row A's stay in registers) ->
//   inverse(row B) -> inverse(row A) -> store.
// A thread owns the entry pairs (2 j, 2 j + 1), j = tt + 512 r, of row A; their partners in row B are the pairs
// 4095 - j, reversed - which no other thread touches, so the products need no barrier.
// Spectra SP[((item 8 + r) 4 + w) 512 + tt] = {A, B} of: w = 0 row A entry 2 j, 1 row A entry 2 j + 1, 2 row B entry
// 2 j' + 1, 3 row B entry 2 j' (j' = 4095 - j): four fully coalesced 16-byte loads per pair.
// grid = segments x 256 (item 0: os_rows0_body), block = 512, two workgroups per CU.
// Bounds: T[(seg 256 + item) 8192 + tt + 512 r], r < 16; SP index < 256 * 8 * 4 * 512; LDS G2_P(8191) < G2_LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(G2B_THREADS, 4) void k_os_rows(float4* __restrict__ Tbuf, const float4* __restrict__ SP,
                                                            const float4* __restrict__ SP0, int nseg) {
    __shared__ float2 s[G2_LDS];
    __shared__ float2 t_lo[128], t_hi[64];
    // items 8 apart in launch order share an XCD: an item's segments follow each other there (its spectra are read into that L2 once)
    const int xq = (int)blockIdx.x >> 3;
    const int item = (xq / nseg) * 8 + ((int)blockIdx.x & 7), seg = xq % nseg;
#ifndef OS_PRIO
#define OS_PRIO 0
#endif
    if (OS_PRIO && nseg * OS_ITEMS >= 2048) {  // issue priority for one of a CU's two workgroups (k_g2_mac, G2_PRIO)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid & 0xfu) >= 2u) __builtin_amdgcn_s_setprio(OS_PRIO);
    }
    g2_tables(t_lo, t_hi);
    __syncthreads();
    constexpr int ROWS = G2_N / G2B_THREADS;
    constexpr int PS = 2 * G2B_THREADS + 2 * G2B_THREADS / 32;
    int tt = threadIdx.x;
    asm volatile("" : "+v"(tt));
    float4* row = Tbuf + ((size_t)(seg * OS_ITEMS + item) << OS_LOG2_N2);
    if (item == 0) {  // (workgroup-uniform) the two rows that pair with themselves
        os_rows0_body(s, t_lo, t_hi, row, SP0);
        return;
    }
    v2f x1[ROWS], x2[ROWS];
    {
        float4 x[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) x[r] = row[tt + G2B_THREADS * r];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            x1[r] = v2f{x[r].x, x[r].y};
            x2[r] = v2f{x[r].z, x[r].w};
        }
    }
    g2_pair<false, 9, false, true>(s, t_lo, t_hi, tt, tt, x1);
    __syncthreads();
    g2_pair<false, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
    G2B_WAVE_SYNC();
    g2_pair<false, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
    __syncthreads();
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
    __syncthreads();
    g2_pair<false, 9, false, true>(s, t_lo, t_hi, tt, tt, x2);
    __syncthreads();
    g2_pair<false, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
    G2B_WAVE_SYNC();
    g2_pair<false, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
    __syncthreads();
    {
        asm volatile("" : "+v"(tt));
        float2* pm = &s[G2_P(2 * (G2B_THREADS - 1 - tt))];  // pair 4095 - j = (511 - tt) + 512 (7 - r)
        constexpr int NP = ROWS / 2, RING = G2B_AHEAD + 1;
        float4 Sq[RING][4];
        const float4* sp = SP + (size_t)item * (NP * 4 * G2B_THREADS) + tt;
        auto request = [&](int r) {
#pragma unroll
            for (int w = 0; w < 4; w++) Sq[r % RING][w] = sp[(size_t)(r * 4 + w) * G2B_THREADS];
        };
#pragma unroll
        for (int r = 0; r < G2B_AHEAD; r++) request(r);
#pragma unroll
        for (int r = 0; r < NP; r++) {
            if (r + G2B_AHEAD < NP) request(r + G2B_AHEAD);
            float2* pb = pm + (NP - 1 - r) * PS;
            const v2f a = vx_ld(pb), b = vx_ld(pb + 1);
            const v2f XB0 = a + b, XB1 = a - b;  // row B entries 2 j', 2 j' + 1
            const v2f XA0 = X1[2 * r], XA1 = X1[2 * r + 1];
            const float4 S0 = Sq[r % RING][0], S1 = Sq[r % RING][1], S2 = Sq[r % RING][2], S3 = Sq[r % RING][3];
            const v2f WA0 = vx_mul(XA0, v2f{S0.x, S0.y}) + vx_mulc(v2f{S0.z, S0.w}, XB1);
            const v2f WA1 = vx_mul(XA1, v2f{S1.x, S1.y}) + vx_mulc(v2f{S1.z, S1.w}, XB0);
            const v2f WB1 = vx_mul(XB1, v2f{S2.x, S2.y}) + vx_mulc(v2f{S2.z, S2.w}, XA0);
            const v2f WB0 = vx_mul(XB0, v2f{S3.x, S3.y}) + vx_mulc(v2f{S3.z, S3.w}, XA1);
            vx_st(pb, WB0 + WB1);  // the inverse transform's radix-2 stage on the way out
            vx_st(pb + 1, WB0 - WB1);
            X1[2 * r] = WA0 + WA1;
            X1[2 * r + 1] = WA0 - WA1;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    v2f yb[ROWS];
    g2_pair<true, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
    G2B_WAVE_SYNC();
    g2_pair<true, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
    __syncthreads();
    g2_pair<true, 9, true, false>(s, t_lo, t_hi, tt, tt, yb);
    __syncthreads();
    {
        asm volatile("" : "+v"(tt));
        float2* pp = &s[G2_P(2 * tt)];
#pragma unroll
        for (int r = 0; r < ROWS / 2; r++) {
            vx_st(pp + r * PS, X1[2 * r]);
            vx_st(pp + r * PS + 1, X1[2 * r + 1]);
        }
    }
    __syncthreads();
    {
        v2f ya[ROWS];
        g2_pair<true, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
        G2B_WAVE_SYNC();
        g2_pair<true, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
        __syncthreads();
        g2_pair<true, 9, true, false>(s, t_lo, t_hi, tt, tt, ya);
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int m = 0; m < ROWS; m++) row[tt + G2B_THREADS * m] = make_float4(ya[m].x, ya[m].y, yb[m].x, yb[m].y);
    }
}

// ---------------------------------------------------------------------------
// Output pass: inverse column transforms and k_inv_wet<true>'s output stage.  grid = segments x 512, block = 1024.
// The lane that holds four consecutive wet frames {L = Re, R = Im} of the batch finishes them: Q1/Q2 window sums from the
// prefix ring (out_window), clamp, dry mix (out_frame), stored at the predelay offset; the frames later calls can reach
// (blocks < wet_head or >= wet_from) also go to the wet ring.  OutArgs as for k_inv_wet (lin unused: null; drop != null in the Q8 regime: the cut terms of the batch's output frames).
// Bounds: wet frames i0 in [base + seg hop, min(base + (seg + 1) hop, wet_end)), wet_end <= n_in; output frames tested against [out_from, out_end) blocks;
// ring indices masked.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(OS_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_os_out(const float4* __restrict__ Tbuf, OsGeo G, float* __restrict__ wet, int wr,
                                                       const float2* __restrict__ g_tw, OutArgs oa) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ __align__(16) float2 s_mem[OS_TILE * OS_WSTR];
    load_twiddles(s_tw, g_tw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int segl, tile;
    os_tile_of((int)blockIdx.x, segl, tile);
    const int n2_0 = tile * OS_TILE;
    const int64_t seg = (int64_t)G.seg0 + segl;
    const float4* src = Tbuf + ((size_t)segl * OS_ITEMS << OS_LOG2_N2) + n2_0;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int e = tid + OS_THREADS * it, item = e >> 4, c = e & 15;
        *reinterpret_cast<float4*>(&s_mem[(item * OS_ROWT + c) * 2]) = src[((size_t)item << OS_LOG2_N2) + c];
    }
    __syncthreads();
    float2 v[8];
    {
        v2f w[8];
        os_twiddles<+1>(n2_0 + wave, lane, w);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int k1 = lane + 64 * r;
            const int item = k1 < 256 ? k1 : ((OS_N1 - k1) & 255), half = k1 >= 256 ? 1 : 0;
            v[r] = vx_to(vx_mul(vx_of(s_mem[(item * OS_ROWT + wave) * 2 + half]), w[r]));
        }
    }
    __syncthreads();
#if OS_ABL & 8
#pragma unroll
    for (int r = 0; r < 8; r++) s_mem[wave * OS_WSTR + lane + 64 * r] = v[r];
#else
    fft512_wave<+1, false>(v, s_mem + wave * OS_WSTR, s_tw, lane);
#endif
    __syncthreads();
#ifdef OS_OUT_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int it = 0; it < 2; it++) {
        const int e = tid + OS_THREADS * it, n1 = e >> 2, q = e & 3;
        const int64_t n = (int64_t)OS_N2 * n1 + n2_0 + 4 * q;
        const int64_t i0 = G.base + seg * G.hop - G.ovl + n;  // wet frame of the batch
        if (n < G.ovl || i0 >= G.wet_end) continue;
        const float2* zz = s_mem + (4 * q) * OS_WSTR + n1;
        const float2 z0 = zz[0], z1 = zz[OS_WSTR], z2 = zz[2 * OS_WSTR], z3 = zz[3 * OS_WSTR];
        const float4 wl4 = make_float4(z0.x, z1.x, z2.x, z3.x), wr4 = make_float4(z0.y, z1.y, z2.y, z3.y);
        const int t = (int)(i0 >> 8);
        if (i0 >= 0 && (t < oa.wet_head || t >= oa.wet_from)) {
            const size_t at = (size_t)((G.tau0 + i0) & (int64_t)(wr - 1));
            *reinterpret_cast<float4*>(wet + at) = wl4;
            *reinterpret_cast<float4*>(wet + wr + at) = wr4;
        }
        const int64_t o0 = i0 + oa.predelay;
        const bool emits = o0 + 3 >= (int64_t)oa.out_from * MC_B && o0 < (int64_t)oa.out_end * MC_B;
        if (!emits) continue;
        const bool whole = ((o0 | oa.n_ref) & 3) == 0 && o0 >= (int64_t)oa.out_from * MC_B;
        const int64_t u0 = oa.tabs0 * MC_B + i0;
        double win[4];
        if (whole) {
            const float4 x1q = *reinterpret_cast<const float4*>(oa.in1 + o0), x2q = *reinterpret_cast<const float4*>(oa.in2 + o0);
#if OS_ABL & 2
            win[0] = win[1] = win[2] = win[3] = 0.0;
#else
            out_window(oa, u0, win);
#endif
            const BlockParams& bp = oa.ptab[(o0 >> 8) * oa.pstride];
            float4 d01 = make_float4(0.f, 0.f, 0.f, 0.f), d23 = d01;  // the Q8 cut terms of the four frames {L, R}, subtracted before the clamp (as k_inv_wet)
            if (oa.drop) {
                const float4* dp = reinterpret_cast<const float4*>(oa.drop + o0);
                d01 = dp[0], d23 = dp[1];
            }
            float4 fl, fr;
            out_frame(false, wl4.x - d01.x, wr4.x - d01.y, x1q.x, x2q.x, bp, win, fl.x, fr.x);
            out_frame(true, wl4.y - d01.z, wr4.y - d01.w, x1q.y, x2q.y, bp, win, fl.y, fr.y);
            out_frame(false, wl4.z - d23.x, wr4.z - d23.y, x1q.z, x2q.z, bp, win, fl.z, fr.z);
            out_frame(true, wl4.w - d23.z, wr4.w - d23.w, x1q.w, x2q.w, bp, win, fl.w, fr.w);
            const int64_t os = o0 - (int64_t)oa.out_blk0 * MC_B;
            *reinterpret_cast<float4*>(oa.outL + os) = fl;
            *reinterpret_cast<float4*>(oa.outR + os) = fr;
        } else {
#pragma unroll 1
            for (int k = 0; k < 4; k++) {  // a predelay that is no multiple of four frames: frame by frame
                const int64_t o = o0 + k;
                if (o < (int64_t)oa.out_from * MC_B || o >= (int64_t)oa.out_end * MC_B) continue;
                const BlockParams& bp = oa.ptab[(o >> 8) * oa.pstride];
                const float a = k == 0 ? wl4.x : (k == 1 ? wl4.y : (k == 2 ? wl4.z : wl4.w));
                const float b = k == 0 ? wr4.x : (k == 1 ? wr4.y : (k == 2 ? wr4.z : wr4.w));
                float fl, fr;
                out_window(oa, u0 + k, win);
                const float2 d = oa.drop ? oa.drop[o] : make_float2(0.f, 0.f);
                out_frame((k & 1) != 0, a - d.x, b - d.y, oa.in1[o], oa.in2[o], bp, win, fl, fr);
                oa.outL[o - (int64_t)oa.out_blk0 * MC_B] = fl;
                oa.outR[o - (int64_t)oa.out_blk0 * MC_B] = fr;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Spectra of an (IR set, gains) pair.
// ---------------------------------------------------------------------------
struct OsMix {
    int n;                         // voices
    const float2* h0[MC_MAXV];     // taps {L, R} of the voice's IR for input 1 / input 2
    const float2* h1[MC_MAXV];
    int L0[MC_MAXV], L1[MC_MAXV];  // their lengths
    float4 g[MC_MAXV];             // {L<-in1, L<-in2, R<-in1, R<-in2}
};

// planes [4][np]: Re / Im of c1 = sum_v (g.x h0.L + j g.z h0.R), then of c2 = sum_v (g.y h1.L + j g.w h1.R)
__global__ __launch_bounds__(256) void k_os_ir_mix(OsMix M, float* __restrict__ planes, int64_t np) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    float c1r = 0.f, c1i = 0.f, c2r = 0.f, c2i = 0.f;
#pragma unroll
    for (int v = 0; v < MC_MAXV; v++) {
        if (v >= M.n) break;
        if (i < M.L0[v]) {
            const float2 h = M.h0[v][i];
            c1r += M.g[v].x * h.x;
            c1i += M.g[v].z * h.y;
        }
        if (i < M.L1[v]) {
            const float2 h = M.h1[v][i];
            c2r += M.g[v].y * h.x;
            c2i += M.g[v].w * h.y;
        }
    }
    planes[i] = c1r;
    planes[np + i] = c1i;
    planes[2 * np + i] = c2r;
    planes[3 * np + i] = c2i;
}

// forward row transforms of a column-pass result, in place (all 256 items, both rows); grid = 256, block = 1024
__global__ __launch_bounds__(G2_THREADS) void k_os_rows_fwd(float4* __restrict__ buf) {
    __shared__ float2 s[2][G2_LDS];
    __shared__ float2 t_lo[128], t_hi[64];
    const int c = threadIdx.x >> 9, tt = threadIdx.x & 511;
    float4* row = buf + ((size_t)blockIdx.x << OS_LOG2_N2);
    g2_tables(t_lo, t_hi);
    for (int i = tt; i < G2_N; i += 512) {
        const float4 v = row[i];
        s[c][G2_P(i)] = c == 0 ? make_float2(v.x, v.y) : make_float2(v.z, v.w);
    }
    __syncthreads();
    g2_forward(s[c], t_lo, t_hi, tt);
    for (int i = threadIdx.x; i < G2_N; i += G2_THREADS) {
        const float2 a = s[0][G2_P(i)], b = s[1][G2_P(i)];
        row[i] = make_float4(a.x, a.y, b.x, b.y);
    }
}

// A = (c1 - j c2) sc, B = (c1 + j c2) sc
__device__ __forceinline__ float4 os_ab(float2 c1, float2 c2, float sc) {
    return make_float4((c1.x + c2.y) * sc, (c1.y - c2.x) * sc, (c1.x - c2.y) * sc, (c1.y + c2.x) * sc);
}

// C1, C2: row spectra [256 items][8192] {row item, row 512 - item} -> SP (items 1..255, k_os_rows' order) and SP0 (item 0)
// grid = 256, block = 512
__global__ __launch_bounds__(512) void k_os_ir_combine(const float4* __restrict__ C1, const float4* __restrict__ C2,
                                                       float4* __restrict__ SP, float4* __restrict__ SP0, float sc) {
    const int item = blockIdx.x, tt = threadIdx.x;
    const float4* r1 = C1 + ((size_t)item << OS_LOG2_N2);
    const float4* r2 = C2 + ((size_t)item << OS_LOG2_N2);
    if (item == 0) {
        for (int p = tt; p < G2_N; p += 512) {
            const float4 a = r1[p], b = r2[p];
            SP0[p] = os_ab(make_float2(a.x, a.y), make_float2(b.x, b.y), sc);
            SP0[G2_N + p] = os_ab(make_float2(a.z, a.w), make_float2(b.z, b.w), sc);
        }
        return;
    }
    float4* sp = SP + (size_t)item * (8 * 4 * 512) + tt;
    for (int r = 0; r < 8; r++) {
        const int j = tt + 512 * r, jp = 4095 - j;
        const float4 a0 = r1[2 * j], a1 = r1[2 * j + 1], b0 = r1[2 * jp], b1 = r1[2 * jp + 1];
        const float4 e0 = r2[2 * j], e1 = r2[2 * j + 1], f0 = r2[2 * jp], f1 = r2[2 * jp + 1];
        sp[(size_t)(r * 4 + 0) * 512] = os_ab(make_float2(a0.x, a0.y), make_float2(e0.x, e0.y), sc);
        sp[(size_t)(r * 4 + 1) * 512] = os_ab(make_float2(a1.x, a1.y), make_float2(e1.x, e1.y), sc);
        sp[(size_t)(r * 4 + 2) * 512] = os_ab(make_float2(b1.z, b1.w), make_float2(f1.z, f1.w), sc);
        sp[(size_t)(r * 4 + 3) * 512] = os_ab(make_float2(b0.z, b0.w), make_float2(f0.z, f0.w), sc);
    }
}
