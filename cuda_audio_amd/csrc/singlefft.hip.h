// singlefft.hip.h — the path in the reference's OWN shape (mc_config.form = 1; BASELINE config 2, SURVEY §8(d)
// "Config 2"): one n_ref-point transform per call instead of 256-tap partitions.  Per call of nframes frames
// (Convolution::onProcess, conv.cu:287-466):
//
//   z = in1 + j in2, zero padded to N            conv.cu:35-45, 321-328
//   live IR spectra += (wet b - live) / (vsteps + 5)   f_interpolate, conv.cu:15-32, 339-353 - literally, per bin
//   Z = FFT_N(z); X1, X2 = two-for-one split     conv.cu:367, f_unpackC22R :47-73 with its s == 0 shortcut (Q1) and
//                                                the N/2 entry that is never written (Q2)
//   Y_c = (X1 L_0c s_0c + X2 L_1c s_1c)          f_pointwiseMultiplyAndScale, conv.cu:102-123, 392-401 (true product)
//   y_c = IFFT_N(Y_c)                            conv.cu:403-408
//   acc_c[s] = clamp(acc_c[s] + y_c[s - predelay]), s < N     f_pointwiseAdd, conv.cu:89-100 - the RUNNING accumulator
//                                                is clamped (Q4) and what the shift pushes past N is dropped (Q8)
//   out = acc[0 .. nframes) + dry mix            f_addDryInterleaved, conv.cu:126-140, 418-427
//   acc slides by nframes                        conv.cu:440-451 (here: a ring of N slots, the origin moves)
//
// Differences from a literal translation, none of which changes a sample:
//   * only the REAL part of the reference's complex accumulators is ever heard (.x, conv.cu:431-437) and the imaginary
//     part never feeds back into it, so Y_L and Y_R go through ONE packed inverse transform (their Hermitian parts:
//     Re Y[0], 0 at N/2) and the accumulators are real;
//   * the input has nframes <= 1024 non-zero samples: with N = 512 M, n = a + 512 b, k = d + M c the forward
//     transform is M transforms of 512 points of the twiddled input (pruned four-step: the pass over b has one or two
//     non-zero terms, folded into the input), each on one wavefront (fft512_wave);
//   * spectra are kept for bins 0 .. N/2 - 1 only, in the order the four-step passes touch them ([d][c] for bin d + M c);
//     the mirrored half of the reference's buffers is implied.
// The inverse is the plain four-step: 512-point transforms over c (k_sf_inv1), twiddle, M-point transforms over d in
// LDS (radix-2 Stockham, k_sf_inv2) or on a wavefront (k_sf_inv2w), whose epilogue accumulates, clamps and emits the period.
// IR preparation (Convolution::prepare, conv.cu:207-253) is the same four-step forward on the packed L + jR taps
// (k_sf_ir_cols, k_sf_ir_rows) and the split (k_sf_ir_unpack).
//
// Bytes per call at N = 131072 (all of it lives in the 256 MB last-level cache): live spectra 2 MiB read + 2 MiB
// written, selected IRs 2 MiB, packed Y 1 MiB w + r, pass-1 result 1 MiB w + r, accumulators 1 MiB r + w = 12 MiB in three
// launches; SURVEY §8(d) counts 3 MiB algorithmic (IR half-spectra + input window).
#pragma once

#define SF_ROWS 8  // 512-point transforms (wavefronts) per workgroup of the row kernels

struct SfCall {
    const float *in1, *in2;  // nframes samples each
    float *outL, *outR;
    int nframes;
    int pd;                  // predelay of half 0 (conv.cu:412,415)
    float wet[2], div[2];    // f_interpolate: live += (wet b - live) / div, div = vsteps + 5
    const float2* b[2];      // spectra of the selected IRs: [H_L | H_R], N/2 bins each
    float sc[2][2];          // [c][i] = pan_c(panWet_i) level_i / N       (conv.cu:386-401)
    float dry[2][2];         // [c][i] = dry_i pan_c(panDry_i) level_i     (conv.cu:418-427)
    unsigned base;           // accumulator slot of this call's output frame 0
    unsigned* done_flag;     // != null (mapped host memory; outL / outR are too): `seq` is stored once the period is on the host
    unsigned seq;
};

__device__ __forceinline__ float2 sf_cis(unsigned ph, int N, float sign) {  // exp(sign 2 pi i ph / N), ph < N
    float sn, cs;
    sincospif(sign * 2.0f * (float)ph / (float)N, &sn, &cs);
    return make_float2(cs, sn);
}

// ---------------------------------------------------------------------------
// Layout.  With N = 512 M a bin k = d + M c is kept at [d][c]: the half spectra
// (IRs, live spectra; bins below N/2 <=> c < 256) as [M][256], the packed output
// spectrum W as [M][512].  Row d is what one 512-point transform of the four-step
// passes produces or consumes, so every kernel below touches whole rows.
// The mirror of bin (d, c) is (M - d, 511 - c) for d > 0 and (0, 512 - c) for d = 0.
// ---------------------------------------------------------------------------
// S1 + S2: X1, X2 of the rows d0 .. d0 + 3 from the period's nframes samples (one
// wavefront per row and input: waves 0 .. 3 take in1, 4 .. 7 in2), then per bin of
// those rows: the four live spectra take their step towards wet x the selected IR
// (f_interpolate), the two output spectra are formed (f_pointwiseMultiplyAndScale)
// and packed for ONE inverse transform, W = Yh_L + j Yh_R with Yh the Hermitian
// part (Re at bin 0, 0 at N/2) - at the bin and at its mirror.
// grid = M / 4, block = 512.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64 * SF_ROWS) void k_sf_fwdmac(SfCall C, int N, int M, float2* __restrict__ live,
                                                            float2* __restrict__ W, const float2* __restrict__ g_tw) {
    constexpr int RD = SF_ROWS / 2;  // rows per workgroup
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[SF_ROWS][FFT_WAVE_LDS];
    load_twiddles(s_tw, g_tw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = blockIdx.x * RD;
    {
        const int i = wave / RD, d = d0 + wave % RD;
        const float* in = i ? C.in2 : C.in1;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            float2 acc = make_float2(0.f, 0.f);
            for (int n = lane + 64 * r; n < C.nframes; n += FFT_N) {  // (the pass over b folded into the input)
                // (system scope: the period may sit in device memory the CPU wrote through the BAR)
                const float x = __hip_atomic_load(in + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const float2 w = sf_cis(((unsigned)d * (unsigned)n) & (unsigned)(N - 1), N, -1.f);
                acc.x += x * w.x;
                acc.y += x * w.y;
            }
            v[r] = acc;
        }
        __syncthreads();
        fft512_wave<-1, false>(v, s_fft[wave], s_tw, lane);
    }
    __syncthreads();
    const int H = N / 2;
    for (int idx = threadIdx.x; idx < RD * 256; idx += 64 * SF_ROWS) {
        const int di = idx >> 8, c = idx & 255, d = d0 + di;
        float2 x[2] = {s_fft[di][c], s_fft[RD + di][c]};
        if (d == 0 && c == 0) {  // the split's s == 0 shortcut (Q1): X1[0] = Z[0] = S1 + j S2, X2[0] = 0
            x[0] = make_float2(x[0].x, x[1].x);
            x[1] = make_float2(0.f, 0.f);
        }
        const size_t p = (size_t)d * 256 + c;
        float2 y[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                float2* lp = live + (size_t)(i * 2 + ch) * H + p;
                const float2 va = *lp, b = C.b[i][(size_t)ch * H + p];
                const float2 vb = make_float2(b.x * C.wet[i], b.y * C.wet[i]);
                const float2 vv = make_float2(va.x + (vb.x - va.x) / C.div[i], va.y + (vb.y - va.y) / C.div[i]);
                *lp = vv;
                const float2 pr = make_float2(x[i].x * vv.x - x[i].y * vv.y, x[i].x * vv.y + x[i].y * vv.x);
                y[ch].x += pr.x * C.sc[ch][i];
                y[ch].y += pr.y * C.sc[ch][i];
            }
        if (d == 0 && c == 0) {
            W[0] = make_float2(y[0].x, y[1].x);
            W[256] = make_float2(0.f, 0.f);  // bin N/2 = (0, 256)
        } else {
            W[(size_t)d * FFT_N + c] = make_float2(y[0].x - y[1].y, y[0].y + y[1].x);  // Y_L + j Y_R
            const size_t mir = d ? (size_t)(M - d) * FFT_N + (FFT_N - 1 - c) : (size_t)(FFT_N - c);
            W[mir] = make_float2(y[0].x + y[1].y, -y[0].y + y[1].x);  // conj(Y_L) + j conj(Y_R)
        }
    }
}

// ---------------------------------------------------------------------------
// S3: inverse, pass 1 - for each d: 512 points over c of row d of W -> a,
// times exp(+2 pi i a d / N), to T[a M + d].  grid = M / SF_ROWS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64 * SF_ROWS) void k_sf_inv1(int N, int M, const float2* __restrict__ W, float2* __restrict__ Tm,
                                                          const float2* __restrict__ g_tw) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[SF_ROWS][FFT_WAVE_LDS];
    load_twiddles(s_tw, g_tw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = blockIdx.x * SF_ROWS;
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = W[(size_t)(d0 + wave) * FFT_N + lane + 64 * r];
    __syncthreads();
    fft512_wave<+1, false>(v, s_fft[wave], s_tw, lane);
    __syncthreads();
    for (int idx = threadIdx.x; idx < SF_ROWS * FFT_N; idx += 64 * SF_ROWS) {
        const int di = idx % SF_ROWS, a = idx / SF_ROWS;
        const float2 val = s_fft[di][a];
        const float2 w = sf_cis(((unsigned)a * (unsigned)(d0 + di)) & (unsigned)(N - 1), N, +1.f);
        Tm[(size_t)a * M + d0 + di] = make_float2(val.x * w.x - val.y * w.y, val.x * w.y + val.y * w.x);
    }
}

// M-point transforms of nseq sequences (contiguous, M entries each) in LDS: radix-2 Stockham, log2 M passes between
// the two buffers; tw[m] = exp(SIGN 2 pi i m / M), m < M/2.  Returns the buffer that holds the result (natural order).
__device__ __forceinline__ float2* sf_fft_lds(float2* src, float2* dst, const float2* tw, int M, int nseq) {
    const int half = M >> 1;
    int lgh = 0;
    while ((1 << lgh) < half) lgh++;
    for (int p = 1, lgp = 0; p < M; p <<= 1, lgp++) {
        for (int idx = threadIdx.x; idx < nseq * half; idx += blockDim.x) {
            const int q = idx >> lgh, j = idx & (half - 1);
            const int k = j & (p - 1);
            const float2 u0 = src[q * M + j], x1 = src[q * M + j + half];
            const float2 w = tw[k << (lgh - lgp)];  // exp(SIGN pi i k / p)
            const float2 u1 = make_float2(x1.x * w.x - x1.y * w.y, x1.x * w.y + x1.y * w.x);
            const int j0 = ((j - k) << 1) + k;
            dst[q * M + j0] = make_float2(u0.x + u1.x, u0.y + u1.y);
            dst[q * M + j0 + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2* t = src;
        src = dst;
        dst = t;
    }
    return src;
}

// ---------------------------------------------------------------------------
// S4: inverse, pass 2 - for each a: M points over d of T[a M + d] -> b,
// y[n = a + 512 b] = {y_L, y_R}; acc[slot(n + pd)] = clamp(acc + y) for
// n + pd < N.  The accumulators are kept in the order this pass produces:
// slot t = (base + s) mod N lives at [t mod 512][t / 512], so the M results of
// one a are one (rotated) contiguous row.  A frame of this period (s < nframes)
// leaves the moment its one contribution of this call has been added: out = acc +
// dry mix, and its slot is cleared (it becomes the far end of the accumulator: the
// reference shifts zeros in, conv.cu:440-451); frames before the predelay get no
// contribution and leave as they are (sf_emit_early).
//   k_sf_inv2w (M <= 512): one wavefront per a - the row zero-padded to 512 points,
//                          U[j] = X[j 512 / M]; grid = 512 / SF_ROWS2, block = 64 SF_ROWS2
//   k_sf_inv2  (any M)   : AT rows per workgroup, radix-2 Stockham in LDS;
//                          grid = 512 / AT, block = 256, dynamic LDS = (2 AT M + M / 2) float2
// ---------------------------------------------------------------------------
#define SF_ROWS2 4

__device__ __forceinline__ size_t sf_slot(const SfCall& C, int M, unsigned s) {  // accumulator entry of output frame s of this call
    const unsigned t = C.base + s;
    return (size_t)(t & (FFT_N - 1)) * M + ((t >> 9) & (unsigned)(M - 1));
}

// out = acc + dry mix for a frame of the period; its slot becomes the far end of the accumulator: the reference shifts zeros in
__device__ __forceinline__ void sf_emit(const SfCall& C, unsigned s, float l, float r) {
    const float x1 = __hip_atomic_load(C.in1 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const float x2 = __hip_atomic_load(C.in2 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    C.outL[s] = l + (x1 * C.dry[0][0] + x2 * C.dry[0][1]);
    C.outR[s] = r + (x1 * C.dry[1][0] + x2 * C.dry[1][1]);
}

// y[n = a + 512 b] into the accumulator at frame s = n + predelay; frames of this period leave at once
__device__ __forceinline__ void sf_accumulate(const SfCall& C, int N, int M, int a, int b, float2 v, float* __restrict__ acc) {
    const unsigned s = (unsigned)C.pd + (unsigned)a + (unsigned)FFT_N * (unsigned)b;
    if (s >= (unsigned)N) return;  // pushed past the end: dropped (Q8)
    const size_t at = sf_slot(C, M, s);
    float l = fminf(fmaxf(acc[at] + v.x, -1.f), 1.f), r = fminf(fmaxf(acc[(size_t)N + at] + v.y, -1.f), 1.f);
    if (s < (unsigned)C.nframes) {
        sf_emit(C, s, l, r);
        l = r = 0.f;
    }
    acc[at] = l;
    acc[(size_t)N + at] = r;
}

// frames of the period that lie before the predelay get no contribution from this call: what the accumulator holds leaves
// as it is.  Called once for every a < 512 (frame s belongs to a = s mod 512).
__device__ __forceinline__ void sf_emit_early(const SfCall& C, int N, int M, int a, float* __restrict__ acc) {
    const unsigned lim = (unsigned)min(C.pd, C.nframes);
    for (unsigned s = (unsigned)a; s < lim; s += FFT_N) {
        const size_t at = sf_slot(C, M, s);
        sf_emit(C, s, acc[at], acc[(size_t)N + at]);
        acc[at] = 0.f;
        acc[(size_t)N + at] = 0.f;
    }
}

// mc_process: the last workgroup to get here publishes the sequence number once the whole period is on the host
__device__ __forceinline__ void sf_publish_if_last(const SfCall& C, unsigned* __restrict__ done_ctr) {
    if (!C.done_flag) return;
    __syncthreads();  // every lane's stores have been issued and acknowledged (vmcnt(0) at the barrier)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: this workgroup's frames are on the host
        if (atomicAdd(done_ctr, 1u) == gridDim.x - 1) {
            *done_ctr = 0;
            __hip_atomic_store(C.done_flag, C.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(64 * SF_ROWS2) void k_sf_inv2w(SfCall C, int N, int M, const float2* __restrict__ Tm,
                                                            float* __restrict__ acc, unsigned* __restrict__ done_ctr,
                                                            const float2* __restrict__ g_tw) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[SF_ROWS2][FFT_WAVE_LDS];
    load_twiddles(s_tw, g_tw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = blockIdx.x * SF_ROWS2 + wave;
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int d = lane + 64 * r;
        v[r] = d < M ? Tm[(size_t)a * M + d] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    fft512_wave<+1, false>(v, s_fft[wave], s_tw, lane);
    const int R = FFT_N / M;
    for (int b = lane; b < M; b += 64) sf_accumulate(C, N, M, a, b, s_fft[wave][b * R], acc);
    if (lane == 0) sf_emit_early(C, N, M, a, acc);
    sf_publish_if_last(C, done_ctr);
}

__global__ __launch_bounds__(256) void k_sf_inv2(SfCall C, int N, int M, int AT, const float2* __restrict__ Tm,
                                                 float* __restrict__ acc, unsigned* __restrict__ done_ctr) {
    extern __shared__ float2 sf_sm[];
    float2* bufA = sf_sm;
    float2* bufB = sf_sm + AT * M;
    float2* tw = sf_sm + 2 * AT * M;
    const int a0 = blockIdx.x * AT;
    for (int m = threadIdx.x; m < M / 2; m += 256) tw[m] = sf_cis((unsigned)m, M, +1.f);
    for (int idx = threadIdx.x; idx < AT * M; idx += 256) bufA[idx] = Tm[(size_t)a0 * M + idx];
    __syncthreads();
    const float2* y = sf_fft_lds(bufA, bufB, tw, M, AT);
    for (int idx = threadIdx.x; idx < AT * M; idx += 256) sf_accumulate(C, N, M, a0 + idx / M, idx % M, y[idx], acc);
    if (threadIdx.x < AT) sf_emit_early(C, N, M, a0 + threadIdx.x, acc);
    sf_publish_if_last(C, done_ctr);
}

// ---------------------------------------------------------------------------
// IR preparation (conv.cu:207-253): Z = FFT_N(L + j R) by the four-step forward
// transform, then the two-for-one split.
//   k_sf_ir_cols: for each a: M points over b of z[a + 512 b] -> d, times exp(-2 pi i a d / N), to U[d 512 + a]
//   k_sf_ir_rows: for each d: 512 points over a of U[d 512 + a] -> c, to Z[d + M c]
//   k_sf_ir_unpack: H_L, H_R for bins s < N/2 from Z[s], Z[N - s] (s == 0: the shortcut of conv.cu:53; Q1), stored [d][c]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sf_ir_cols(int N, int M, int AT, const float2* __restrict__ z, float2* __restrict__ U) {
    extern __shared__ float2 sf_sm[];
    float2* bufA = sf_sm;
    float2* bufB = sf_sm + AT * M;
    float2* tw = sf_sm + 2 * AT * M;
    const int a0 = blockIdx.x * AT;
    for (int m = threadIdx.x; m < M / 2; m += 256) tw[m] = sf_cis((unsigned)m, M, -1.f);
    for (int idx = threadIdx.x; idx < AT * M; idx += 256) {
        const int ai = idx % AT, b = idx / AT;
        bufA[ai * M + b] = z[(size_t)a0 + ai + (size_t)FFT_N * b];
    }
    __syncthreads();
    const float2* y = sf_fft_lds(bufA, bufB, tw, M, AT);
    for (int idx = threadIdx.x; idx < AT * M; idx += 256) {
        const int ai = idx % AT, d = idx / AT;
        const float2 val = y[ai * M + d];
        const float2 w = sf_cis(((unsigned)(a0 + ai) * (unsigned)d) & (unsigned)(N - 1), N, -1.f);
        U[(size_t)d * FFT_N + a0 + ai] = make_float2(val.x * w.x - val.y * w.y, val.x * w.y + val.y * w.x);
    }
}

__global__ __launch_bounds__(64 * SF_ROWS) void k_sf_ir_rows(int M, const float2* __restrict__ U, float2* __restrict__ Z,
                                                             const float2* __restrict__ g_tw) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[SF_ROWS][FFT_WAVE_LDS];
    load_twiddles(s_tw, g_tw);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d0 = blockIdx.x * SF_ROWS;
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = U[(size_t)(d0 + wave) * FFT_N + lane + 64 * r];
    fft512_wave<-1, false>(v, s_fft[wave], s_tw, lane);
    __syncthreads();
    for (int idx = threadIdx.x; idx < SF_ROWS * FFT_N; idx += 64 * SF_ROWS) {
        const int di = idx % SF_ROWS, c = idx / SF_ROWS;
        Z[(size_t)d0 + di + (size_t)M * c] = s_fft[di][c];
    }
}

__global__ __launch_bounds__(256) void k_sf_ir_unpack(int N, int M, const float2* __restrict__ Z, float2* __restrict__ Hb) {
    const int p = blockIdx.x * 256 + threadIdx.x;  // [d][c], bin s = d + M c
    const int H = N / 2;
    if (p >= H) return;
    const int s = (p >> 8) + M * (p & 255);
    const float2 va = Z[s];
    float2 vb = va;
    if (s) {
        const float2 t = Z[N - s];
        vb = make_float2(t.x, -t.y);
    }
    const float2 la = make_float2(0.5f * (va.x + vb.x), 0.5f * (va.y + vb.y));
    const float2 dd = make_float2(-0.5f * (va.x - vb.x), -0.5f * (va.y - vb.y));
    Hb[p] = la;
    Hb[(size_t)H + p] = make_float2(-dd.y, dd.x);  // times j
}
