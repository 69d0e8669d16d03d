// kernels.hip.h — the per-block hot path of the reference (conv.cu:287-466),
// restructured as a uniform-partitioned overlap-add pipeline for gfx950.
//
// Data layout in HBM (all fp32; "bin" = one of 256 packed bins of a 512-point
// real-signal spectrum, bin 0 packs {DC, Nyquist} which are both real):
//   IR spectra   H[idx]   : float4 [256 bins][Pstride]   {H_L.re, H_L.im, H_R.re, H_R.im}
//   delay line   FDL      : float4 [256 bins][R slots]   {X_1.re, X_1.im, X_2.re, X_2.im}  (raw spectra)
//   slot gains   gain[v]  : float4 [R slots]             {L<-in1, L<-in2, R<-in1, R<-in2} of voice v
//   MAC output   Y        : float4 [256 bins][Tcap]      {Y_L.re, Y_L.im, Y_R.re, Y_R.im}
//   segments     seg      : float  [SR blocks][2 ch][512] inverse transforms (overlap-add halves)
//   wet ring     wet      : float  [2 ch][WR samples]    indexed by absolute sample mod WR
// Bin-major layouts make the partition sum of one bin a contiguous stream that
// one workgroup owns: no cross-workgroup reduction, coalesced 16-byte lanes.
//
// Gains.  In the reference every scalar that shapes the wet signal — pan
// (conv.cu:386-389), level (:394), wet and the cross-fade state of the live
// IR spectra (f_interpolate, :15-32) — multiplies the WHOLE contribution of the
// input block at which it was current.  Here the delay line keeps raw spectra
// and each slot carries its four path gains; a "voice" is one pair of IRs
// (one per input) with its own gain table.  Normally one voice is active; after
// a select CC the outgoing IR and the incoming IR are two voices whose
// coefficients follow the reference's recurrence exactly.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fft512.hip.h"

#if !defined(MCCONV_LAB) && (defined(G2_ABL) || defined(G2_STAMPS) || defined(G2_ALIAS_TEST) || defined(DF_DBG) || defined(DF_LINEAR) || \
                             defined(TAILP_VGPR_CAP) || defined(IW_LINEAR_TILES) || defined(OS_ABL) || defined(OS_OUT_UNROLL) || defined(MC_JACK_TRACE) || defined(MC_FD_WARM) || defined(MC_TAIL_FFT0))
#error "the measurement builds (timing ablations, time stamps, traces) are part of the lab build only: add -DMCCONV_LAB"
#endif
#define MC_B 256
#define MC_K 512
#define MC_NB 256
#define MC_MAXV 3    // voices (IR pairs) that may sound at once
#define FWD_TILE 8   // blocks per workgroup of the forward / inverse transform kernels (2 per wave)
#define FWD_TILE_LOG2 3
#ifndef XF_WAVES
#define XF_WAVES 8  // waves per workgroup of the forward / inverse transform kernels (FWD_TILE / XF_WAVES blocks each)
#endif
#define XF_THREADS (64 * XF_WAVES)

// fp16 storage of a spectrum entry: four halves {a.re, a.im, b.re, b.im} in 8 bytes
__device__ __forceinline__ uint2 pack_half4(float4 v, float scale) {
    const __half2 lo = __floats2half2_rn(v.x * scale, v.y * scale), hi = __floats2half2_rn(v.z * scale, v.w * scale);
    uint2 r;
    r.x = *reinterpret_cast<const unsigned*>(&lo);
    r.y = *reinterpret_cast<const unsigned*>(&hi);
    return r;
}
__device__ __forceinline__ float4 unpack_half4(uint2 r) {
    const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&r.x));
    const float2 hi = __half22float2(*reinterpret_cast<const __half2*>(&r.y));
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
#define FDL16_SCALE 16.0f  // |X| <= 256 for |x| <= 1: 4096 in half, well inside its range

// Per-block parameters computed on the host in double (cross-fade recurrence, pans, levels)
struct BlockParams {
    double G[MC_MAXV][4];  // wet gain of voice v, path c*2+i, attached to this input block
    float g[MC_MAXV][4];   // the same in float: the slot's entry of the gain tables
    float d[4];            // dry gain c*2+i: dry_i pan_c(panDry_i) level_i      (conv.cu:418-427)
};

// IR sums of the voices' IRs: sig = sum h, alp = sum h (-1)^m; [voice][half][L/R]
struct VoiceSums {
    double sig[MC_MAXV][2][2];
    double alp[MC_MAXV][2][2];
};

// acc += h * x for complex h, x (PACKED: bin 0 packs {DC, Nyquist}: two real products)
template <bool PACKED>
__device__ __forceinline__ void cmac(float2& acc, float hx, float hy, float xx, float xy) {
    if (PACKED) {
        acc.x = fmaf(hx, xx, acc.x);
        acc.y = fmaf(hy, xy, acc.y);
    } else {
        acc.x = fmaf(hx, xx, acc.x);
        acc.x = fmaf(-hy, xy, acc.x);
        acc.y = fmaf(hx, xy, acc.y);
        acc.y = fmaf(hy, xx, acc.y);
    }
}

// Q8 regime, whole batches, when every output block's cut terms are ONE (kappa, partition) term of ONE source block (the reference's
// shipped operating point: predelay 1024, an IR of n_ref - 1024 frames): the wave that has just transformed block t holds all that
// term needs - X_t in registers, the block's gains, one partition of the voices' spectra - and sums the cut terms of output block
// t + shift on the spot (product, inverse transform, slice: what k_drop_fft does, minus its 4 KB read of the delay line per block and
// a launch that is bound by its round trips).  Output blocks whose source lies before the batch stay with k_drop_fft.
struct DropAhead {
    int shift;                   // output block = source block + shift (kappa + predelay / 256 + partition)
    int kappa, c;                // slice of the segment: index 256 kappa + frame - c
    int nv;
    const float4* Ht0[MC_MAXV];  // the term's partition of voice v's IR for input 1, [256 bins] {H_L, H_R} (null: the IR has no such partition)
    const float4* Ht1[MC_MAXV];
    float2* drop;                // [T][256] {L, R} of the batch's output blocks
};

// ---------------------------------------------------------------------------
// K1: forward transform of T input blocks -> delay-line slots (raw spectra).
// Replaces f_pack2R2C + memset + cufftExecC2C + f_unpackC22R (conv.cu:35-73,
// 321-328, 367-371) for one zero-padded 256-frame block per wave pass.  The
// same kernel prepares IR partitions (conv.cu:207-253): the IR's L/R channels
// are the two "inputs", slot = partition index.
// grid = ceil(T / 8), block = 256 (4 waves x 2 transforms each).
// ---------------------------------------------------------------------------
template <bool DA>
__attribute__((amdgpu_waves_per_eu(DA ? 6 : 1, DA ? 6 : 8)))  // DA: 80 registers (7 spilled) keep three workgroups on a CU: 281 against 330 us per 130 000 blocks at 122
__global__ __launch_bounds__(XF_THREADS) void k_fwd(const float* __restrict__ in1, const float* __restrict__ in2,
                                             int in_stride,     // floats between successive frames (1, or 2 for interleaved IR)
                                             int64_t n_frames,  // valid frames in in1/in2 (zero beyond)
                                             int T, float4* __restrict__ fdl, int ring, int slot0,
                                             const BlockParams* __restrict__ ptab, int pstride,
                                             float4* __restrict__ sums,      // [T] raw {S1,S2,A1,A2} or null
                                             float4* __restrict__ slotgain,  // [MC_MAXV][ring] or null
                                             const float2* __restrict__ g_tw,
                                             uint2* __restrict__ fdl16,  // fp16 mirror of the delay line or null
                                             float* __restrict__ xhist, int xr,  // [2][xr] input history ring or null
                                             float4* __restrict__ gring, int rc,  // [MC_MAXV][rc] gains of past blocks
                                             int64_t tabs0,                       // absolute block of t = 0
                                             int need_a0, int need_a1, int need_b0,
                                             int hist_from,  // input history is kept from this block on (see run_front)
                                             int t_base,     // first block of this launch (a block-sliced rank launches its ranges)
                                             DropAhead da,   // DA: the cut terms of output block t + da.shift (see DropAhead)
                                             int store_from = 0) {  // blocks before this one leave no delay-line slot (the overlap-save form keeps the batch's tail only; their cut terms and histories are still produced)
    // Block-sliced engines transform only the blocks some window of theirs can reach: t in [need_a0, need_a1) or
    // t >= need_b0 (the tail the next call reaches back to).  The others get zero Q1/Q2 sums and nothing else.
    __shared__ float2 s_tw[FFT_N];
    // eight wave-private transform buffers, then (every wave has its spectra in registers) the transposed tile
    // [256 bins][8 blocks + 1] of float4 in the same memory: 41 KB of LDS, three workgroups per CU
    __shared__ __align__(16) float2 s_mem[XF_WAVES * FFT_WAVE_LDS];
    static_assert(XF_WAVES == FWD_TILE, "one wave per block of the tile");
    static_assert(sizeof(float2) * XF_WAVES * FFT_WAVE_LDS >= sizeof(float4) * MC_NB * (FWD_TILE + 1), "tile fits the transform buffers");
    float4(*s_tile)[FWD_TILE + 1] = reinterpret_cast<float4(*)[FWD_TILE + 1]>(s_mem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tb0 = t_base + blockIdx.x * FWD_TILE;
    auto needed = [&](int t) { return (t >= need_a0 && t < need_a1) || t >= need_b0; };
    const int tb = wave;  // block within tile
    const int t = tb0 + tb;
    const bool active = t < T && needed(t);  // wave-uniform
    if (t < T && !active && sums && lane == 0) sums[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        // a tile none of whose blocks is needed (most tiles of a block-sliced rank) ends here, before any barrier
        const int lo = tb0, hi = min(tb0 + FWD_TILE, T);  // [lo, hi) against [need_a0, need_a1) and [need_b0, T)
        if (!((lo < need_a1 && hi > need_a0) || hi > need_b0)) return;
    }
    load_twiddles(s_tw, g_tw);
    __syncthreads();
    float2* lds = s_mem + wave * FFT_WAVE_LDS;
    const bool vec_in = in_stride == 1 && ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2)) & 15) == 0;
    float4 xs[4];
    if (active) {
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = make_float2(0.f, 0.f);
        if (vec_in && (int64_t)(t + 1) * MC_B <= n_frames) {
            // whole block of contiguous frames: 16 bytes per lane and access, re-striped through the wave's LDS
            const int64_t f = (int64_t)t * MC_B + 4 * lane;
            const float4 a = *reinterpret_cast<const float4*>(in1 + f), b = *reinterpret_cast<const float4*>(in2 + f);
            if (xhist && t >= hist_from) {  // input history for the Q8 pass of later calls
                const size_t at = (size_t)((tabs0 * MC_B + f) & (xr - 1));
                *reinterpret_cast<float4*>(xhist + at) = a;
                *reinterpret_cast<float4*>(xhist + xr + at) = b;
            }
            lds[4 * lane] = make_float2(a.x, b.x);
            lds[4 * lane + 1] = make_float2(a.y, b.y);
            lds[4 * lane + 2] = make_float2(a.z, b.z);
            lds[4 * lane + 3] = make_float2(a.w, b.w);
            fft_sync<false>();
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = lds[lane + 64 * r];
            fft_sync<false>();
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {  // n = lane + 64 r < 256: the block; 256..511 stay zero
                int64_t f = (int64_t)t * MC_B + lane + 64 * r;
                if (f < n_frames) v[r] = make_float2(in1[f * in_stride], in2[f * in_stride]);
                if (xhist && t >= hist_from) {  // input history for the Q8 pass of later calls
                    const int64_t tau = tabs0 * MC_B + f;
                    xhist[(size_t)(tau & (xr - 1))] = v[r].x;
                    xhist[(size_t)xr + (tau & (xr - 1))] = v[r].y;
                }
            }
        }
        fft512_wave<-1, false>(v, lds, s_tw, lane);  // wave-private LDS: waves that skip a block stay out of it
        // two-for-one split of the packed transform (true spectra; the
        // reference's DC/Nyquist shortcuts Q1/Q2 are rank-1 terms added in k_post)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int k = lane + 64 * j;
            float2 za = lds[k], zb = lds[(FFT_N - k) & (FFT_N - 1)];
            float2 x1, x2;
            if (k == 0) {
                float2 zn = lds[MC_B];
                x1 = make_float2(za.x, zn.x);  // {DC, Nyquist} of input 1
                x2 = make_float2(za.y, zn.y);  // {DC, Nyquist} of input 2
                if (sums) sums[t] = make_float4(za.x, za.y, zn.x, zn.y);  // S1, S2, A1, A2
            } else {
                x1 = make_float2(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y));
                x2 = make_float2(0.5f * (za.y + zb.y), -0.5f * (za.x - zb.x));
            }
            xs[j] = make_float4(x1.x, x1.y, x2.x, x2.y);
        }
        if (slotgain && lane < MC_MAXV) {
            const float* gv = ptab[(int64_t)t * pstride].g[lane];
            slotgain[(size_t)lane * ring + ((slot0 + t) & (ring - 1))] = make_float4(gv[0], gv[1], gv[2], gv[3]);
            if (gring) gring[(size_t)lane * rc + (size_t)((tabs0 + t) & (rc - 1))] = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
        if (DA && t + da.shift < T) {  // (wave-uniform) the arithmetic of k_drop_fft's single term, in its order
            const BlockParams& bp = ptab[(int64_t)t * pstride];
            float4 y[4];
#pragma unroll
            for (int j = 0; j < 4; j++) y[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int vi = 0; vi < MC_MAXV; vi++) {
                if (vi >= da.nv || (!da.Ht0[vi] && !da.Ht1[vi])) continue;
                const float4* __restrict__ B0 = da.Ht0[vi] ? da.Ht0[vi] : da.Ht1[vi];
                const float4* __restrict__ B1 = da.Ht1[vi] ? da.Ht1[vi] : da.Ht0[vi];
                float4 g = make_float4(bp.g[vi][0], bp.g[vi][1], bp.g[vi][2], bp.g[vi][3]);
                g.x = da.Ht0[vi] ? g.x : 0.f, g.z = da.Ht0[vi] ? g.z : 0.f;
                g.y = da.Ht1[vi] ? g.y : 0.f, g.w = da.Ht1[vi] ? g.w : 0.f;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = lane + 64 * j;
                    const float4 h0 = B0[k], h1 = B1[k], x = xs[j];
                    float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
                    cmac<false>(a0, h0.x, h0.y, x.x, x.y);
                    cmac<false>(a1, h1.x, h1.y, x.z, x.w);
                    cmac<false>(a2, h0.z, h0.w, x.x, x.y);
                    cmac<false>(a3, h1.z, h1.w, x.z, x.w);
                    if (j == 0 && lane == 0) {  // bin 0 packs {DC, Nyquist}: two real products
                        a0 = make_float2(h0.x * x.x, h0.y * x.y);
                        a1 = make_float2(h1.x * x.z, h1.y * x.w);
                        a2 = make_float2(h0.z * x.x, h0.w * x.y);
                        a3 = make_float2(h1.z * x.z, h1.w * x.w);
                    }
                    y[j].x += g.x * a0.x + g.y * a1.x;
                    y[j].y += g.x * a0.y + g.y * a1.y;
                    y[j].z += g.z * a2.x + g.w * a3.x;
                    y[j].w += g.z * a2.y + g.w * a3.y;
                }
            }
            float4* ybin = reinterpret_cast<float4*>(lds);  // [256] {Y_L, Y_R} per bin (the forward transform's buffer: every lane has read its spectrum)
            fft_sync<false>();
#pragma unroll
            for (int j = 0; j < 4; j++) ybin[lane + 64 * j] = y[j];
            fft_sync<false>();
            float2 w8[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {  // Hermitian extension of the packed spectrum Y_L + j Y_R (as k_inv)
                const int n = lane + 64 * r;
                float2 w;
                if (n == 0) {
                    const float4 yy = ybin[0];
                    w = make_float2(yy.x, yy.z);
                } else if (n == MC_B) {
                    const float4 yy = ybin[0];
                    w = make_float2(yy.y, yy.w);
                } else if (n < MC_B) {
                    const float4 yy = ybin[n];
                    w = make_float2(yy.x - yy.w, yy.y + yy.z);
                } else {
                    const float4 yy = ybin[FFT_N - n];
                    w = make_float2(yy.x + yy.w, -yy.y + yy.z);
                }
                w8[r] = w;
            }
            fft_sync<false>();
            fft512_wave<+1, false>(w8, lds, s_tw, lane);
            const float sc = 1.0f / FFT_N;
            float dl[4], dr[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 256 * da.kappa + 4 * lane + q - da.c;  // segment index of frame 4 lane + q
                const bool in = i >= 0 && i < FFT_N;
                const float2 z = lds[in ? i : 0];
                dl[q] = in ? z.x * sc : 0.f;
                dr[q] = in ? z.y * sc : 0.f;
            }
            float4* dst = reinterpret_cast<float4*>(da.drop + (size_t)(t + da.shift) * MC_B + 4 * lane);
            dst[0] = make_float4(dl[0], dr[0], dl[1], dr[1]);
            dst[1] = make_float4(dl[2], dr[2], dl[3], dr[3]);
        }
    }
    __syncthreads();  // every wave has read its transform: the buffers become the tile
    if (active) {
#pragma unroll
        for (int j = 0; j < 4; j++) s_tile[lane + 64 * j][tb] = xs[j];
    }
    __syncthreads();
    // transposed, coalesced store: 8 consecutive slots (128 B) per bin
    for (int idx = threadIdx.x; idx < MC_NB * FWD_TILE; idx += XF_THREADS) {
        int tb = idx & (FWD_TILE - 1), k = idx >> FWD_TILE_LOG2;
        int t = tb0 + tb;
        if (t < T && needed(t) && t >= store_from) {
            const size_t at = (size_t)k * ring + ((slot0 + t) & (ring - 1));
            fdl[at] = s_tile[k][tb];
            if (fdl16) fdl16[at] = pack_half4(s_tile[k][tb], FDL16_SCALE);
        }
    }
}

// Fast-FIR components.  Level L splits a sequence into 3^L sub-sequences at 1/2^L of the rate; component c has
// base-3 digits (d_1 .. d_L, most significant first), digit 0 = even samples, 1 = odd samples, 2 = their sum.
// In terms of the original sequence a component is the sum over a set of residues j modulo 2^L:
//   ffa_mask(c, L) bit j set  <=>  x_c[m] includes X[2^L m + j].
__host__ __device__ __forceinline__ int ffa_digit_mask(int d) { return d == 0 ? 1 : (d == 1 ? 2 : 3); }
__host__ __device__ __forceinline__ int ffa_mask(int c, int lvl) {
    // digit l (l = 1 most significant) selects residues j_l in {0,1} at stride 2^l: j = sum_l 2^(l-1) j_l
    int mask = 1, span = 1;  // residues selected so far, modulo `span`
    int div = lvl == 1 ? 1 : (lvl == 2 ? 3 : 9);
    for (int l = 0; l < lvl; l++) {
        const int dm = ffa_digit_mask((c / div) % 3);
        int m = 0;
        if (dm & 1) m |= mask;
        if (dm & 2) m |= mask << span;
        mask = m;
        span *= 2;
        div /= 3;
    }
    return mask;
}

// Components of an IR's partition sequence (load time): Hp[c][bin][q] = sum_{j in mask(c)} H[bin][2^L q + j],
// 3^L arrays of pstride / 2^L partitions each.
__global__ __launch_bounds__(256) void k_polyphase(const float4* __restrict__ H, float4* __restrict__ Hp, int pstride, int lvl) {
    const int ph = pstride >> lvl, nc = lvl == 1 ? 3 : (lvl == 2 ? 9 : 27);
    const size_t n = (size_t)MC_NB * ph;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n * nc; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / n);
        const size_t r = i - (size_t)c * n, bin = r / ph, q = r - bin * ph;
        const int mask = ffa_mask(c, lvl);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < (1 << lvl); j++)
            if ((mask >> j) & 1) {
                const float4 h = H[bin * pstride + (q << lvl) + j];
                a.x += h.x;
                a.y += h.y;
                a.z += h.z;
                a.w += h.w;
            }
        Hp[i] = a;
    }
}

// fp32 -> scaled fp16 copy of an IR's spectra (load time)
__global__ __launch_bounds__(256) void k_to_half(const float4* __restrict__ src, uint2* __restrict__ dst, size_t n, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = pack_half4(src[i], scale);
}

// Linear combination of up to MC_MAXV IRs (spectra as float4, taps as float2 viewed as float): dst = sum_j c_j src_j,
// each source read only below its own length.  Used when more IRs are cross-fading than there are voices: the
// reference's live spectrum is sum_j c_j H_j, and every deselected c_j decays by the same factor per block
// (f_interpolate, conv.cu:27), so the deselected IRs can be merged into one.
struct MixSrc {
    const float* p[MC_MAXV];
    size_t n[MC_MAXV];  // floats valid in p[j]
    float c[MC_MAXV];
};
__global__ __launch_bounds__(256) void k_mix(float* __restrict__ dst, size_t n, MixSrc src) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < MC_MAXV; j++)
            if (src.p[j] && i < src.n[j]) a = fmaf(src.c[j], src.p[j][i], a);
        dst[i] = a;
    }
}

// ---------------------------------------------------------------------------
// complex multiply-accumulate helpers.  PACKED handles bin 0, whose float2
// holds two independent real bins {DC, Nyquist}.
// (replaces f_pointwiseMultiplyAndScale, conv.cu:102-123, with the true product; Q3)
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// K2 (batch): partition x bin complex MAC with the IR held on chip.
// One workgroup = one bin x 256 consecutive output blocks.  Lane j owns output
// blocks 4j..4j+3; the IR spectra of the bin are wave-uniform and arrive as
// scalar loads (SGPR operands of the FMAs), the delay-line window of the tile
// sits in LDS de-interleaved by 4 so that every ds_read_b128 of the sliding
// window is conflict-free, and each window entry is reused for 4 outputs from
// registers.  The 4 waves split the partition range; partial sums meet in LDS.
// grid = 256 bins x ceil(T/256); block ids with equal bin share an XCD (id % 8).
//
// SLOTGAIN = false: every slot in the window carries the same four gains (the
//   steady state): the window holds raw spectra, four path sums are kept and the
//   gains are applied once at the end.
// SLOTGAIN = true : gains differ between slots (cold-start ramp, a parameter
//   or IR change within the last P blocks): the window is filled with the
//   per-path products gain(slot) * X, two windows {L paths, R paths}.
// ---------------------------------------------------------------------------
#define MAC_PSEG 1024                       // partitions per LDS window segment
#define MAC_WQ ((MAC_PSEG + 256 + 16) / 4)  // quarter-window length (entries), incl. prefetch slack

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One wave's share of a window segment: partitions pse-1-q for q in [q_lo, q_hi).
template <bool PACKED>
__device__ __forceinline__ void mac_sweep(const float4* __restrict__ H0k, const float4* __restrict__ H1k, int pse,
                                          int q_lo, int q_hi, const float4* s_win, int lane, float2 (&acc)[4][4]) {
    // window registers: w[0..6] = entries 4*lane + q + 0..6 (w[7] is the next iteration's w[3])
    float4 w[8], wn[4];
    const float4* wbase = s_win + lane + (q_lo >> 2);
#pragma unroll
    for (int u = 0; u < 4; u++) w[u] = wbase[u * MAC_WQ];
#pragma unroll
    for (int u = 0; u < 4; u++) w[4 + u] = wbase[u * MAC_WQ + 1];
    // partitions pb..pb+3 of both IRs as one aligned 64-byte scalar load each
    f32x16 hc0 = *reinterpret_cast<const f32x16*>(H0k + (pse - 4 - q_lo));
    f32x16 hc1 = *reinterpret_cast<const f32x16*>(H1k + (pse - 4 - q_lo));
    for (int q = q_lo; q < q_hi; q += 4) {
        const int qn = min(q + 4, q_hi - 4);
        const f32x16 hn0 = *reinterpret_cast<const f32x16*>(H0k + (pse - 4 - qn));
        const f32x16 hn1 = *reinterpret_cast<const f32x16*>(H1k + (pse - 4 - qn));
        const float4* wp = s_win + lane + (q >> 2) + 2;
#pragma unroll
        for (int u = 0; u < 4; u++) wn[u] = wp[u * MAC_WQ];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // step u uses partition pb + 3 - u = floats [4*(3-u) .. 4*(3-u)+3] of the 16
            const int o = 4 * (3 - u);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 x = w[u + r];
                cmac<PACKED>(acc[r][0], hc0[o + 0], hc0[o + 1], x.x, x.y);  // L <- in1 * h_sel0,L
                cmac<PACKED>(acc[r][1], hc1[o + 0], hc1[o + 1], x.z, x.w);  // L <- in2 * h_sel1,L
                cmac<PACKED>(acc[r][2], hc0[o + 2], hc0[o + 3], x.x, x.y);  // R <- in1 * h_sel0,R
                cmac<PACKED>(acc[r][3], hc1[o + 2], hc1[o + 3], x.z, x.w);  // R <- in2 * h_sel1,R
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            w[u] = w[4 + u];
            w[4 + u] = wn[u];
        }
        hc0 = hn0;
        hc1 = hn1;
    }
}

// the same sweep over gain-scaled windows: wl = {gL0 X1, gL1 X2}, wr = {gR0 X1, gR1 X2}
template <bool PACKED>
__device__ __forceinline__ void mac_sweep_g(const float4* __restrict__ H0k, const float4* __restrict__ H1k, int pse,
                                            int q_lo, int q_hi, const float4* s_wl, const float4* s_wr, int lane,
                                            float2 (&acc)[4][2]) {
    float4 wl[8], wr[8];
    const int b0 = lane + (q_lo >> 2);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        wl[u] = s_wl[b0 + u * MAC_WQ];
        wr[u] = s_wr[b0 + u * MAC_WQ];
        wl[4 + u] = s_wl[b0 + u * MAC_WQ + 1];
        wr[4 + u] = s_wr[b0 + u * MAC_WQ + 1];
    }
    for (int q = q_lo; q < q_hi; q += 4) {
        const f32x16 hc0 = *reinterpret_cast<const f32x16*>(H0k + (pse - 4 - q));
        const f32x16 hc1 = *reinterpret_cast<const f32x16*>(H1k + (pse - 4 - q));
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int o = 4 * (3 - u);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 xl = wl[u + r], xr = wr[u + r];
                cmac<PACKED>(acc[r][0], hc0[o + 0], hc0[o + 1], xl.x, xl.y);
                cmac<PACKED>(acc[r][0], hc1[o + 0], hc1[o + 1], xl.z, xl.w);
                cmac<PACKED>(acc[r][1], hc0[o + 2], hc0[o + 3], xr.x, xr.y);
                cmac<PACKED>(acc[r][1], hc1[o + 2], hc1[o + 3], xr.z, xr.w);
            }
        }
        const int bn = lane + (q >> 2) + 2;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            wl[u] = wl[4 + u];
            wr[u] = wr[4 + u];
            wl[4 + u] = s_wl[bn + u * MAC_WQ];
            wr[4 + u] = s_wr[bn + u * MAC_WQ];
        }
    }
}

template <bool SLOTGAIN>
__global__ __launch_bounds__(256) void k_mac_resident(const float4* __restrict__ H0, const float4* __restrict__ H1,
                                                      int pstride_ir,          // slots per bin of the IR arrays
                                                      int p_begin, int p_end,  // partition range, multiples of 16
                                                      const float4* __restrict__ fdl, int ring, int slot0, int T,
                                                      float4 ugain,                         // SLOTGAIN = false
                                                      const float4* __restrict__ slotgain,  // SLOTGAIN = true: [ring]
                                                      float4* __restrict__ Y, int tcap, int accumulate,
                                                      int psplit, int pchunk, int lvl, int tiles, int64_t yplane) {
    // grid = 256 bins x tiles x psplit: short batches split the partition range over `psplit` workgroups
    // (planes of Y summed by k_inv) so that the launch still fills the chip.
    // The kernel is a convolution along the block axis, out[n] = sum_q H[q] x[n - q].  lvl = 0, the direct form:
    // x = the delay line from slot0, H = the IR's partitions, outputs n = 0 .. T-1.  lvl = 1, 2: the fast-FIR form
    // (launch_mac_batch) - the grid also spans the 3^lvl components; component c convolves
    // x_c[m] = sum_{j in mask(c)} X[slot0 + 2^lvl m + j] with the matching component of the IR (H0/H1 point to the
    // component arrays, `pstride_ir` partitions each) for outputs n = -1 .. T-2, into plane c of Y.
    __shared__ float4 s_win[(SLOTGAIN ? 8 : 4) * MAC_WQ];
    const int bin = blockIdx.x & (MC_NB - 1);
    int rest = blockIdx.x >> 8;
    const int comp = rest / (tiles * psplit);  // 0 in the direct form
    rest -= comp * tiles * psplit;
    const int tile = rest / psplit, split = rest - tile * psplit;
    const int t0 = tile * 256;  // first output of the tile, relative to n_first
    const int xstride = 1 << lvl, xmask = lvl ? ffa_mask(comp, lvl) : 1, n_first = lvl ? -1 : 0;
    p_begin += split * pchunk;
    p_end = min(p_end, p_begin + pchunk);
    Y += (size_t)split * MC_NB * tcap + (size_t)comp * yplane;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float4* H0k = H0 + ((size_t)comp * MC_NB + bin) * pstride_ir;
    const float4* H1k = H1 + ((size_t)comp * MC_NB + bin) * pstride_ir;
    const float4* fk = fdl + (size_t)bin * ring;

    float2 acc[4][4];   // SLOTGAIN = false: four path sums per output
    float2 accg[4][2];  // SLOTGAIN = true : Y_L, Y_R per output
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int c = 0; c < 4; c++) acc[r][c] = make_float2(0.f, 0.f);
        accg[r][0] = accg[r][1] = make_float2(0.f, 0.f);
    }

    for (int ps = p_begin; ps < p_end; ps += MAC_PSEG) {
        const int pse = min(ps + MAC_PSEG, p_end);
        const int seg = pse - ps;  // multiple of 16
        // window entry e <-> block (t0 - pse + 1 + e), e in [0, seg + 255]
        const int nwin = seg + 256;
        const int mbase = n_first + t0 - pse + 1;  // sequence index of window entry 0
        __syncthreads();  // previous segment's readers are done
        for (int e = threadIdx.x; e < nwin + 12; e += 256) {
            const int sb = slot0 + xstride * (mbase + e);
            const int pos = (e & 3) * MAC_WQ + (e >> 2);
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j < xstride && ((xmask >> j) & 1) && e < nwin) {
                    const int slot = (sb + j) & (ring - 1);
                    const float4 x = fk[slot];
                    if (SLOTGAIN) {
                        const float4 g = slotgain[slot];
                        a = make_float4(a.x + g.x * x.x, a.y + g.x * x.y, a.z + g.y * x.z, a.w + g.y * x.w);
                        b = make_float4(b.x + g.z * x.x, b.y + g.z * x.y, b.z + g.w * x.z, b.w + g.w * x.w);
                    } else {
                        a = make_float4(a.x + x.x, a.y + x.y, a.z + x.z, a.w + x.w);
                    }
                }
            }
            s_win[pos] = a;
            if (SLOTGAIN) s_win[4 * MAC_WQ + pos] = b;
        }
        __syncthreads();
        const int per = seg >> 2;  // partitions per wave, multiple of 4
        const int q_lo = wave * per, q_hi = q_lo + per;
        if (SLOTGAIN) {
            if (bin == 0)
                mac_sweep_g<true>(H0k, H1k, pse, q_lo, q_hi, s_win, s_win + 4 * MAC_WQ, lane, accg);
            else
                mac_sweep_g<false>(H0k, H1k, pse, q_lo, q_hi, s_win, s_win + 4 * MAC_WQ, lane, accg);
        } else {
            if (bin == 0)
                mac_sweep<true>(H0k, H1k, pse, q_lo, q_hi, s_win, lane, acc);
            else
                mac_sweep<false>(H0k, H1k, pse, q_lo, q_hi, s_win, lane, acc);
        }
    }
    // sum the 4 waves' partials in LDS (uniform gains applied here)
    __syncthreads();
    float* red = reinterpret_cast<float*>(s_win);  // [wave][16][64]
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float2 yl, yr;
        if (SLOTGAIN) {
            yl = accg[r][0];
            yr = accg[r][1];
        } else {
            yl = make_float2(ugain.x * acc[r][0].x + ugain.y * acc[r][1].x, ugain.x * acc[r][0].y + ugain.y * acc[r][1].y);
            yr = make_float2(ugain.z * acc[r][2].x + ugain.w * acc[r][3].x, ugain.z * acc[r][2].y + ugain.w * acc[r][3].y);
        }
        red[(wave * 16 + r * 4 + 0) * 64 + lane] = yl.x;
        red[(wave * 16 + r * 4 + 1) * 64 + lane] = yl.y;
        red[(wave * 16 + r * 4 + 2) * 64 + lane] = yr.x;
        red[(wave * 16 + r * 4 + 3) * 64 + lane] = yr.y;
    }
    __syncthreads();
    {
        const int o = threadIdx.x;  // output block within the tile: o = 4 j + r
        const int j = o >> 2, r = o & 3;
        float f[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; w++) s += red[(w * 16 + r * 4 + c) * 64 + j];
            f[c] = s;
        }
        if (t0 + o < T) {
            float4* dst = Y + (size_t)bin * tcap + t0 + o;
            if (accumulate) {  // a further voice adds to what the first one wrote
                const float4 y = *dst;
                f[0] += y.x;
                f[1] += y.y;
                f[2] += y.z;
                f[3] += y.w;
            }
            *dst = make_float4(f[0], f[1], f[2], f[3]);
        }
    }
}

// ---------------------------------------------------------------------------
// K2 (stream): the same sum as a bandwidth-bound reduction, for single blocks
// (the JACK path) and short batches.  Lanes = partitions: each lane loads 16 B
// of each IR and 16 B of the delay line per partition (coalesced) plus the
// slot's gains (or one uniform set).  All loads of STREAM_U partitions are
// issued before the first use.  grid = (256 bins, nchunk, T); partial sums per
// chunk (and voice) are added by k_inv / k_tail1.
// ---------------------------------------------------------------------------
#define STREAM_U 4  // partitions per lane and loop trip

// HALF: spectra and delay line are read as scaled half4 (8 B per entry, half the bytes), products and sums
// stay fp32; `inv` = 1 / (scale of IR 0 * FDL scale), 1 / (scale of IR 1 * FDL scale) undoes the scaling.
template <bool UNIFORM, int NT, bool HALF>
__device__ __forceinline__ void mac_stream_body(const int bin, const int ch, const int t, const void* __restrict__ H0v,
                                                const void* __restrict__ H1v, int pstride_ir, int p_begin, int p_end, int chunk,
                                                const void* __restrict__ fdlv, const float4* __restrict__ slotgain, int ring,
                                                int slot0, float4* __restrict__ part, int nsum, int ch_off, float4 ugain, float2 inv) {
    typedef typename std::conditional<HALF, uint2, float4>::type ST;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const ST* H0k = reinterpret_cast<const ST*>(H0v) + (size_t)bin * pstride_ir;
    const ST* H1k = reinterpret_cast<const ST*>(H1v) + (size_t)bin * pstride_ir;
    const ST* fk = reinterpret_cast<const ST*>(fdlv) + (size_t)bin * ring;
    const int lo = p_begin + ch * chunk, hi = min(lo + chunk, p_end);
    const int st = slot0 + t;
    float2 yl = make_float2(0.f, 0.f), yr = make_float2(0.f, 0.f);
    const bool packed = (bin == 0);
    for (int p0 = lo + (int)threadIdx.x; p0 < hi; p0 += NT * STREAM_U) {
        float4 x[STREAM_U], g[STREAM_U], h0[STREAM_U], h1[STREAM_U];
        ST xs[STREAM_U], h0s[STREAM_U], h1s[STREAM_U];
#pragma unroll
        for (int u = 0; u < STREAM_U; u++) {
            const int p = min(p0 + NT * u, hi - 1);  // clamped: the load is always in range, masked below
            const int slot = (st - p) & (ring - 1);
            xs[u] = fk[slot];
            g[u] = ugain;
            if (!UNIFORM) g[u] = slotgain[slot];
            h0s[u] = H0k[p];
            h1s[u] = H1k[p];
        }
#pragma unroll
        for (int u = 0; u < STREAM_U; u++) {
            if constexpr (HALF) {
                x[u] = unpack_half4(xs[u]);
                h0[u] = unpack_half4(h0s[u]);
                h1[u] = unpack_half4(h1s[u]);
                g[u] = make_float4(g[u].x * inv.x, g[u].y * inv.y, g[u].z * inv.x, g[u].w * inv.y);
            } else {
                x[u] = xs[u];
                h0[u] = h0s[u];
                h1[u] = h1s[u];
            }
        }
#pragma unroll
        for (int u = 0; u < STREAM_U; u++) {
            if (p0 + NT * u < hi) {
                float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
                if (packed) {
                    cmac<true>(a0, h0[u].x, h0[u].y, x[u].x, x[u].y);
                    cmac<true>(a1, h1[u].x, h1[u].y, x[u].z, x[u].w);
                    cmac<true>(a2, h0[u].z, h0[u].w, x[u].x, x[u].y);
                    cmac<true>(a3, h1[u].z, h1[u].w, x[u].z, x[u].w);
                } else {
                    cmac<false>(a0, h0[u].x, h0[u].y, x[u].x, x[u].y);
                    cmac<false>(a1, h1[u].x, h1[u].y, x[u].z, x[u].w);
                    cmac<false>(a2, h0[u].z, h0[u].w, x[u].x, x[u].y);
                    cmac<false>(a3, h1[u].z, h1[u].w, x[u].z, x[u].w);
                }
                yl.x += g[u].x * a0.x + g[u].y * a1.x;
                yl.y += g[u].x * a0.y + g[u].y * a1.y;
                yr.x += g[u].z * a2.x + g[u].w * a3.x;
                yr.y += g[u].z * a2.y + g[u].w * a3.y;
            }
        }
    }
    // wavefront butterfly reduction (64 lanes), then the waves through LDS
    float v[4] = {yl.x, yl.y, yr.x, yr.y};
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_xor(v[c], off, 64);
    }
    __shared__ float s_red[NT / 64][4];
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 4; c++) s_red[wave][c] = v[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NT / 64; w++) {
            o.x += s_red[w][0];
            o.y += s_red[w][1];
            o.z += s_red[w][2];
            o.w += s_red[w][3];
        }
        part[((size_t)t * MC_NB + bin) * nsum + ch_off + ch] = o;
    }
}

#define MC_STAMP_WGS 2048  // workgroups of a launch that leave time stamps (k_mac_stream)
template <bool UNIFORM, int NT, bool HALF>
__global__ __launch_bounds__(NT) void k_mac_stream(const void* __restrict__ H0v, const void* __restrict__ H1v,
                                                   int pstride_ir, int p_begin, int p_end, int chunk,
                                                   const void* __restrict__ fdlv, const float4* __restrict__ slotgain,
                                                   int ring, int slot0, float4* __restrict__ part, int nsum, int ch_off,
                                                   float4 ugain, float2 inv, unsigned long long* __restrict__ stamps) {
    // stamps != null (kernel timing of a single period's sweep, whose few microseconds HIP events cannot bracket): every
    // workgroup leaves {start, end} in ticks of the 100 MHz counter in its own slot (no atomics: 512 of them on one
    // address cost more than the kernel); the host takes the earliest start and the latest end
    const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned long long t_start = 0;
    if (stamps) t_start = __builtin_amdgcn_s_memrealtime();
    mac_stream_body<UNIFORM, NT, HALF>((int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, H0v, H1v, pstride_ir, p_begin, p_end, chunk, fdlv,
                                       slotgain, ring, slot0, part, nsum, ch_off, ugain, inv);
    if (stamps && wg < MC_STAMP_WGS) {
        __syncthreads();  // (every wave's stores have been issued and acknowledged)
        if (threadIdx.x == 0) {
            stamps[2 * wg] = t_start;
            stamps[2 * wg + 1] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// ---------------------------------------------------------------------------
// Fast-FIR form: combine the 3^lvl component sequences the resident kernel wrote (each `plane` elements apart,
// [bin][tcap], stored from sequence index -1: Z[n] sits at n + 1) into the partition sums of blocks 0 .. T-1,
// Yc[bin][t].  One level, with A = even, B = odd, C = sum component:
//   Y[2n] = A[n] + B[n-1],   Y[2n+1] = C[n] - A[n] - B[n];
// with more levels A, B, C are themselves combined the same way from their own three components (component index
// = base-3 digits, most significant = outermost split).
// One thread produces the 2^lvl blocks of one group g from the 3^lvl components at indices g - 1 and g.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_cab(float4 c, float4 a, float4 b) {
    return make_float4(c.x - a.x - b.x, c.y - a.y - b.y, c.z - a.z - b.z, c.w - a.w - b.w);
}

// out[i] = value at index 2^LVL g - 1 + i (i = 0 .. 2^LVL) of the sequence whose 3^LVL stored components start at
// plane c0
template <int LVL>
__device__ __forceinline__ void ffa_expand(const float4* __restrict__ base, int64_t plane, int c0, int g, float4* out) {
    if constexpr (LVL == 0) {
        out[0] = base[(int64_t)c0 * plane + g - 1];
        out[1] = base[(int64_t)c0 * plane + g];
    } else {
        constexpr int M = 1 << (LVL - 1);
        constexpr int W = LVL == 1 ? 1 : (LVL == 2 ? 3 : 9);  // planes per sub-sequence
        float4 a[M + 1], b[M + 1], c[M + 1];
        ffa_expand<LVL - 1>(base, plane, c0, g, a);
        ffa_expand<LVL - 1>(base, plane, c0 + W, g, b);
        ffa_expand<LVL - 1>(base, plane, c0 + 2 * W, g, c);
        out[0] = f4_cab(c[0], a[0], b[0]);
#pragma unroll
        for (int u = 0; u < M; u++) {
            out[1 + 2 * u] = f4_add(a[u + 1], b[u]);
            out[2 + 2 * u] = f4_cab(c[u + 1], a[u + 1], b[u + 1]);
        }
    }
}

template <int LVL>
__global__ __launch_bounds__(256) void k_ffa_combine(const float4* __restrict__ Yp, int64_t plane, int tcap, int T,
                                                     float4* __restrict__ Yc, int ycap) {
    // grid = (ceil(T / 2^LVL / 256), 256 bins): consecutive threads = consecutive groups of one bin
    constexpr int S = 1 << LVL;
    const int g = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
    if (g * S >= T) return;
    const float4* base = Yp + (int64_t)k * tcap + 1;  // + 1: sequences start at index -1
    float4 out[S + 1];
    ffa_expand<LVL>(base, plane, 0, g, out);
    float4* dst = Yc + (size_t)k * ycap + (size_t)g * S;
#pragma unroll
    for (int r = 0; r < S; r++)
        if (g * S + r < T) dst[r] = out[1 + r];
}

// ---------------------------------------------------------------------------
// K3: packed inverse transform.  W = Y_L + j Y_R (Hermitian-extended), one
// 512-point inverse per block, real part = left segment, imaginary = right
// (replaces the two cufftExecC2C inverse calls, conv.cu:403-408).
// Y element (bin k, block t) = sum_{c<nsum} Ysrc[k*sk + t*st + c*sc].
// seg[(seg0 + t) mod sr][ch][512].   grid = ceil(T/8), block = 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(XF_THREADS) void k_inv(const float4* __restrict__ Ysrc, int64_t sk, int64_t st, int nsum, int64_t sc,
                                             int T, float* __restrict__ seg, int sr, int seg0,
                                             const float2* __restrict__ g_tw) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ float2 s_fft[XF_WAVES][FFT_WAVE_LDS];
    __shared__ float4 s_tile[MC_NB][FWD_TILE + 1];
    load_twiddles(s_tw, g_tw);
    const int tb0 = blockIdx.x * FWD_TILE;
    {
        constexpr int NR = MC_NB * FWD_TILE / XF_THREADS, KS = XF_THREADS / FWD_TILE;  // entries per thread, bins between them
        const int tb = threadIdx.x & (FWD_TILE - 1), k0 = threadIdx.x >> FWD_TILE_LOG2;
        const int t = tb0 + tb;
        const float4* src = Ysrc + (int64_t)k0 * sk + (int64_t)t * st;
        float4 y[NR];
#pragma unroll
        for (int r = 0; r < NR; r++)  // the first (usually only) term of all of the thread's bins in flight together
            y[r] = t < T ? src[(int64_t)KS * r * sk] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T)
            for (int c = 1; c < nsum; c++) {
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const float4 a = src[(int64_t)KS * r * sk + c * sc];
                    y[r].x += a.x;
                    y[r].y += a.y;
                    y[r].z += a.z;
                    y[r].w += a.w;
                }
            }
#pragma unroll
        for (int r = 0; r < NR; r++) s_tile[k0 + KS * r][tb] = y[r];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2* lds = s_fft[wave];
    for (int it = 0; it < FWD_TILE / XF_WAVES; it++) {
        const int tb = it * XF_WAVES + wave;
        const int t = tb0 + tb;
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int n = lane + 64 * r;
            float2 w;
            if (n == 0) {
                float4 y = s_tile[0][tb];
                w = make_float2(y.x, y.z);  // DC of L, R
            } else if (n == MC_B) {
                float4 y = s_tile[0][tb];
                w = make_float2(y.y, y.w);  // Nyquist of L, R
            } else if (n < MC_B) {
                float4 y = s_tile[n][tb];
                w = make_float2(y.x - y.w, y.y + y.z);  // Y_L + j Y_R
            } else {
                float4 y = s_tile[FFT_N - n][tb];
                w = make_float2(y.x + y.w, -y.y + y.z);  // conj(Y_L) + j conj(Y_R)
            }
            v[r] = w;
        }
        fft512_wave<+1>(v, lds, s_tw, lane);
        if (t < T) {
            float* dst = seg + (size_t)((seg0 + t) & (sr - 1)) * 2 * FFT_N;
            const float sc = 1.0f / FFT_N;
#pragma unroll
            for (int j = 0; j < 2; j++) {  // 16 bytes per lane and store
                const int n = 4 * lane + 256 * j;
                const float2 z0 = lds[n], z1 = lds[n + 1], z2 = lds[n + 2], z3 = lds[n + 3];
                *reinterpret_cast<float4*>(dst + n) = make_float4(z0.x * sc, z1.x * sc, z2.x * sc, z3.x * sc);
                *reinterpret_cast<float4*>(dst + FFT_N + n) = make_float4(z0.y * sc, z1.y * sc, z2.y * sc, z3.y * sc);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K3w: inverse transform + overlap-add straight into the wet ring (whole-batch
// path; k_post then reads the ring instead of the segment ring, which saves the
// 4 KB per block round trip through memory).  A workgroup finishes the 15
// blocks of its tile; a sixteenth wave repeats the inverse of the block before
// the tile (its second half is the other addend of the tile's first block) -
// for the first tile of a launch that half comes from the segment ring, where
// the last block of every launch is kept in full for the next launch / call.
// The transforms reuse the memory of the transposed tile (all waves have their
// inputs in registers by then): 78 KB of LDS, two workgroups = 32 waves per CU.
// Y element as in k_inv.  wet[ch][(tau0 + 256 t + m) mod wr].  grid = ceil(T/15), block = 1024.
// ---------------------------------------------------------------------------
#define IW_WAVES 16            // 15 blocks of the tile + the block before it: a multiple of the four SIMDs, two workgroups
#define IW_NEW (IW_WAVES - 1)  // fill a CU's 32 wave slots (nine-wave workgroups put three waves on one SIMD: only two fit)
#define IW_THREADS (64 * IW_WAVES)
// OUT: the launch also finishes the output (what k_post does for the general case): Q1/Q2 window sums, clamp, dry mix,
// shifted by the predelay - for batches with no Q8 pass, no retired predelay epoch ringing out and final prefix sums
// (cring) at launch.  The wet signal then never makes the 8 bytes per frame round trip through memory; only the blocks
// later calls can reach (the last 8192 + frames) and the first block(s), which k_post finishes when the predelay
// reaches back into the previous batch, are still written to the wet ring.
struct OutArgs {
    const float *in1, *in2;  // the batch's input [T * 256] each
    float *outL, *outR;      // the batch's output
    const float2* drop;      // != null (Q8 regime): the cut terms {L, R} of the batch's output frames, [T * 256] (k_drop_fft), subtracted before the clamp
    float* lin;              // != null (a partition shard): the delayed wet partial [2][out_end * 256] is emitted instead - the
                             // summand of the cross-GPU reduce, what k_ola produces; no window sums, clamp or dry mix
    const BlockParams* ptab;
    int pstride;
    const double* cring;
    int rc;
    int64_t tabs0;  // first block of the batch (absolute)
    int64_t predelay, n_ref, b0;  // b0: first block of the live predelay epoch
    int compat, pm;
    int out_end;             // output frames of batch blocks [out_from, out_end) are emitted; the out buffers start at block out_blk0
    int out_blk0;            // (whole batch: out_end = T, out_blk0 = 0; a block-sliced engine: its slice)
    int blk0;                // block of the batch this launch starts at
    int out_from;            // whole batch: blocks < out_from are left to k_post (they need wet samples of earlier calls)
    int wet_head, wet_from;  // blocks of the batch < wet_head or >= wet_from also go to the wet ring
};

// Q1/Q2 window sums {D_L, D_R, Q_L, Q_R} of the wet frame u, which sounds at tau = u + predelay (the arithmetic of k_post):
// calls q of the live epoch with predelay <= tau - q * period < n_ref; the prefix sums are per block
__device__ __forceinline__ void out_window(const OutArgs& A, const int64_t u, double (&win)[4]) {
    win[0] = win[1] = win[2] = win[3] = 0.0;
    if (!A.compat) return;
    // the last block of the call (pm blocks, a power of two) that block b belongs to: b | (pm - 1)
    const int64_t thi = (u >> 8) | (A.pm - 1);
    const int64_t v = u + A.predelay - A.n_ref;
    int64_t tlo = v >= 0 ? ((v >> 8) | (A.pm - 1)) : -1;
    if (tlo < A.b0 - 1) tlo = A.b0 - 1;
    if (thi <= tlo) return;
    const double* a = A.cring + (size_t)(thi & (A.rc - 1)) * 4;
    win[0] = a[0], win[1] = a[1], win[2] = a[2], win[3] = a[3];
    if (tlo >= 0) {
        const double* b = A.cring + (size_t)(tlo & (A.rc - 1)) * 4;
        win[0] -= b[0];
        win[1] -= b[1];
        win[2] -= b[2];
        win[3] -= b[3];
    }
}

__device__ __forceinline__ void out_frame(const bool odd, const float wl, const float wr, const float x1, const float x2,
                                          const BlockParams& bp, const double (&win)[4], float& ol, float& orr) {
    const double sg = odd ? -1.0 : 1.0;
    const double cl = win[0] + sg * win[2], cr = win[1] + sg * win[3];
    const float vl = fminf(fmaxf((float)((double)wl + cl), -1.f), 1.f);
    const float vr = fminf(fmaxf((float)((double)wr + cr), -1.f), 1.f);
    ol = vl + x1 * bp.d[0] + x2 * bp.d[1];
    orr = vr + x1 * bp.d[2] + x2 * bp.d[3];
}

template <bool OUT>
__global__ __launch_bounds__(IW_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_inv_wet(const float4* __restrict__ Ysrc, int64_t sk, int64_t st, int nsum, int64_t sc,
                                                        int T, float* __restrict__ seg, int sr, int seg0, float* __restrict__ wet,
                                                        int wr, int64_t tau0, const float2* __restrict__ g_tw, OutArgs oa) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ __align__(16) float2 s_mem[IW_WAVES * FFT_WAVE_LDS];  // tile [256 bins][16 blocks + 1] of float4, then 16 transforms
    static_assert(sizeof(float2) * IW_WAVES * FFT_WAVE_LDS >= sizeof(float4) * MC_NB * (IW_WAVES + 1), "tile fits the transform buffers");
    float4(*s_tile)[IW_WAVES + 1] = reinterpret_cast<float4(*)[IW_WAVES + 1]>(s_mem);
    load_twiddles(s_tw, g_tw);
    // Workgroup ids 8 apart run on one XCD: give each XCD a contiguous run of tiles, so that the 128-byte lines two
    // neighbouring tiles share (a tile's rows start 16 bytes before a multiple of 15 blocks) are fetched into ONE L2
    int tile;
    {
        const int nt = (int)gridDim.x, q = nt >> 3, r = nt & 7, x = (int)blockIdx.x & 7;
        tile = x * q + min(x, r) + ((int)blockIdx.x >> 3);
#ifdef IW_LINEAR_TILES  // (measurement build: tiles in workgroup order)
        tile = (int)blockIdx.x;
#endif
    }
    const int tb0 = tile * IW_NEW;
    {
        const int c = threadIdx.x % IW_WAVES, k0 = threadIdx.x / IW_WAVES;  // column c <-> block tb0 - 1 + c
        const int t = tb0 - 1 + c;
        const bool live = t >= 0 && t < T;
        const float4* src = Ysrc + (int64_t)k0 * sk + (int64_t)t * st;
        float4 y[MC_NB / 64];
#pragma unroll
        for (int r = 0; r < MC_NB / 64; r++)  // the first (usually only) term of all four bins in flight together
            y[r] = live ? src[(int64_t)64 * r * sk] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
            for (int cc = 1; cc < nsum; cc++) {
#pragma unroll
                for (int r = 0; r < MC_NB / 64; r++) {
                    const float4 a = src[(int64_t)64 * r * sk + cc * sc];
                    y[r].x += a.x;
                    y[r].y += a.y;
                    y[r].z += a.z;
                    y[r].w += a.w;
                }
            }
#pragma unroll
        for (int r = 0; r < MC_NB / 64; r++) s_tile[k0 + 64 * r][c] = y[r];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = tb0 - 1 + wave;
    const int m0 = 4 * lane;
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int n = lane + 64 * r;
        float2 w;
        if (n == 0) {
            float4 y = s_tile[0][wave];
            w = make_float2(y.x, y.z);  // DC of L, R
        } else if (n == MC_B) {
            float4 y = s_tile[0][wave];
            w = make_float2(y.y, y.w);  // Nyquist of L, R
        } else if (n < MC_B) {
            float4 y = s_tile[n][wave];
            w = make_float2(y.x - y.w, y.y + y.z);  // Y_L + j Y_R
        } else {
            float4 y = s_tile[FFT_N - n][wave];
            w = make_float2(y.x + y.w, -y.y + y.z);  // conj(Y_L) + j conj(Y_R)
        }
        v[r] = w;
    }
    __syncthreads();  // the tile is in registers: its memory becomes the transform buffers
    float2* lds = s_mem + wave * FFT_WAVE_LDS;
    if (t >= 0 && t < T) {
        fft512_wave<+1, false>(v, lds, s_tw, lane);
    } else if (t < 0) {
        // the block before the launch: its second half from the segment ring, in the transforms' scale (x 512, exact)
        const float* prv = seg + (size_t)((seg0 + sr - 1) & (sr - 1)) * 2 * FFT_N;
        const float4 a = *reinterpret_cast<const float4*>(prv + MC_B + 4 * lane);
        const float4 b = *reinterpret_cast<const float4*>(prv + FFT_N + MC_B + 4 * lane);
        const float up = (float)FFT_N;
        lds[MC_B + 4 * lane] = make_float2(a.x * up, b.x * up);
        lds[MC_B + 4 * lane + 1] = make_float2(a.y * up, b.y * up);
        lds[MC_B + 4 * lane + 2] = make_float2(a.z * up, b.z * up);
        lds[MC_B + 4 * lane + 3] = make_float2(a.w * up, b.w * up);
    }
    __syncthreads();
    if (wave > 0 && t < T) {
        // OUT: the lane's four wet frames [i0, i0 + 4) of the batch sound at output frames [o0, o0 + 4)
        const int64_t i0 = OUT ? ((int64_t)(oa.blk0 + t) * MC_B + m0) : 0;
        const int64_t o0 = i0 + (OUT ? oa.predelay : 0);
        const bool emits = OUT && o0 + 3 >= (int64_t)oa.out_from * MC_B && o0 < (int64_t)oa.out_end * MC_B;
        const bool whole = emits && ((o0 | oa.n_ref) & 3) == 0 && o0 >= (int64_t)oa.out_from * MC_B;  // one aligned quad (then o0 + 3 is inside too)
        float4 x1q = make_float4(0.f, 0.f, 0.f, 0.f), x2q = x1q;
        if (whole && !oa.lin) {
            x1q = *reinterpret_cast<const float4*>(oa.in1 + o0);
            x2q = *reinterpret_cast<const float4*>(oa.in2 + o0);
        }
        const float scl = 1.0f / FFT_N;
        const float2* prev = lds - FFT_WAVE_LDS + MC_B;
        const float2 o0_ = lds[m0], o1 = lds[m0 + 1], o2 = lds[m0 + 2], o3 = lds[m0 + 3];
        const float2 p0 = prev[m0], p1 = prev[m0 + 1], p2 = prev[m0 + 2], p3 = prev[m0 + 3];
        // (a + b) / 512 == a / 512 + b / 512 exactly: the same bits as overlap-adding scaled segments
        const float4 wl4 = make_float4((o0_.x + p0.x) * scl, (o1.x + p1.x) * scl, (o2.x + p2.x) * scl, (o3.x + p3.x) * scl);
        const float4 wr4 = make_float4((o0_.y + p0.y) * scl, (o1.y + p1.y) * scl, (o2.y + p2.y) * scl, (o3.y + p3.y) * scl);
        if (!OUT || oa.blk0 + t < oa.wet_head || oa.blk0 + t >= oa.wet_from) {
            const size_t at = (size_t)((tau0 + (int64_t)t * MC_B + m0) & (wr - 1));
            *reinterpret_cast<float4*>(wet + at) = wl4;
            *reinterpret_cast<float4*>(wet + wr + at) = wr4;
        }
        if (emits && oa.lin) {
            float* pl = oa.lin + o0;
            float* pr = oa.lin + (size_t)oa.out_end * MC_B + o0;
            if (whole) {
                *reinterpret_cast<float4*>(pl) = wl4;
                *reinterpret_cast<float4*>(pr) = wr4;
            } else {
                const float a[4] = {wl4.x, wl4.y, wl4.z, wl4.w}, b[4] = {wr4.x, wr4.y, wr4.z, wr4.w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (o0 + k >= (int64_t)oa.out_from * MC_B && o0 + k < (int64_t)oa.out_end * MC_B) pl[k] = a[k], pr[k] = b[k];
            }
        } else if (emits) {
            const int64_t u0 = oa.tabs0 * MC_B + i0;
            double win[4];
            if (whole) {
                // four frames of one block, and (predelay, n_ref multiples of four) of one window
                out_window(oa, u0, win);
                const BlockParams& bp = oa.ptab[(o0 >> 8) * oa.pstride];
                float4 d01 = make_float4(0.f, 0.f, 0.f, 0.f), d23 = d01;  // the Q8 cut terms of the four frames {L, R} (as k_post: wet - cut, then the window sums in double)
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
                for (int k = 0; k < 4; k++) {  // a predelay that is no multiple of four frames: frame by frame (rolled: registers)
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
        if (t == T - 1) {  // the launch's last block stays in the segment ring in full
            float* dst = seg + (size_t)((seg0 + t) & (sr - 1)) * 2 * FFT_N;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int n = 4 * lane + 256 * j;
                const float2 z0 = lds[n], z1 = lds[n + 1], z2 = lds[n + 2], z3 = lds[n + 3];
                *reinterpret_cast<float4*>(dst + n) = make_float4(z0.x * scl, z1.x * scl, z2.x * scl, z3.x * scl);
                *reinterpret_cast<float4*>(dst + FFT_N + n) = make_float4(z0.y * scl, z1.y * scl, z2.y * scl, z3.y * scl);
            }
        }
    }
}

// Wet sample m of absolute block b: the overlap-add of that block's first half with
// the previous block's second half, straight from the segment ring (slot = block mod sr).
__device__ __forceinline__ float2 batch_wet(const float* __restrict__ seg, int sr, int64_t b, int m) {
    const float* cur = seg + (size_t)(b & (sr - 1)) * 2 * FFT_N;
    const float* prv = seg + (size_t)((b + sr - 1) & (sr - 1)) * 2 * FFT_N;
    return make_float2(cur[m] + prv[MC_B + m], cur[FFT_N + m] + prv[FFT_N + MC_B + m]);
}

// ---------------------------------------------------------------------------
// Predelay epochs.  The reference adds every call's N_ref-long contribution to
// its accumulator at the predelay current at THAT call (conv.cu:411-415): after
// a predelay change the blocks already played keep ringing out at the old
// offset.  When the predelay changes, the host renders everything the old
// blocks still owe into two residual rings indexed by absolute OUTPUT sample
// and restarts the live pipeline from silence (k_flush_ola / k_flush_fix):
//   mac : the partition sums (of this engine's partition shard), already delayed
//   fix : Q1/Q2 window terms minus Q8 drops of the old blocks
// ---------------------------------------------------------------------------
struct Retired {
    float* mac;   // [2][rr]
    float* fix;   // [2][rr]
    int rr;
    int64_t end;  // the rings hold output samples tau < end (0: nothing retired)
    int64_t b0;   // first block of the live epoch; older blocks live in the rings
};

__device__ __forceinline__ float2 retired_at(const float* __restrict__ ring, int rr, int64_t tau) {
    return make_float2(ring[(size_t)(tau & (rr - 1))], ring[(size_t)rr + (tau & (rr - 1))]);
}

// Wet sample of this engine delayed by the live predelay: writes the thread's own
// overlap-added sample (absolute sample tau) to the wet ring (history for later batches /
// periods) and returns sample tau - predelay: from the segment ring when it is no older than
// win0, the first sample whose segments this call has computed, else from the wet ring.
__device__ __forceinline__ float2 delayed_wet(const float* __restrict__ seg, int sr, float* __restrict__ wet, int wr, int64_t tau,
                                              int64_t win0, int64_t predelay) {
    const int64_t u = tau - predelay;
    const float2 own = batch_wet(seg, sr, tau >> 8, (int)(tau & 255));
    wet[(size_t)(tau & (wr - 1))] = own.x;
    wet[(size_t)wr + (tau & (wr - 1))] = own.y;
    if (u == tau) return own;
    if (u >= win0) return batch_wet(seg, sr, u >> 8, (int)(u & 255));
    if (u >= 0) return make_float2(wet[(size_t)(u & (wr - 1))], wet[(size_t)wr + (u & (wr - 1))]);
    return make_float2(0.f, 0.f);
}

// ---------------------------------------------------------------------------
// K4 (sharded operation only): this shard's overlap-added, predelayed wet
// signal plus what its retired epochs owe, into the caller's linear partial
// buffer [2][T*256] that goes into the cross-GPU sum.  (A single engine needs
// no such pass: k_post does the same straight from the segment ring.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ola(const float* __restrict__ seg, int sr, int T, float* __restrict__ wet, int wr,
                                             int64_t tabs0, int64_t predelay, Retired ret, float* __restrict__ lin) {
    const int t = blockIdx.x, m = threadIdx.x;
    const int64_t tau = (tabs0 + t) * MC_B + m;
    float2 w = delayed_wet(seg, sr, wet, wr, tau, tabs0 * MC_B, predelay);
    if (tau < ret.end) {
        const float2 r = retired_at(ret.mac, ret.rr, tau);
        w.x += r.x;
        w.y += r.y;
    }
    lin[(size_t)t * MC_B + m] = w.x;
    lin[(size_t)T * MC_B + (size_t)t * MC_B + m] = w.y;
}

// the first blocks of a shard's partial when k_inv_wet<true> emitted the rest: their delayed wet frames come from the wet
// ring (the previous call's tail, and this batch's first blocks, which the fused launch also wrote there).  grid = blocks.
__global__ __launch_bounds__(256) void k_ola_head(const float* __restrict__ wet, int wr, int T, int64_t tabs0, int64_t predelay,
                                                  float* __restrict__ lin) {
    const int t = blockIdx.x, m = threadIdx.x;
    const int64_t u = (tabs0 + t) * MC_B + m - predelay;
    float2 w = make_float2(0.f, 0.f);
    if (u >= 0) w = make_float2(wet[(size_t)(u & (wr - 1))], wet[(size_t)wr + (u & (wr - 1))]);
    lin[(size_t)t * MC_B + m] = w.x;
    lin[(size_t)T * MC_B + (size_t)t * MC_B + m] = w.y;
}

// ---------------------------------------------------------------------------
// K5: per-block rank-1 terms of the reference's DC / Nyquist quirks (Q1, Q2;
// conv.cu:61 and :55-71) and their running prefix sums (float64 ring indexed
// by absolute block).  With G = wet gain of a voice's path, sig/alp its IR sums:
//   D_L = -sum_v (G_L0 S2 sig_0R + G_L1 S2 sig_1L)/N   D_R = -sum_v (G_R0 S1 sig_0R + G_R1 S2 sig_1R)/N
//   Q_c = -sum_v (G_c0 A1 alp_0c + G_c1 A2 alp_1c)/N
// ---------------------------------------------------------------------------
__device__ __forceinline__ void corr_terms(const float4 sa, const BlockParams& bp, const VoiceSums& vs, double inv_n,
                                           double (&d)[4]) {
    const double S1 = sa.x, S2 = sa.y, A1 = sa.z, A2 = sa.w;
    d[0] = d[1] = d[2] = d[3] = 0.0;
#pragma unroll
    for (int v = 0; v < MC_MAXV; v++) {
        const double* G = bp.G[v];
        d[0] -= (G[0] * S2 * vs.sig[v][0][1] + G[1] * S2 * vs.sig[v][1][0]) * inv_n;
        d[1] -= (G[2] * S1 * vs.sig[v][0][1] + G[3] * S2 * vs.sig[v][1][1]) * inv_n;
        d[2] -= (G[0] * A1 * vs.alp[v][0][0] + G[1] * A2 * vs.alp[v][1][0]) * inv_n;
        d[3] -= (G[2] * A1 * vs.alp[v][0][1] + G[3] * A2 * vs.alp[v][1][1]) * inv_n;
    }
}

// Prefix sums over the batch, spread over many CUs (one CU alone moves only ~15 GB/s):
//   k_corr_terms: one workgroup per 256 blocks computes the blocks' terms, scans them locally (inclusive) into
//                 the ring and leaves the chunk total in ctot[chunk];
//   k_corr_fix  : one workgroup per 256 blocks adds its chunk's base = previous batch's last sums + totals of
//                 the chunks before it.  Afterwards cring[t] = cumulative {D_L, D_R, Q_L, Q_R} up to block t.
#define CORR_CHUNK 256

// Arguments of the two prefix-sum steps.  They are tiny (one workgroup per 256 blocks) and depend only on the block sums
// k_fwd leaves, so on the headline path they do not get launches of their own: the terms ride along as extra
// workgroups of the k_g2_mac launch, the chunk bases as extra workgroups of the k_inv_wet launch that follows
// (5 us each as kernels: launch and drain latency, not work).
struct CorrArgs {
    const float4* sums;
    const float4* parts;  // != null: sums[t] = the sum of sixteen partial sums the column pass of ossave.hip.h left, [segment][512 tiles][512 rows]:
    int parts_hop, parts_ovl;  // with u = t - parts_t0: block t lies parts_ovl + u % parts_hop blocks into segment u / parts_hop (u < 0: parts_ovl + u blocks into
    int parts_t0, parts_end;   // segment 0, its history), i.e. in row n1 of the sixteen tiles from n2 / 16 on; blocks >= parts_end take sums[t] (a block-sliced engine's tail run)
    const BlockParams* ptab;
    int pstride, T;
    VoiceSums vs;
    double inv_n;
    int compat;
    double* cring;
    int rc;
    int64_t tabs0;
    double* ctot;
    int need_a0, need_a1, need_b0;
    int nchunks;  // 0: nothing rides along
    // chunk cb covers blocks run0 + 256 cb (cb < nrun0) or run1 + 256 (cb - nrun0): a block-sliced rank transforms only the two
    // runs of blocks its windows reach, all other terms are zero and their ring entries are never read, so only the chunks
    // that overlap the runs exist (corr_chunks on the host; a whole batch: run0 = 0, nrun0 = all chunks)
    int run0, nrun0, run1;
    // chain != 0 (workgroups riding along with another kernel's launch): one pass - a workgroup takes a ticket (its chunk),
    // scans its chunk, publishes the chunk total under flags[chunk] = seq and adds the totals of the chunks before it
    // itself, so the ring is final when the launch ends and k_corr_fix is not needed.  A workgroup waits only for
    // lower tickets, which were taken by workgroups already running: the chain always advances.
    int chain;
    unsigned* ticket;      // counter the tickets come from (never reset: ticket = old value - ticket_base)
    unsigned ticket_base;  // tickets handed out by earlier launches
    unsigned* flags;       // [nchunks] sequence number of the launch whose total ctot[chunk] holds
    unsigned seq;
};

// chunk cb (256 blocks) of the batch: Q1/Q2 terms, inclusive scan inside the chunk, chunk total.  Any workgroup size
// >= CORR_CHUNK: threads beyond it only keep the barriers company.
__device__ __forceinline__ void corr_terms_body(int cb, double (*s_part)[4], const CorrArgs& A) {
    // blocks outside [need_a0, need_a1) and [need_b0, T) were not transformed (block-sliced rank): zero terms
    // a single chunk: the base of the previous batch is added here and k_corr_fix is not run
    const int tid = threadIdx.x;
    const bool on = tid < CORR_CHUNK;
    if (A.chain) {  // the chunk is the ticket: tickets are taken in the order the workgroups start
        __shared__ unsigned s_ticket;
        if (tid == 0) s_ticket = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.ticket_base;
        __syncthreads();
        cb = (int)s_ticket;
        if (cb < 0 || cb >= A.nchunks) return;  // (cannot happen: one ticket per riding workgroup)
    }
    const int t = (cb < A.nrun0 ? A.run0 + cb * CORR_CHUNK : A.run1 + (cb - A.nrun0) * CORR_CHUNK) + tid;
    double d[4] = {0, 0, 0, 0};
    if (on && t < A.T && A.compat && ((t >= A.need_a0 && t < A.need_a1) || t >= A.need_b0)) {
        float4 sa;
        if (A.parts && t < A.parts_end) {  // sixteenths in frame order, summed pairwise
            const int u = t - A.parts_t0;
            const int sg = u < 0 ? 0 : u / A.parts_hop, nb = A.parts_ovl + (u < 0 ? u : u % A.parts_hop);  // 32 blocks per row of 8192 frames
            const float4* q = A.parts + ((size_t)sg * 512 + (size_t)(nb & 31) * 16) * 512 + (nb >> 5);
            float4 h[16];
#pragma unroll
            for (int i = 0; i < 16; i++) h[i] = q[(size_t)i * 512];
#pragma unroll
            for (int w = 8; w > 0; w >>= 1)
#pragma unroll
                for (int i = 0; i < w; i++) h[i] = make_float4(h[i].x + h[i + w].x, h[i].y + h[i + w].y, h[i].z + h[i + w].z, h[i].w + h[i + w].w);
            sa = h[0];
        } else
            sa = A.sums[t];
        corr_terms(sa, A.ptab[(int64_t)t * A.pstride], A.vs, A.inv_n, d);
    }
    if (!A.chain && A.nchunks == 1 && tid == 0 && A.tabs0 > 0) {
        const double* p = A.cring + (size_t)((A.tabs0 - 1) & (A.rc - 1)) * 4;
        for (int c = 0; c < 4; c++) d[c] += p[c];
    }
    if (on)
        for (int c = 0; c < 4; c++) s_part[tid][c] = d[c];
    __syncthreads();
    for (int off = 1; off < CORR_CHUNK; off <<= 1) {  // inclusive Hillis-Steele scan
        double v[4] = {0, 0, 0, 0};
        if (on && tid >= off)
            for (int c = 0; c < 4; c++) v[c] = s_part[tid - off][c];
        __syncthreads();
        if (on)
            for (int c = 0; c < 4; c++) s_part[tid][c] += v[c];
        __syncthreads();
    }
    if (!A.chain) {
        if (on && t < A.T) {
            double* o = A.cring + (size_t)((A.tabs0 + t) & (A.rc - 1)) * 4;
            for (int c = 0; c < 4; c++) o[c] = s_part[tid][c];
        }
        if (tid == CORR_CHUNK - 1)
            for (int c = 0; c < 4; c++) A.ctot[cb * 4 + c] = s_part[tid][c];
        return;
    }
    double own[4] = {0, 0, 0, 0};
    if (on)
        for (int c = 0; c < 4; c++) own[c] = s_part[tid][c];
    if (tid == CORR_CHUNK - 1) {  // publish the chunk total
        for (int c = 0; c < 4; c++) __hip_atomic_store(A.ctot + cb * 4 + c, own[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.flags + cb, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // everyone has its own entry in registers: the scan buffer becomes the reduction buffer
    double part[4] = {0, 0, 0, 0};
    bool lost = false;
    if (on)
        for (int k = tid; k < cb; k += CORR_CHUNK) {
            int spins = 0;  // bounded: a total that never arrives poisons the sums (NaN) instead of hanging the GPU
            while (__hip_atomic_load(A.flags + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != A.seq) {
                if (++spins > (1 << 22)) {
                    lost = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            for (int c = 0; c < 4; c++) part[c] += __hip_atomic_load(A.ctot + k * 4 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    if (lost) part[0] = part[1] = part[2] = part[3] = __builtin_nan("");
    if (on)
        for (int c = 0; c < 4; c++) s_part[tid][c] = part[c];
    __syncthreads();
    for (int off = CORR_CHUNK / 2; off > 0; off >>= 1) {
        if (tid < off)
            for (int c = 0; c < 4; c++) s_part[tid][c] += s_part[tid + off][c];
        __syncthreads();
    }
    if (on && t < A.T) {
        double base[4];
        for (int c = 0; c < 4; c++) base[c] = s_part[0][c];
        if (A.tabs0 > 0) {  // the previous batch's last entry (final since an earlier launch)
            const double* p = A.cring + (size_t)((A.tabs0 - 1) & (A.rc - 1)) * 4;
            for (int c = 0; c < 4; c++) base[c] += p[c];
        }
        double* o = A.cring + (size_t)((A.tabs0 + t) & (A.rc - 1)) * 4;
        for (int c = 0; c < 4; c++) o[c] = own[c] + base[c];
    }
}

// chunk cb: add the totals of the chunks before it (and the previous batch's last prefix entry) to its entries
__device__ __forceinline__ void corr_fix_body(const int cb, double (*s_red)[4], const CorrArgs& A) {
    const int tid = threadIdx.x;
    const bool on = tid < CORR_CHUNK;
    const int t = (cb < A.nrun0 ? A.run0 + cb * CORR_CHUNK : A.run1 + (cb - A.nrun0) * CORR_CHUNK) + tid;
    // totals of the chunks before this one: strided partial sums, then a tree in LDS (long batches have hundreds)
    double part[4] = {0, 0, 0, 0};
    if (on)
        for (int k = tid; k < cb; k += CORR_CHUNK)
            for (int c = 0; c < 4; c++) part[c] += A.ctot[k * 4 + c];
    if (on)
        for (int c = 0; c < 4; c++) s_red[tid][c] = part[c];
    __syncthreads();
    for (int off = CORR_CHUNK / 2; off > 0; off >>= 1) {
        if (tid < off)
            for (int c = 0; c < 4; c++) s_red[tid][c] += s_red[tid + off][c];
        __syncthreads();
    }
    double base[4];
    for (int c = 0; c < 4; c++) base[c] = s_red[0][c];
    if (A.tabs0 > 0) {
        const double* p = A.cring + (size_t)((A.tabs0 - 1) & (A.rc - 1)) * 4;
        for (int c = 0; c < 4; c++) base[c] += p[c];
    }
    if (on && t < A.T) {
        double* o = A.cring + (size_t)((A.tabs0 + t) & (A.rc - 1)) * 4;
        for (int c = 0; c < 4; c++) o[c] += base[c];
    }
}

__global__ __launch_bounds__(CORR_CHUNK) void k_corr_terms(CorrArgs A) {
    __shared__ double s_part[CORR_CHUNK][4];
    corr_terms_body((int)blockIdx.x, s_part, A);
}

__global__ __launch_bounds__(CORR_CHUNK) void k_corr_fix(CorrArgs A) {
    __shared__ double s_red[CORR_CHUNK][4];
    corr_fix_body((int)blockIdx.x, s_red, A);
}

// ---------------------------------------------------------------------------
// Q8: the reference adds block t's N_ref-long contribution w_t[s] to its
// accumulator at s + predelay and discards what falls past N_ref
// (f_pointwiseAdd loops s < N, conv.cu:94-98).  For IRs and predelays with
// taps + 255 + predelay > N_ref (calls of pm blocks: taps + 256 pm - 1 + predelay) that cuts real signal: every output sample tau
// loses  sum_{t: 256 t <= tau - N_ref} w_t[tau - predelay - 256 t].  The lost
// terms are recomputed here in the time domain (a 256-term dot product per
// affected block, voice and path) and subtracted before the clamp; the host
// enables the pass only in that regime.  Needs the input history (ring), the
// gains of past blocks (ring) and the time-domain IRs.
// ---------------------------------------------------------------------------
struct TailDrop {
    int on;
    int nv;                     // voices to consider
    const float2* h0[MC_MAXV];  // time-domain taps {L, R} of voice v's IR for input 1
    const float2* h1[MC_MAXV];  // ... for input 2
    int L0[MC_MAXV], L1[MC_MAXV];
    int lmax;
    float* xhist;   // [2][xr] input history ring (absolute sample index mod xr)
    int xr;
    float4* gring;  // [MC_MAXV][rc] wet gains of past blocks
    // frequency-domain form of the same terms (k_drop_fft): the voices' partition spectra, the delay line and
    // its slot gains (both indexed by absolute block mod ring), the transform's twiddles; fft != 0: use it
    int fft;
    const float4* H0s[MC_MAXV];  // [256 bins][pstride_ir] {H_L, H_R} of voice v's IR for input 1
    const float4* H1s[MC_MAXV];  // ... for input 2
    int P0[MC_MAXV], P1[MC_MAXV];
    int pstride_ir;
    const float4* fdl;
    const float4* slotgain;
    int ring;
    const float2* g_tw;
    const float4* Ht0[MC_MAXV];  // the same spectra partition-major for the last partitions, [P - tp][256] (null: read the bank)
    const float4* Ht1[MC_MAXV];
    int tp0[MC_MAXV], tp1[MC_MAXV];
    const float2* dropbuf;  // fft == 2: the terms of the launch's blocks, [count][256] {L, R}, summed by k_drop_fft ahead of k_post
};

__device__ __forceinline__ void tail_drop(const TailDrop& td, int64_t tau, int64_t tau0, int T, int64_t pd, int64_t n_ref,
                                          const BlockParams* __restrict__ ptab, int pstride, int rc,
                                          const float* __restrict__ cur1, const float* __restrict__ cur2, float& dl,
                                          float& dr, int m, int64_t blo, int64_t bhi) {
    // blo..bhi: blocks of the predelay epoch this pass accounts for (inclusive)
    dl = dr = 0.f;
    const int64_t v = tau - n_ref;
    if (v < 0) return;
    // blocks of every call (m blocks each) that started at or before tau - n_ref
    int64_t hi_tb = ((v >> 8) / m + 1) * m - 1;
    const int64_t lo = tau - pd - 254 - td.lmax;
    int64_t lo_tb = lo <= 0 ? 0 : ((lo + 255) >> 8);
    if (lo_tb < blo) lo_tb = blo;
    if (hi_tb > bhi) hi_tb = bhi;
    for (int64_t tb = lo_tb; tb <= hi_tb; tb++) {
        const int64_t s = tau - pd - (tb << 8);  // position inside block tb's own contribution
        const int64_t rel = tb - (tau0 >> 8);
        const int64_t base = tb << 8;
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++) {
            if (vi >= td.nv) break;
            float4 g;
            if (rel >= 0 && rel < T) {
                const float* gv = ptab[rel * pstride].g[vi];
                g = make_float4(gv[0], gv[1], gv[2], gv[3]);
            } else {
                g = td.gring[(size_t)vi * rc + (size_t)(tb & (rc - 1))];
            }
            if (g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) continue;
            float aL0 = 0.f, aR0 = 0.f, aL1 = 0.f, aR1 = 0.f;
            for (int m = 0; m < MC_B; m++) {
                const int64_t j = s - m;
                if (j < 0) break;
                const int64_t sig = base + m;
                float x1, x2;
                if (sig >= tau0) {
                    x1 = cur1[sig - tau0];
                    x2 = cur2[sig - tau0];
                } else {
                    x1 = td.xhist[(size_t)(sig & (td.xr - 1))];
                    x2 = td.xhist[(size_t)td.xr + (sig & (td.xr - 1))];
                }
                if (j < td.L0[vi]) {
                    const float2 h = td.h0[vi][j];
                    aL0 = fmaf(x1, h.x, aL0);
                    aR0 = fmaf(x1, h.y, aR0);
                }
                if (j < td.L1[vi]) {
                    const float2 h = td.h1[vi][j];
                    aL1 = fmaf(x2, h.x, aL1);
                    aR1 = fmaf(x2, h.y, aR1);
                }
            }
            dl += g.x * aL0 + g.y * aL1;
            dr += g.z * aR0 + g.w * aR1;
        }
    }
}

// history for later calls: input samples and this block's wet gains (Q8 pass)
__device__ __forceinline__ void write_history(const TailDrop& td, int64_t tau, int64_t tblock, int m, float x1, float x2,
                                              const BlockParams& bp, int rc) {
    td.xhist[(size_t)(tau & (td.xr - 1))] = x1;
    td.xhist[(size_t)td.xr + (tau & (td.xr - 1))] = x2;
    if (m < MC_MAXV)
        td.gring[(size_t)m * rc + (size_t)(tblock & (rc - 1))] = make_float4(bp.g[m][0], bp.g[m][1], bp.g[m][2], bp.g[m][3]);
}

// The same terms for a TILE of output samples, by the whole workgroup (256 threads; contains barriers: every thread of
// the workgroup calls it).  Round 3: the reference's SHIPPED operating point is in this regime (settings.txt:19,38-45:
// fftSize 131072, predelay 1024 and 14 of the shipped IRs longer than fftSize - 1024), and with tail_drop() above - one
// thread per output sample walking up to 256 input samples and taps through scattered loads - it ran 15x slower in
// batches and 7x slower per JACK period than the same IR without the regime (profiles/r3_shipped_defaults.md).
// Here the source block's 256 input samples and the stretch of taps the tile's samples can pair with them
// (SPAN + 255 taps per input, zero outside [0, L): no bounds tests in the loop) are staged in LDS once per source block
// and voice; a thread then walks the 256 input samples for its NS output samples.  With consecutive samples
// (RSTRIDE = 1, k_post: four frames per lane) the taps slide through registers: one new tap pair per input sample
// for 4 NS multiply-adds.  The order of summation per sample is tail_drop()'s (m ascending per source block).
//   tau_tile: first sample of the tile; thread's samples: tau_tile + r0 + q RSTRIDE, q < NS (all inside one 256-frame
//   block per q).  s_x: [2][256] floats, s_h: [2][SPAN + 256] float2 of LDS.
template <int NS, int RSTRIDE, int SPAN>
__device__ __forceinline__ void tail_drop_tile(const TailDrop& td, float* s_x, float2* s_h, int64_t tau_tile, int r0, bool active,
                                               int64_t tau0, int T, int64_t pd, int64_t n_ref, const BlockParams* __restrict__ ptab,
                                               int pstride, int rc, const float* cur1, const float* cur2, int pm, int64_t blo,
                                               int64_t bhi, float (&dl)[NS], float (&dr)[NS]) {
    constexpr int HW = SPAN + 256;
#pragma unroll
    for (int q = 0; q < NS; q++) dl[q] = dr[q] = 0.f;
    const int64_t v_last = tau_tile + SPAN - 1 - n_ref;
    if (v_last < 0) return;  // (uniform: the whole workgroup leaves)
    int64_t hi_all = ((v_last >> 8) / pm + 1) * pm - 1;
    const int64_t lo = tau_tile - pd - 254 - td.lmax;
    int64_t lo_all = lo <= 0 ? 0 : ((lo + 255) >> 8);
    if (lo_all < blo) lo_all = blo;
    if (hi_all > bhi) hi_all = bhi;
    int64_t hi_q[NS];  // last source block whose contribution this sample has lost (calls that started at or before tau - n_ref)
#pragma unroll
    for (int q = 0; q < NS; q++) {
        const int64_t v = tau_tile + r0 + q * RSTRIDE - n_ref;
        hi_q[q] = (active && v >= 0) ? ((v >> 8) / pm + 1) * pm - 1 : -1;
        if (hi_q[q] > bhi) hi_q[q] = bhi;
    }
    const int tid = threadIdx.x;
    for (int64_t tb = lo_all; tb <= hi_all; tb++) {
        const int64_t base = tb << 8, rel = tb - (tau0 >> 8);
        __syncthreads();  // (the previous source block's samples are no longer read)
        {
            const int64_t sig = base + tid;
            float x1, x2;
            if (sig >= tau0) {
                x1 = cur1[sig - tau0];
                x2 = cur2[sig - tau0];
            } else {
                x1 = td.xhist[(size_t)(sig & (td.xr - 1))];
                x2 = td.xhist[(size_t)td.xr + (sig & (td.xr - 1))];
            }
            s_x[tid] = x1;
            s_x[256 + tid] = x2;
        }
        const int64_t kb = tau_tile - pd - base - 255;  // tap index of window entry 0
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++) {
            if (vi >= td.nv) break;
            float4 g;
            if (rel >= 0 && rel < T) {
                const float* gv = ptab[rel * pstride].g[vi];
                g = make_float4(gv[0], gv[1], gv[2], gv[3]);
            } else {
                g = td.gring[(size_t)vi * rc + (size_t)(tb & (rc - 1))];
            }
            if (g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) continue;  // (uniform)
            __syncthreads();  // (the previous voice's taps are no longer read)
            for (int e = tid; e < HW; e += 256) {
                const int64_t k = kb + e;
                s_h[e] = (k >= 0 && k < td.L0[vi]) ? td.h0[vi][k] : make_float2(0.f, 0.f);
                s_h[HW + e] = (k >= 0 && k < td.L1[vi]) ? td.h1[vi][k] : make_float2(0.f, 0.f);
            }
            __syncthreads();
            float aL0[NS], aR0[NS], aL1[NS], aR1[NS];
#pragma unroll
            for (int q = 0; q < NS; q++) aL0[q] = aR0[q] = aL1[q] = aR1[q] = 0.f;
            const int e0 = r0 + 255;  // window entry of (sample q = 0, input sample m = 0); entry of (q, m) = e0 + q RSTRIDE - m
            // a wave whose samples cannot pair with this source block - it is not one they have lost, or every tap index
            // tau - pd - 256 tb - m of the wave lies outside [0, L) - skips the walk (wave-uniform; the barriers stay outside)
            bool work = false;
#pragma unroll
            for (int q = 0; q < NS; q++) {
                const int64_t s_hi = tau_tile + r0 + q * RSTRIDE - pd - base;  // tap index at m = 0 (the largest of this sample)
                work = work || (tb <= hi_q[q] && s_hi >= 0 && s_hi - 255 < (int64_t)td.lmax);
            }
            if (!__any(work)) {
                // nothing to add
            } else if (RSTRIDE == 1) {
                float2 wa[NS], wb[NS];
#pragma unroll
                for (int q = 0; q < NS; q++) {
                    wa[q] = s_h[e0 + q];
                    wb[q] = s_h[HW + e0 + q];
                }
                for (int m = 0; m < MC_B; m++) {
                    const float x1 = s_x[m], x2 = s_x[256 + m];
#pragma unroll
                    for (int q = 0; q < NS; q++) {
                        aL0[q] = fmaf(x1, wa[q].x, aL0[q]);
                        aR0[q] = fmaf(x1, wa[q].y, aR0[q]);
                        aL1[q] = fmaf(x2, wb[q].x, aL1[q]);
                        aR1[q] = fmaf(x2, wb[q].y, aR1[q]);
                    }
#pragma unroll
                    for (int q = NS - 1; q > 0; q--) {
                        wa[q] = wa[q - 1];
                        wb[q] = wb[q - 1];
                    }
                    const int en = e0 - (m + 1) < 0 ? 0 : e0 - (m + 1);  // (m = 255 at r0 = 0: nothing follows)
                    wa[0] = s_h[en];
                    wb[0] = s_h[HW + en];
                }
            } else {
                for (int m = 0; m < MC_B; m++) {
                    const float x1 = s_x[m], x2 = s_x[256 + m];
#pragma unroll
                    for (int q = 0; q < NS; q++) {
                        const float2 ha = s_h[e0 + q * RSTRIDE - m], hb = s_h[HW + e0 + q * RSTRIDE - m];
                        aL0[q] = fmaf(x1, ha.x, aL0[q]);
                        aR0[q] = fmaf(x1, ha.y, aR0[q]);
                        aL1[q] = fmaf(x2, hb.x, aL1[q]);
                        aR1[q] = fmaf(x2, hb.y, aR1[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NS; q++)
                if (tb <= hi_q[q]) {
                    dl[q] += g.x * aL0[q] + g.y * aL1[q];
                    dr[q] += g.z * aR0[q] + g.w * aR1[q];
                }
        }
    }
    __syncthreads();  // (the caller may reuse the arrays)
}

// The same terms in the frequency domain (round 3).  Block t's contribution at lag s from
// its start is sum_p seg_{t,p}[s - 256 p], seg_{t,p} = IFFT512(X_t H_p); the reference drops it where 256 (b - t) + r >= n_ref,
// i.e. - n_ref being a multiple of 256 - for whole block distances delta = b - t >= dmin(b), whatever the predelay.  With
// predelay = 256 a + c the segment index is 256 kappa + r - c, kappa = delta - a - p in {0, 1, 2}: per kappa the dropped terms of
// block b are ONE slice of IFFT512( sum_{p >= dmin - a - kappa} g(t) H_p X_{b - kappa - a - p} ) - a partition sum over the LAST
// few partitions only (at most predelay / 256 + 2 of them; ONE at the shipped operating point) and one inverse transform, against
// up to 255 x 256 multiply-adds per block, voice and path in the time domain.  Same spectra, gains and segments as the
// partition sums themselves (k_mac_* / k_inv): exact up to fp32 rounding.

// partition-major copy of partitions [p0, P) of an IR's spectra (H: [256 bins][pstride]) -> Ht[(p - p0) * 256 + bin]; grid = span
__global__ void k_h_tail(const float4* __restrict__ H, int pstride, int p0, int P, float4* __restrict__ Ht) {
    const int p = p0 + (int)blockIdx.x, k = threadIdx.x;
    Ht[(size_t)blockIdx.x * MC_NB + k] = p < P ? H[(size_t)k * pstride + p] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// A launch of its own ahead of k_post (k_post<3> only loads the result).  The first version sat inside k_post, one wave per block: 64
// lanes read 64 different 128-byte lines of the delay line per load, behind a chain of dependent round trips (gains, then spectra) at
// two waves per SIMD - 0.9 ms per 125 000 blocks, the same as the time-domain tiles.  Here a workgroup takes DF_WAVES = 8 consecutive
// output blocks: thread (k0 = tid / 8, col = tid % 8) sums bins k0 + 64 r of block col - the delay line is bin-major, so the 8 columns
// of one bin are one 128-byte line - into a [256 bins][DF_WAVES + 1] tile in LDS, wave w then transforms column w and keeps the slice
// of block w's frames; per kappa one partition sum over the last partitions and one inverse transform, as above.  0.40 ms per
// 125 000 blocks at the shipped operating point (one term): without the sum 0.27, without the transform 0.29 (-DDF_DBG=1 / 2); what is
// left is latency per workgroup at 16 waves per CU (k_inv_wet runs the same transforms at 32).  drop: [count][256] {L, R}.
#ifndef DF_WAVES
#define DF_WAVES 8
#endif
#ifndef DF_EU
#define DF_EU 4  // 128 registers: two workgroups per CU (the kernel wants 138; at 80 it spills 51 and takes twice as long)
#endif
__global__ __launch_bounds__(64 * DF_WAVES) __attribute__((amdgpu_waves_per_eu(DF_EU, DF_EU))) void k_drop_fft(TailDrop td, float2* __restrict__ drop, int64_t tabs0, int first,
                                                                                                        int count, int64_t pd, int64_t n_ref, int pm, int64_t blo) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ __align__(16) float2 s_mem[DF_WAVES * FFT_WAVE_LDS];
    static_assert(sizeof(float2) * DF_WAVES * FFT_WAVE_LDS >= sizeof(float4) * MC_NB * (DF_WAVES + 1), "tile fits the transform buffers");
    float4(*s_tile)[DF_WAVES + 1] = reinterpret_cast<float4(*)[DF_WAVES + 1]>(s_mem);
    load_twiddles(s_tw, td.g_tw);
    // workgroup ids 8 apart run on one XCD: each XCD takes a contiguous run of tiles (as k_inv_wet)
    int tile;
    {
        const int nt = (int)gridDim.x, q = nt >> 3, r = nt & 7, xc = (int)blockIdx.x & 7;
        tile = xc * q + min(xc, r) + ((int)blockIdx.x >> 3);
#ifdef DF_LINEAR
        tile = (int)blockIdx.x;
#endif
    }
    const int tb0 = tile * DF_WAVES;
    const int col = threadIdx.x % DF_WAVES, k0 = threadIdx.x / DF_WAVES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = (int)(pd >> 8), c = (int)(pd & 255);
    // the summing role: block b_s = column col; the transforming role: block b_w = column wave
    const int64_t b_s = tabs0 + first + tb0 + col, b_w = tabs0 + first + tb0 + wave;
    const bool act_s = tb0 + col < count && (b_s << 8) >= n_ref, act_w = tb0 + wave < count && (b_w << 8) >= n_ref;
    int dmin_s = 0;
    if (act_s) {
        const int64_t v_own = (b_s << 8) - n_ref;
        dmin_s = (int)(b_s - (((v_own >> 8) / pm + 1) * pm - 1));  // block distances >= dmin_s are cut
    }
    float dl[4] = {0.f, 0.f, 0.f, 0.f}, dr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kappa = 0; kappa < 3; kappa++) {
        if (kappa == 2 && c == 0) break;
        const int p_lo = max(dmin_s - a - kappa, 0);
        bool any = false;  // (from the partition counts alone: a voice whose gains are all zero adds zeros)
        if (act_s)
            for (int vi = 0; vi < td.nv; vi++) any = any || p_lo < max(td.P0[vi], td.P1[vi]);
        if (!__syncthreads_or(any ? 1 : 0)) continue;  // no block of the tile has a term at this kappa (also: the buffers are free again)
        // per partition ONE round trip: the column's gains, the four rows of bins (k0 + 64 rr) and their spectra are requested together, with
        // no branch between them.  The partition index runs over the tile's common range as a scalar (uniform bases and strides); a column
        // whose own range starts later (periods of 2 or 4 blocks), or whose source block lies before the epoch, gets zero gains; so does the
        // input whose IR has no such partition (the bank is zero beyond an IR's last partition, but a voice without an IR points at another one's)
        float4 y[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) y[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int p_lo_u = max((int)(n_ref >> 8) - pm + 1 - a - kappa, 0);
#pragma unroll 1
        for (int vi = 0; vi < td.nv; vi++) {
            const int P0 = td.P0[vi], P1 = td.P1[vi], pmax = max(P0, P1);
            const float4* __restrict__ sg = td.slotgain + (size_t)vi * td.ring;
#pragma unroll 1
#if defined(DF_DBG) && DF_DBG == 1  // (timing ablation: no partition sum)
            for (int p = p_lo_u; p < pmax && pd < 0; p++) {
#else
            for (int p = p_lo_u; p < pmax; p++) {
#endif
                const int64_t t = b_s - kappa - a - p;
                const bool ok = act_s && p >= p_lo && t >= blo && t >= 0;
                const unsigned slot = (unsigned)(t < 0 ? 0 : t) & (unsigned)(td.ring - 1);
                // (uniform) the partition-major copy holds partitions [tp, P) of its IR and nothing beyond: the loads below are unconditional, so
                // a partition the other input's longer IR still has is read from the bank, which is zero there (its gains are zero too)
                const bool t0 = td.Ht0[vi] && p >= td.tp0[vi] && p < P0, t1 = td.Ht1[vi] && p >= td.tp1[vi] && p < P1;
                const float4* __restrict__ B0 = t0 ? td.Ht0[vi] + (size_t)(p - td.tp0[vi]) * MC_NB : td.H0s[vi] + p;
                const float4* __restrict__ B1 = t1 ? td.Ht1[vi] + (size_t)(p - td.tp1[vi]) * MC_NB : td.H1s[vi] + p;
                const unsigned s0 = t0 ? 1u : (unsigned)td.pstride_ir, s1 = t1 ? 1u : (unsigned)td.pstride_ir;
                const unsigned xo = (unsigned)k0 * (unsigned)td.ring + slot, xs = 64u * (unsigned)td.ring;
                float4 g = sg[slot];
                float4 x[4], h0[4], h1[4];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const unsigned k = (unsigned)k0 + 64u * rr;
                    x[rr] = td.fdl[xo + rr * xs];
                    h0[rr] = B0[k * s0];
                    h1[rr] = B1[k * s1];
                }
                const bool m0 = ok && p < P0, m1 = ok && p < P1;
                g.x = m0 ? g.x : 0.f, g.z = m0 ? g.z : 0.f;
                g.y = m1 ? g.y : 0.f, g.w = m1 ? g.w : 0.f;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
                    cmac<false>(a0, h0[rr].x, h0[rr].y, x[rr].x, x[rr].y);
                    cmac<false>(a1, h1[rr].x, h1[rr].y, x[rr].z, x[rr].w);
                    cmac<false>(a2, h0[rr].z, h0[rr].w, x[rr].x, x[rr].y);
                    cmac<false>(a3, h1[rr].z, h1[rr].w, x[rr].z, x[rr].w);
                    if (rr == 0 && k0 == 0) {  // bin 0 packs {DC, Nyquist}: two real products
                        a0 = make_float2(h0[0].x * x[0].x, h0[0].y * x[0].y);
                        a1 = make_float2(h1[0].x * x[0].z, h1[0].y * x[0].w);
                        a2 = make_float2(h0[0].z * x[0].x, h0[0].w * x[0].y);
                        a3 = make_float2(h1[0].z * x[0].z, h1[0].w * x[0].w);
                    }
                    y[rr].x += g.x * a0.x + g.y * a1.x;
                    y[rr].y += g.x * a0.y + g.y * a1.y;
                    y[rr].z += g.z * a2.x + g.w * a3.x;
                    y[rr].w += g.z * a2.y + g.w * a3.y;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) s_tile[k0 + 64 * rr][col] = y[rr];
        __syncthreads();
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {  // Hermitian extension of the packed spectrum Y_L + j Y_R (as k_inv)
            const int n = lane + 64 * r;
            float2 w;
            if (n == 0) {
                const float4 yy = s_tile[0][wave];
                w = make_float2(yy.x, yy.z);
            } else if (n == MC_B) {
                const float4 yy = s_tile[0][wave];
                w = make_float2(yy.y, yy.w);
            } else if (n < MC_B) {
                const float4 yy = s_tile[n][wave];
                w = make_float2(yy.x - yy.w, yy.y + yy.z);
            } else {
                const float4 yy = s_tile[FFT_N - n][wave];
                w = make_float2(yy.x + yy.w, -yy.y + yy.z);
            }
            v[r] = w;
        }
        __syncthreads();  // the tile is in registers: its memory becomes the transform buffers
        if (act_w) {
            float2* lds = s_mem + wave * FFT_WAVE_LDS;
#if defined(DF_DBG) && DF_DBG == 2  // (timing ablation: no transform)
            lds[lane] = v[0], lds[lane + 64] = v[1], lds[lane + 128] = v[2], lds[lane + 192] = v[3];
            lds[lane + 256] = v[4], lds[lane + 320] = v[5], lds[lane + 384] = v[6], lds[lane + 448] = v[7];
#else
            fft512_wave<+1, false>(v, lds, s_tw, lane);
#endif
            const float sc = 1.0f / FFT_N;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 256 * kappa + 4 * lane + q - c;  // segment index of frame r = 4 lane + q
                if (i >= 0 && i < FFT_N) {
                    const float2 z = lds[i];
                    dl[q] += z.x * sc;
                    dr[q] += z.y * sc;
                }
            }
        }
        __syncthreads();
    }
    if (tb0 + wave < count) {
        float4* dst = reinterpret_cast<float4*>(drop + (size_t)(tb0 + wave) * MC_B + 4 * lane);
        dst[0] = make_float4(dl[0], dr[0], dl[1], dr[1]);
        dst[1] = make_float4(dl[2], dr[2], dl[3], dr[3]);
    }
}

// k_post's form of the same sum: a workgroup finishes four consecutive output blocks, one wave per block, four consecutive
// frames per lane.  The tap index of (output sample r of block b, input sample m of source block tb) is
// 256 (b - tb) + r - pd - m: it depends on the block DISTANCE delta = b - tb only, so the walk goes diagonal by diagonal -
// for one delta every wave pairs its own block with the source block delta before it (four source blocks staged at once,
// zero where the pairing is not one the reference cut: then nothing is added), all against ONE window of 511 taps.  All
// four waves walk at once (walking source block by source block left three of them idle at the barriers: 6.6 ms per
// 125 000-block batch at the shipped operating point, 3.4 ms with idle walks skipped, against this form's figure in
// profiles/r3_shipped_defaults.md).  Four input samples per trip: the taps of a trip are two aligned 32-byte reads per
// IR (the window is stored one entry late for that), the inputs one 16-byte broadcast read each, and the four taps a
// lane holds rotate through registers without moves.
// s_x: [4 waves][2 inputs][256] floats, s_h: [2 IRs][516] float2.
__device__ __forceinline__ void tail_drop_tile4(const TailDrop& td, float* s_x, float2* s_h, int64_t b_tile, bool active, int64_t tabs0, int T,
                                                int64_t pd, int64_t n_ref, const BlockParams* __restrict__ ptab, int pstride, int rc,
                                                const float* __restrict__ cur1, const float* __restrict__ cur2, int pm, int64_t blo,
                                                float (&dl)[4], float (&dr)[4]) {
    constexpr int HW = 516;
#pragma unroll
    for (int q = 0; q < 4; q++) dl[q] = dr[q] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t nb = n_ref >> 8;
    // block distances that can carry a cut term: the source block belongs to a call that started at or before tau - n_ref
    // (delta >= n_ref / 256 - (pm - 1)), and some tap index 256 delta + r - pd - m lies in [0, L) (delta <= (L + pd + 254) / 256)
    const int64_t d_lo = nb - (pm - 1), d_hi = ((int64_t)td.lmax + pd + 254) >> 8;
    if (b_tile + 3 - d_lo < blo || b_tile + 3 < d_lo) return;  // (no source block old enough exists yet; uniform)
    const int64_t b_own = b_tile + wave;
    // last source block this wave's samples have lost: the blocks of every call that started at or before tau - n_ref
    const int64_t v_own = (b_own << 8) - n_ref;
    const int64_t hi_own = (active && v_own >= 0) ? ((v_own >> 8) / pm + 1) * pm - 1 : -1;
    for (int64_t delta = d_lo; delta <= d_hi; delta++) {
        __syncthreads();  // (the previous diagonal's samples and taps are no longer read)
        // the four source blocks of this diagonal, 256 samples of both inputs each: thread i stages sample i of every one
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int64_t tb = b_tile + w - delta;
            const int64_t vw = ((b_tile + w) << 8) - n_ref;
            const int64_t hiw = vw >= 0 ? ((vw >> 8) / pm + 1) * pm - 1 : -1;
            float x1 = 0.f, x2 = 0.f;
            if (tb >= blo && tb >= 0 && tb <= hiw) {
                const int64_t sig = (tb << 8) + tid;
                if (tb >= tabs0) {
                    x1 = cur1[sig - (tabs0 << 8)];
                    x2 = cur2[sig - (tabs0 << 8)];
                } else {
                    x1 = td.xhist[(size_t)(sig & (td.xr - 1))];
                    x2 = td.xhist[(size_t)td.xr + (sig & (td.xr - 1))];
                }
            }
            s_x[(w * 2 + 0) * 256 + tid] = x1;
            s_x[(w * 2 + 1) * 256 + tid] = x2;
        }
        const int64_t tb_own = b_own - delta;
        const int64_t rel = tb_own - tabs0;
        const int64_t kb = (delta << 8) - pd - 255;  // tap index of window entry 0 (stored at s_h[1])
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++) {
            if (vi >= td.nv) break;
            if (td.L0[vi] == 0 && td.L1[vi] == 0) continue;  // (uniform)
            if (vi > 0) __syncthreads();  // (the previous voice's taps are no longer read)
            for (int e = tid; e < HW; e += 256) {
                const int64_t k = kb + e - 1;
                s_h[e] = (k >= 0 && k < td.L0[vi]) ? td.h0[vi][k] : make_float2(0.f, 0.f);
                s_h[HW + e] = (k >= 0 && k < td.L1[vi]) ? td.h1[vi][k] : make_float2(0.f, 0.f);
            }
            __syncthreads();
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tb_own >= 0 && tb_own >= blo && tb_own <= hi_own) {
                if (rel >= 0 && rel < T) {
                    const float* gv = ptab[rel * pstride].g[vi];
                    g = make_float4(gv[0], gv[1], gv[2], gv[3]);
                } else {
                    g = td.gring[(size_t)vi * rc + (size_t)(tb_own & (rc - 1))];
                }
            }
            if (g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) continue;  // (wave-uniform; the barriers are behind us)
            // entry of (sample q, input sample m) = 4 lane + 255 + q - m, stored one later
            const float2* ha = s_h + 4 * lane + 256;
            const float2* hb = ha + HW;
            const float* xa = s_x + (wave * 2 + 0) * 256;
            const float* xb = xa + 256;
            v2f a0[4], a1[4];  // {L, R} sums of the four samples: input 1, input 2
#pragma unroll
            for (int q = 0; q < 4; q++) a0[q] = a1[q] = v2f{0.f, 0.f};
            v2f wa[4], wb[4];  // taps of samples q = 0..3 for the current input sample
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float2 ta = ha[q], tb2 = hb[q];
                wa[q] = v2f{ta.x, ta.y};
                wb[q] = v2f{tb2.x, tb2.y};
            }
            for (int m = 0; m < MC_B; m += 4) {
                const float4 x1 = *reinterpret_cast<const float4*>(xa + m), x2 = *reinterpret_cast<const float4*>(xb + m);
                // the four taps that come in during this trip: entries (4 lane + 255 - m) - 1 .. - 4, one aligned 32-byte run
                const float4 na01 = *reinterpret_cast<const float4*>(ha - m - 4), na23 = *reinterpret_cast<const float4*>(ha - m - 2);
                const float4 nb01 = *reinterpret_cast<const float4*>(hb - m - 4), nb23 = *reinterpret_cast<const float4*>(hb - m - 2);
                const v2f n1a = v2f{na23.z, na23.w}, n2a = v2f{na23.x, na23.y}, n3a = v2f{na01.z, na01.w}, n4a = v2f{na01.x, na01.y};
                const v2f n1b = v2f{nb23.z, nb23.w}, n2b = v2f{nb23.x, nb23.y}, n3b = v2f{nb01.z, nb01.w}, n4b = v2f{nb01.x, nb01.y};
#define TD4_STEP(X1, X2, W0A, W1A, W2A, W3A, W0B, W1B, W2B, W3B)                 \
    {                                                                            \
        const v2f xx1 = v2f{X1, X1}, xx2 = v2f{X2, X2};                          \
        a0[0] = __builtin_elementwise_fma(xx1, W0A, a0[0]);                      \
        a0[1] = __builtin_elementwise_fma(xx1, W1A, a0[1]);                      \
        a0[2] = __builtin_elementwise_fma(xx1, W2A, a0[2]);                      \
        a0[3] = __builtin_elementwise_fma(xx1, W3A, a0[3]);                      \
        a1[0] = __builtin_elementwise_fma(xx2, W0B, a1[0]);                      \
        a1[1] = __builtin_elementwise_fma(xx2, W1B, a1[1]);                      \
        a1[2] = __builtin_elementwise_fma(xx2, W2B, a1[2]);                      \
        a1[3] = __builtin_elementwise_fma(xx2, W3B, a1[3]);                      \
    }
                TD4_STEP(x1.x, x2.x, wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]);
                TD4_STEP(x1.y, x2.y, n1a, wa[0], wa[1], wa[2], n1b, wb[0], wb[1], wb[2]);
                TD4_STEP(x1.z, x2.z, n2a, n1a, wa[0], wa[1], n2b, n1b, wb[0], wb[1]);
                TD4_STEP(x1.w, x2.w, n3a, n2a, n1a, wa[0], n3b, n2b, n1b, wb[0]);
#undef TD4_STEP
                wa[3] = n1a, wa[2] = n2a, wa[1] = n3a, wa[0] = n4a;
                wb[3] = n1b, wb[2] = n2b, wb[1] = n3b, wb[0] = n4b;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                dl[q] += g.x * a0[q].x + g.y * a1[q].x;
                dr[q] += g.z * a0[q].y + g.w * a1[q].y;
            }
        }
    }
    __syncthreads();
}

// JACK path: the Q8 terms of one period's PM x 256 samples into drop[2][PM * 256] ({L, R}).  They come from blocks at least
// n_ref frames old, so the host launches this ahead of the period's tail kernel (in the shadow of the previous period, like
// the sweep): off the critical path, and the tail kernels keep their register budget.  One workgroup of 256 threads.
template <int PM>
__global__ __launch_bounds__(256) void k_drop_period(TailDrop td, float* __restrict__ drop, int64_t tabs0, int64_t predelay, int64_t n_ref,
                                                     int rc, int64_t blo) {
    __shared__ float s_tdx[2 * 256];
    __shared__ float2 s_tdh[2 * (PM * MC_B + 256)];
    float dl[PM], dr[PM];
    const int m = threadIdx.x;
    tail_drop_tile<PM, MC_B, PM * MC_B>(td, s_tdx, s_tdh, tabs0 * MC_B, m, true, tabs0 * MC_B, 0, predelay, n_ref, nullptr, 0, rc, nullptr, nullptr, PM, blo,
                                        tabs0 - 1, dl, dr);
#pragma unroll
    for (int j = 0; j < PM; j++) {
        drop[j * MC_B + m] = dl[j];
        drop[PM * MC_B + j * MC_B + m] = dr[j];
    }
}

// The same period's terms in the frequency domain (the sum of k_drop_fft, one wave per block of the period: a period is too few
// blocks for whole-line reads of the delay line to matter, the round trips do): per kappa one request of the gains, the four rows
// of bins and their spectra together per partition, one inverse transform, one slice.  PM waves.
// (body: every thread of the workgroup enters - the twiddles are loaded by all of them - and waves beyond the PM-th leave after the barrier)
template <int PM>
__device__ __forceinline__ void drop_period_fft_body(const TailDrop& td, float* __restrict__ drop, int64_t tabs0, int64_t pd, int64_t n_ref, int64_t blo) {
    __shared__ float2 s_tw[FFT_N];
    __shared__ __align__(16) float2 s_fft[PM][FFT_WAVE_LDS];
    load_twiddles(s_tw, td.g_tw);
    __syncthreads();
    if (threadIdx.x >= 64 * PM) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2* lds = s_fft[wave];
    float4* ybin = reinterpret_cast<float4*>(lds);  // [256] {Y_L, Y_R} per bin, before the transform reuses the memory
    const int64_t b = tabs0 + wave;
    const int a = (int)(pd >> 8), c = (int)(pd & 255);
    float dl[4] = {0.f, 0.f, 0.f, 0.f}, dr[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t v_own = (b << 8) - n_ref;
    if (v_own >= 0) {  // (wave-uniform; nothing below involves another wave)
        const int dmin = (int)(b - (((v_own >> 8) / PM + 1) * PM - 1));
#pragma unroll 1
        for (int kappa = 0; kappa < 3; kappa++) {
            if (kappa == 2 && c == 0) break;
            const int p_lo = max(dmin - a - kappa, 0);
            float4 y[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) y[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
            bool any = false;
#pragma unroll 1
            for (int vi = 0; vi < td.nv; vi++) {
                const int P0 = td.P0[vi], P1 = td.P1[vi], pmax = max(P0, P1);
                const float4* __restrict__ sg = td.slotgain + (size_t)vi * td.ring;
#pragma unroll 1
                for (int p = p_lo; p < pmax; p++) {
                    const int64_t t = b - kappa - a - p;
                    if (t < blo || t < 0) break;
                    any = true;
                    const unsigned slot = (unsigned)t & (unsigned)(td.ring - 1);
                    const bool t0 = td.Ht0[vi] && p >= td.tp0[vi] && p < P0, t1 = td.Ht1[vi] && p >= td.tp1[vi] && p < P1;  // (as k_drop_fft: nothing beyond an IR's last partition in its copy)
                    const float4* __restrict__ B0 = t0 ? td.Ht0[vi] + (size_t)(p - td.tp0[vi]) * MC_NB : td.H0s[vi] + p;
                    const float4* __restrict__ B1 = t1 ? td.Ht1[vi] + (size_t)(p - td.tp1[vi]) * MC_NB : td.H1s[vi] + p;
                    const unsigned s0 = t0 ? 1u : (unsigned)td.pstride_ir, s1 = t1 ? 1u : (unsigned)td.pstride_ir;
                    float4 g = sg[slot];
                    float4 x[4], h0[4], h1[4];
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        const unsigned k = (unsigned)lane + 64u * rr;
                        x[rr] = td.fdl[k * (unsigned)td.ring + slot];
                        h0[rr] = B0[k * s0];
                        h1[rr] = B1[k * s1];
                    }
                    g.x = p < P0 ? g.x : 0.f, g.z = p < P0 ? g.z : 0.f;
                    g.y = p < P1 ? g.y : 0.f, g.w = p < P1 ? g.w : 0.f;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        float2 a0 = make_float2(0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
                        cmac<false>(a0, h0[rr].x, h0[rr].y, x[rr].x, x[rr].y);
                        cmac<false>(a1, h1[rr].x, h1[rr].y, x[rr].z, x[rr].w);
                        cmac<false>(a2, h0[rr].z, h0[rr].w, x[rr].x, x[rr].y);
                        cmac<false>(a3, h1[rr].z, h1[rr].w, x[rr].z, x[rr].w);
                        if (rr == 0 && lane == 0) {  // bin 0 packs {DC, Nyquist}: two real products
                            a0 = make_float2(h0[0].x * x[0].x, h0[0].y * x[0].y);
                            a1 = make_float2(h1[0].x * x[0].z, h1[0].y * x[0].w);
                            a2 = make_float2(h0[0].z * x[0].x, h0[0].w * x[0].y);
                            a3 = make_float2(h1[0].z * x[0].z, h1[0].w * x[0].w);
                        }
                        y[rr].x += g.x * a0.x + g.y * a1.x;
                        y[rr].y += g.x * a0.y + g.y * a1.y;
                        y[rr].z += g.z * a2.x + g.w * a3.x;
                        y[rr].w += g.z * a2.y + g.w * a3.y;
                    }
                }
            }
            if (!any) continue;  // (wave-uniform)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) ybin[lane + 64 * rr] = y[rr];
            fft_sync<false>();
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {  // Hermitian extension of the packed spectrum Y_L + j Y_R (as k_inv)
                const int n = lane + 64 * r;
                float2 w;
                if (n == 0) {
                    const float4 yy = ybin[0];
                    w = make_float2(yy.x, yy.z);
                } else if (n == MC_B) {
                    const float4 yy = ybin[0];
                    w = make_float2(yy.y, yy.w);
                } else if (n < MC_B) {
                    const float4 yy = ybin[n];
                    w = make_float2(yy.x - yy.w, yy.y + yy.z);
                } else {
                    const float4 yy = ybin[FFT_N - n];
                    w = make_float2(yy.x + yy.w, -yy.y + yy.z);
                }
                v[r] = w;
            }
            fft_sync<false>();
            fft512_wave<+1, false>(v, lds, s_tw, lane);
            const float sc = 1.0f / FFT_N;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 256 * kappa + 4 * lane + q - c;  // segment index of frame r = 4 lane + q
                if (i >= 0 && i < FFT_N) {
                    const float2 z = lds[i];
                    dl[q] += z.x * sc;
                    dr[q] += z.y * sc;
                }
            }
            fft_sync<false>();
        }
    }
    *reinterpret_cast<float4*>(drop + wave * MC_B + 4 * lane) = make_float4(dl[0], dl[1], dl[2], dl[3]);
    *reinterpret_cast<float4*>(drop + PM * MC_B + wave * MC_B + 4 * lane) = make_float4(dr[0], dr[1], dr[2], dr[3]);
}
template <int PM>
__global__ __launch_bounds__(64 * PM) void k_drop_period_fft(TailDrop td, float* __restrict__ drop, int64_t tabs0, int64_t pd, int64_t n_ref, int64_t blo) {
    drop_period_fft_body<PM>(td, drop, tabs0, pd, n_ref, blo);
}

// ---------------------------------------------------------------------------
// K6: overlap-add + predelay + Q1/Q2 window sums + Q8 + saturating clamp + dry
// mix (replaces f_pointwiseAdd, f_addDryInterleaved and the residual slide,
// conv.cu:89-100, 126-140, 411-451).  One thread per output frame.
// ---------------------------------------------------------------------------
// wet sample u of this engine when it is not the thread's own: from the segment ring when no older than win0 (the
// first sample whose segments this call has computed), else from the wet ring
__device__ __forceinline__ float2 wet_at(const float* __restrict__ seg, int sr, const float* __restrict__ wet, int wr, int64_t u,
                                         int64_t win0) {
    if (u >= win0) return batch_wet(seg, sr, u >> 8, (int)(u & 255));
    if (u >= 0) return make_float2(wet[(size_t)(u & (wr - 1))], wet[(size_t)wr + (u & (wr - 1))]);
    return make_float2(0.f, 0.f);
}

// TD = 0: the variant for calls whose Q8 pass is off (no tail-drop code, half the registers); 1: the cut terms in the time
// domain (tail_drop_tile4); 3: loaded - summed in the frequency domain by k_drop_fft ahead of the launch (2 was that sum inside this
// kernel, one wave per block: no faster than the tiles, see k_drop_fft) - instantiations of their own, each with its registers
template <int TD>
__global__ __launch_bounds__(256) void k_post(const float* __restrict__ seg, int sr, const float* __restrict__ lin,
                                              float* wet, int wr, const double* __restrict__ cring, int rc,
                                              const BlockParams* __restrict__ ptab, int pstride,
                                              const float* __restrict__ in1, const float* __restrict__ in2,
                                              float* __restrict__ outL, float* __restrict__ outR, int T, int64_t tabs0,
                                              int first, int count, int64_t win0, int64_t predelay, int64_t n_ref, int compat,
                                              TailDrop td, int pm, Retired ret, unsigned* __restrict__ done_flag, unsigned seq,
                                              unsigned* __restrict__ done_ctr, int64_t lin_stride, int lin_first) {
    // lin (partition shards): the summed partial, [2][lin_stride] floats whose first frame is block lin_first of the batch
    // (the whole batch: lin_stride = T * 256, lin_first = 0; a rank that received its slice of a reduce-scatter:
    // lin_stride = count * 256, lin_first = first)
    // done_flag (mapped host memory) != null: outL/outR are host buffers and the last workgroup to finish
    // publishes `seq` once every workgroup's output is visible to the host (one JACK period of 512 / 1024 frames)
    // T blocks in the batch starting at absolute block tabs0; this launch finishes blocks first .. first + count - 1
    // of it (the whole batch unless the engine runs block-sliced) into outL/outR, which start at block `first`.
    // win0: first absolute sample whose segments this call has computed.
    // pm = blocks per reference call (JACK period / 256): Q1/Q2/Q8 windows are measured from the call start
    // One wave per block, four consecutive frames per lane (16-byte loads and stores); grid = ceil(count / 4).
    const int tb = blockIdx.x * 4 + (threadIdx.x >> 6);  // block within the slice
    float tdl[4] = {0.f, 0.f, 0.f, 0.f}, tdr[4] = {0.f, 0.f, 0.f, 0.f};
    if (TD == 3) {  // summed by k_drop_fft ahead of this launch
        if (tb < count) {
            const float4* src = reinterpret_cast<const float4*>(td.dropbuf + (size_t)tb * MC_B + 4 * (threadIdx.x & 63));
            const float4 u = src[0], w = src[1];
            tdl[0] = u.x, tdr[0] = u.y, tdl[1] = u.z, tdr[1] = u.w;
            tdl[2] = w.x, tdr[2] = w.y, tdl[3] = w.z, tdr[3] = w.w;
        }
    } else if (TD && td.on) {  // the Q8 terms of the workgroup's four blocks (1024 samples): four source blocks + one window of taps in LDS
        __shared__ __attribute__((aligned(32))) float s_tdx[4 * 2 * 256];
        __shared__ __attribute__((aligned(32))) float2 s_tdh[2 * 516];
        tail_drop_tile4(td, s_tdx, s_tdh, tabs0 + first + (int64_t)blockIdx.x * 4, tb < count, tabs0, T, predelay, n_ref, ptab, pstride, rc, in1, in2,
                        pm, ret.b0, tdl, tdr);
    }
    if (tb < count) {
        const int t = first + tb, m0 = (threadIdx.x & 63) * 4;
        const int64_t i0 = (int64_t)t * MC_B + m0;
        const int64_t o0 = (int64_t)tb * MC_B + m0;
        const int64_t tau0 = tabs0 * MC_B;
        const int64_t tau_0 = tau0 + i0;
        float wl[4], wr_[4];
        if (lin) {  // shards: the sum over ranks of k_ola's output (predelay and retired partition sums included)
            const int64_t l0 = i0 - (int64_t)lin_first * MC_B;
            const float4 a = *reinterpret_cast<const float4*>(lin + l0), b = *reinterpret_cast<const float4*>(lin + lin_stride + l0);
            wl[0] = a.x, wl[1] = a.y, wl[2] = a.z, wl[3] = a.w;
            wr_[0] = b.x, wr_[1] = b.y, wr_[2] = b.z, wr_[3] = b.w;
        } else {
            float4 ol, orr;
            if (!seg) {  // k_inv_wet has overlap-added the whole window into the wet ring (win0 = INT64_MAX)
                ol = *reinterpret_cast<const float4*>(wet + (size_t)(tau_0 & (wr - 1)));
                orr = *reinterpret_cast<const float4*>(wet + (size_t)wr + (tau_0 & (wr - 1)));
            } else {
                // own samples: this block's first half + the previous block's second half, both channels
                const int64_t b = tau_0 >> 8;
                const float* cur = seg + (size_t)(b & (sr - 1)) * 2 * FFT_N;
                const float* prv = seg + (size_t)((b + sr - 1) & (sr - 1)) * 2 * FFT_N;
                const float4 cl4 = *reinterpret_cast<const float4*>(cur + m0), pl4 = *reinterpret_cast<const float4*>(prv + MC_B + m0);
                const float4 cr4 = *reinterpret_cast<const float4*>(cur + FFT_N + m0);
                const float4 pr4 = *reinterpret_cast<const float4*>(prv + FFT_N + MC_B + m0);
                ol = make_float4(cl4.x + pl4.x, cl4.y + pl4.y, cl4.z + pl4.z, cl4.w + pl4.w);
                orr = make_float4(cr4.x + pr4.x, cr4.y + pr4.y, cr4.z + pr4.z, cr4.w + pr4.w);
                *reinterpret_cast<float4*>(wet + (size_t)(tau_0 & (wr - 1))) = ol;
                *reinterpret_cast<float4*>(wet + (size_t)wr + (tau_0 & (wr - 1))) = orr;
            }
            wl[0] = ol.x, wl[1] = ol.y, wl[2] = ol.z, wl[3] = ol.w;
            wr_[0] = orr.x, wr_[1] = orr.y, wr_[2] = orr.z, wr_[3] = orr.w;
            if (predelay != 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float2 w = wet_at(seg, sr, wet, wr, tau_0 + k - predelay, win0);
                    wl[k] = w.x;
                    wr_[k] = w.y;
                }
            }
            if (tau_0 < ret.end) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (tau_0 + k < ret.end) {
                        const float2 r = retired_at(ret.mac, ret.rr, tau_0 + k);
                        wl[k] += r.x;
                        wr_[k] += r.y;
                    }
            }
        }
        if (tau_0 < ret.end) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (tau_0 + k < ret.end) {
                    const float2 r = retired_at(ret.fix, ret.rr, tau_0 + k);
                    wl[k] += r.x;
                    wr_[k] += r.y;
                }
        }
        const BlockParams& bp = ptab[(int64_t)t * pstride];
        const float4 x1 = *reinterpret_cast<const float4*>(in1 + i0), x2 = *reinterpret_cast<const float4*>(in2 + i0);
        const float x1a[4] = {x1.x, x1.y, x1.z, x1.w}, x2a[4] = {x2.x, x2.y, x2.z, x2.w};
        float ol_[4], or_[4];
        double d0 = 0, d1 = 0, q0 = 0, q1 = 0;
        int64_t have_thi = -2, have_tlo = -2;  // the window of the previous frame: four frames usually share one
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t tau = tau_0 + k, u = tau - predelay;
            double cl = 0.0, cr = 0.0;
            if (compat && u >= 0) {
                // calls q (pm blocks each) of the live epoch with predelay <= tau - q*period < n_ref (shift by predelay,
                // cut at n_ref: Q8); prefix sums are per block, so a call ends at block (q + 1) pm - 1
                const int64_t thi = ((u >> 8) / pm + 1) * pm - 1;
                const int64_t v = tau - n_ref;
                int64_t tlo = v >= 0 ? ((v >> 8) / pm + 1) * pm - 1 : -1;
                if (tlo < ret.b0 - 1) tlo = ret.b0 - 1;
                if (thi > tlo) {
                    if (thi != have_thi || tlo != have_tlo) {
                        const double* a = cring + (size_t)(thi & (rc - 1)) * 4;
                        d0 = a[0], d1 = a[1], q0 = a[2], q1 = a[3];
                        if (tlo >= 0) {
                            const double* b = cring + (size_t)(tlo & (rc - 1)) * 4;
                            d0 -= b[0];
                            d1 -= b[1];
                            q0 -= b[2];
                            q1 -= b[3];
                        }
                        have_thi = thi;
                        have_tlo = tlo;
                    }
                    const double sg = (u & 1) ? -1.0 : 1.0;
                    cl = d0 + sg * q0;
                    cr = d1 + sg * q1;
                }
            }
            float a = wl[k], b = wr_[k];
            if (TD && td.on) {  // input and gain history of the whole batch is in the rings already (k_fwd)
                a -= tdl[k];
                b -= tdr[k];
            }
            const float vl = fminf(fmaxf((float)((double)a + cl), -1.f), 1.f);
            const float vr = fminf(fmaxf((float)((double)b + cr), -1.f), 1.f);
            ol_[k] = vl + x1a[k] * bp.d[0] + x2a[k] * bp.d[1];
            or_[k] = vr + x1a[k] * bp.d[2] + x2a[k] * bp.d[3];
        }
        *reinterpret_cast<float4*>(outL + o0) = make_float4(ol_[0], ol_[1], ol_[2], ol_[3]);
        *reinterpret_cast<float4*>(outR + o0) = make_float4(or_[0], or_[1], or_[2], or_[3]);
    }
    if (done_flag) {
        __syncthreads();  // every lane's stores have been issued and acknowledged (vmcnt(0) at the barrier)
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: this workgroup's output is on the host
            const unsigned old = atomicAdd(done_ctr, 1u);
            if (old == gridDim.x - 1) {
                *done_ctr = 0;
                __hip_atomic_store(done_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Predelay change (host: retire_epoch).  k_flush_ola: the partition sums the
// old blocks still owe, rendered by a run of the pipeline over silent input,
// are added to the residual ring at their OLD predelay; what the reference
// would cut at n_ref after the shift (Q8) is not written.  k_flush_ring: wet
// samples of the old blocks that the old predelay has not released yet move
// from the wet ring to the residual ring.  k_flush_fix: the
// Q1/Q2 window terms of the old blocks for future output samples, minus their
// Q8 drops.  Both accumulate: tails of earlier epochs may still be pending.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flush_ola(const float* __restrict__ seg, int sr, int64_t tabs0,
                                                   int64_t predelay_old, float* __restrict__ res, int rr, int64_t end) {
    const int t = blockIdx.x, m = threadIdx.x;
    const float2 w = batch_wet(seg, sr, tabs0 + t, m);
    const int64_t tau = (tabs0 + t) * MC_B + m + predelay_old;
    if (tau < end) {
        res[(size_t)(tau & (rr - 1))] += w.x;
        res[(size_t)rr + (tau & (rr - 1))] += w.y;
    }
}

// wet samples already computed under the old predelay but not yet played (the last predelay_old samples of the ring)
__global__ __launch_bounds__(256) void k_flush_ring(const float* __restrict__ wet, int wr, int64_t tau_begin, int64_t predelay_old,
                                                    float* __restrict__ res, int rr, int64_t end) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tau = tau_begin + k, u = tau - predelay_old;
    if (k >= predelay_old || u < 0 || tau >= end) return;
    res[(size_t)(tau & (rr - 1))] += wet[(size_t)(u & (wr - 1))];
    res[(size_t)rr + (tau & (rr - 1))] += wet[(size_t)wr + (u & (wr - 1))];
}

__global__ __launch_bounds__(256) void k_flush_fix(const double* __restrict__ cring, int rc, float* __restrict__ res, int rr,
                                                   int64_t tau_begin, int64_t end, int64_t blo, int64_t bhi,
                                                   int64_t predelay_old, int64_t n_ref, int compat, TailDrop td, int pm) {
    // blo..bhi (inclusive): the blocks of the epoch being retired
    const int64_t tau = tau_begin + (int64_t)blockIdx.x * MC_B + threadIdx.x;
    if (tau >= end) return;
    const int64_t u = tau - predelay_old;
    double cl = 0.0, cr = 0.0;
    if (compat && u >= 0) {
        int64_t thi = ((u >> 8) / pm + 1) * pm - 1;
        if (thi > bhi) thi = bhi;
        const int64_t v = tau - n_ref;
        int64_t tlo = v >= 0 ? ((v >> 8) / pm + 1) * pm - 1 : -1;
        if (tlo < blo - 1) tlo = blo - 1;
        if (thi > tlo) {
            const double* a = cring + (size_t)(thi & (rc - 1)) * 4;
            double d0 = a[0], d1 = a[1], q0 = a[2], q1 = a[3];
            if (tlo >= 0) {
                const double* b = cring + (size_t)(tlo & (rc - 1)) * 4;
                d0 -= b[0];
                d1 -= b[1];
                q0 -= b[2];
                q1 -= b[3];
            }
            const double sg = (u & 1) ? -1.0 : 1.0;
            cl = d0 + sg * q0;
            cr = d1 + sg * q1;
        }
    }
    float dl = 0.f, dr = 0.f;
    if (td.on)  // every block of the epoch is history: no current batch (T = 0, tau0 beyond reach)
        tail_drop(td, tau, INT64_MAX, 0, predelay_old, n_ref, nullptr, 0, rc, nullptr, nullptr, dl, dr, pm, blo, bhi);
    res[(size_t)(tau & (rr - 1))] += (float)(cl - (double)dl);
    res[(size_t)rr + (tau & (rr - 1))] += (float)(cr - (double)dr);
}

#include "jack_tail.hip.h"  // the JACK path: k_tail1, k_jack, k_tailp

// ---------------------------------------------------------------------------
// Second-level transform along the block axis (long batches, uniform gains).
// Per bin the partition sum Y[t] = sum_p H[p] X[t - p] is a convolution along
// the block axis; with thousands of outputs per launch it is done as one
// circular convolution of length F2_N per (bin, chunk of blocks): forward
// transforms of the two input sequences, products with the transformed
// partition sequences of the IRs, inverse transforms (overlap-save: the first
// P - 1 outputs of the circle are discarded).  One workgroup of 1024 threads
// holds a whole sequence in LDS (128 KB of gfx950's 160).  The forward
// transform is a decimation-in-frequency radix-4 (natural in, digit-reversed
// out), the inverse a decimation-in-time radix-4 (digit-reversed in, natural
// out) - the exact stage-by-stage inverse - so the spectra never need
// reordering: the IR's second-level spectra are stored in the same
// digit-reversed order (k_fft2_ir).
// Bin 0 packs two real bins {DC, Nyquist} as one complex number z; their two
// real filters act as  y = h1 * z + h2 * conj(z),  h1 = (h_dc + h_ny)/2,
// h2 = (h_dc - h_ny)/2, so bin 0 transforms conj(z) as well.
// ---------------------------------------------------------------------------
#define F2_N 16384
#define F2_THREADS 1024

__device__ __forceinline__ float2 f2_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// The transforms' arithmetic on two-element vectors, the complex products and the +-j rotations spelled out as the
// packed instructions they are (operand swizzles and sign modifiers of v_pk_mul / v_pk_fma / v_pk_add): 144 VALU
// instructions per radix-16 pass and thread.  From the same arithmetic on float2 structs the compiler re-packs scalar
// operations through 70 register moves (262 instructions); from plain vector expressions it still materialises the
// lane-wise negations (186).
// 8-byte LDS accesses of the second-level transforms, LDS-qualified and volatile: the compiler would otherwise pair
// neighbours into ds_read2_b64 / ds_write2_b64, which move their 16 bytes per lane at half the rate of two
// ds_read_b64 (MI355X_MICROARCH.md, LDS table); measured 115 -> 111 us per 32320-block launch of k_g2_mac.
// MC_LDS_PAIRED builds the paired form for comparison.
#ifndef MC_LDS_PAIRED
typedef __attribute__((address_space(3))) volatile v2f lds_vol_v2f;
__device__ __forceinline__ v2f vx_ld(const float2* p) { return *(const lds_vol_v2f*)(p); }
__device__ __forceinline__ void vx_st(float2* p, v2f v) { *(lds_vol_v2f*)(p) = v; }
#else
__device__ __forceinline__ v2f vx_ld(const float2* p) { return *reinterpret_cast<const v2f*>(p); }
__device__ __forceinline__ void vx_st(float2* p, v2f v) { *reinterpret_cast<v2f*>(p) = v; }
#endif
// (v2f, vx_mul, vx_mulc, vx_add_j, vx_sub_j: fft512.hip.h)
// radix-4 butterfly in place: forward y_m = sum_n a_n (-j)^(mn), inverse with +j
template <bool INV>
__device__ __forceinline__ void vx_bfly4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
    const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = INV ? vx_add_j(t1, t3) : vx_sub_j(t1, t3);
    a3 = INV ? vx_sub_j(t1, t3) : vx_add_j(t1, t3);
}
// a1, a2, a3 *= w, w^2, w^3 (INV: their conjugates)
template <bool INV>
__device__ __forceinline__ void vx_tw3(v2f& a1, v2f& a2, v2f& a3, v2f w) {
    const v2f w2 = vx_mul(w, w), w3 = vx_mul(w2, w);
    a1 = INV ? vx_mulc(a1, w) : vx_mul(a1, w);
    a2 = INV ? vx_mulc(a2, w2) : vx_mul(a2, w2);
    a3 = INV ? vx_mulc(a3, w3) : vx_mul(a3, w3);
}
__device__ __forceinline__ v2f vx_tw(const float2* t_lo, const float2* t_hi, int e) { return vx_mul(vx_ld(t_lo + (e & 127)), vx_ld(t_hi + (e >> 7))); }
__device__ __forceinline__ float2 f2_mulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// w^e, w = exp(-2 pi i / F2_N): t_lo[k] = w^k, t_hi[k] = w^(128 k), k < 128
__device__ __forceinline__ float2 f2_tw(const float2* t_lo, const float2* t_hi, int e) { return f2_mul(t_lo[e & 127], t_hi[e >> 7]); }

__device__ __forceinline__ void f2_tables(float2* t_lo, float2* t_hi) {
    if (threadIdx.x < 256) {
        const int k = threadIdx.x & 127;
        const float frac = (threadIdx.x < 128 ? (float)k : (float)(128 * k)) * (-2.0f / (float)F2_N);
        float sn, cs;
        sincospif(frac, &sn, &cs);
        (threadIdx.x < 128 ? t_lo : t_hi)[k] = make_float2(cs, sn);
    }
}

// LDS layout of a sequence: element i lives at F2_P(i) - one pad entry per 64 elements, so that the last three
// radix-4 stages (which stay inside aligned groups of 64 elements) can put consecutive lanes on consecutive GROUPS
// without bank conflicts (stride 65 entries), while the first four stages put consecutive lanes on consecutive
// elements.
#define F2_P(i) ((i) + ((i) >> 6))
#define F2_LDS (F2_N + F2_N / 64)

// radix-4 butterfly in place: forward y_m = sum_n a_n (-j)^(mn), inverse with +j
template <bool INV>
__device__ __forceinline__ void f2_bfly4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y), t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
    const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y), t3 = make_float2(a1.x - a3.x, a1.y - a3.y);
    a0 = make_float2(t0.x + t2.x, t0.y + t2.y);
    a2 = make_float2(t0.x - t2.x, t0.y - t2.y);
    if (INV) {
        a1 = make_float2(t1.x - t3.y, t1.y + t3.x);
        a3 = make_float2(t1.x + t3.y, t1.y - t3.x);
    } else {
        a1 = make_float2(t1.x + t3.y, t1.y - t3.x);
        a3 = make_float2(t1.x - t3.y, t1.y + t3.x);
    }
}
// a1, a2, a3 *= w, w^2, w^3 (INV: their conjugates)
template <bool INV>
__device__ __forceinline__ void f2_tw3(float2& a1, float2& a2, float2& a3, float2 w) {
    if (INV) w.y = -w.y;
    const float2 w2 = f2_mul(w, w), w3 = f2_mul(w2, w);
    a1 = f2_mul(a1, w);
    a2 = f2_mul(a2, w2);
    a3 = f2_mul(a3, w3);
}

// Two consecutive radix-4 stages on the 16 elements pos0 + Q m (m = 0..15) of a thread, in registers: the stage
// with quarter length 4 Q (butterflies over m = r, r+4, r+8, r+12) and the stage with quarter length Q (m = 4g .. 4g+3).
// j0 = pos0 mod Q.  Forward: decimation in frequency (butterfly, then twiddle), first the wide stage; the inverse
// undoes them in the opposite order (conjugate twiddle, then inverse butterfly).
// The sixteenth roots of unity w16^r, r = 1..3.  The four twiddles of a radix-16 pass's wide stage are
// w^((j0 + Q r) step1) = w^(j0 step1) w16^r (Q step1 = N / 16 in every pass): one table lookup and three products with
// constants instead of four lookups (each two LDS reads, a product and index arithmetic).
#define W16_1 v2f{0.92387953251128674f, -0.38268343236508977f}
#define W16_2 v2f{0.70710678118654752f, -0.70710678118654752f}
#define W16_3 v2f{0.38268343236508977f, -0.92387953251128674f}

// LDS addresses of a thread's 16 elements: the pad of F2_P / G2_P is one entry per GROUP elements, and in every pass
// the elements pos0 + Q m are either whole groups apart (Q >= GROUP: (pos0 + Q m) / GROUP = pos0 / GROUP + m Q / GROUP)
// or inside one group (Q m < GROUP - pos0 mod GROUP by the passes' index maps), so they are base + m x constant:
// one address per thread and pass, the rest are immediate offsets of the ds instructions.
template <bool INV, int LQ>
__device__ __forceinline__ void f2_pair(float2* s, const float2* t_lo, const float2* t_hi, int pos0, int j0) {
    constexpr int Q = 1 << LQ;
    constexpr int stride = Q >= 64 ? Q + Q / 64 : Q;
    constexpr int step1 = F2_N >> (LQ + 4), step2 = F2_N >> (LQ + 2);
    float2* p = s + F2_P(pos0);
    v2f a[16];
#pragma unroll
    for (int m = 0; m < 16; m++) a[m] = vx_ld(p + stride * m);
    const v2f wa = vx_tw(t_lo, t_hi, j0 * step1), w = vx_tw(t_lo, t_hi, j0 * step2);
    const v2f wr[4] = {wa, vx_mul(wa, W16_1), vx_mul(wa, W16_2), vx_mul(wa, W16_3)};
    if (!INV) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            vx_bfly4<false>(a[r], a[r + 4], a[r + 8], a[r + 12]);
            vx_tw3<false>(a[r + 4], a[r + 8], a[r + 12], wr[r]);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            vx_bfly4<false>(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
            vx_tw3<false>(a[4 * g + 1], a[4 * g + 2], a[4 * g + 3], w);
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            vx_tw3<true>(a[4 * g + 1], a[4 * g + 2], a[4 * g + 3], w);
            vx_bfly4<true>(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            vx_tw3<true>(a[r + 4], a[r + 8], a[r + 12], wr[r]);
            vx_bfly4<true>(a[r], a[r + 4], a[r + 8], a[r + 12]);
        }
    }
#pragma unroll
    for (int m = 0; m < 16; m++) vx_st(p + stride * m, a[m]);
}

// the last forward / first inverse stage (quarter length 1, no twiddles): four butterflies per thread on quads of
// different 64-element groups (consecutive lanes: consecutive groups)
template <bool INV>
__device__ __forceinline__ void f2_quads(float2* s, int t) {
    const int G = t & 255, u0 = t >> 8;
    v2f a[16];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) a[4 * r + k] = vx_ld(&s[F2_P(64 * G + 4 * (u0 + 4 * r) + k)]);
#pragma unroll
    for (int r = 0; r < 4; r++) vx_bfly4<INV>(a[4 * r], a[4 * r + 1], a[4 * r + 2], a[4 * r + 3]);
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) vx_st(&s[F2_P(64 * G + 4 * (u0 + 4 * r) + k)], a[4 * r + k]);
}

// natural order in, digit-reversed order out (unscaled forward transform); every pass moves the sequence through
// registers in 1024 pieces of 16 elements
#define F2_PIECES (F2_N / 16)
// QUADS = false leaves the last stage (radix-4 on adjacent quads, no twiddles) to the caller: k_f2_prod applies it to
// the quads a thread owns on the way into the products and its inverse on the way out (one trip through LDS less in
// k_f2_fwd and one in k_f2_prod)
template <bool QUADS = true>
__device__ __forceinline__ void f2_forward(float2* s, const float2* t_lo, const float2* t_hi) {
    for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS) f2_pair<false, 10>(s, t_lo, t_hi, t, t);  // quarter lengths 4096, 1024
    __syncthreads();
    for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS)
        f2_pair<false, 6>(s, t_lo, t_hi, ((t >> 6) << 10) + (t & 63), t & 63);  // 256, 64
    __syncthreads();
    for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS)
        f2_pair<false, 2>(s, t_lo, t_hi, ((t & 255) << 6) + (t >> 8), t >> 8);  // 16, 4
    __syncthreads();
    if (QUADS) {
        for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS) f2_quads<false>(s, t);  // 1
        __syncthreads();
    }
}

// digit-reversed order in, natural order out (unscaled inverse: F2_N times the input sequence)
template <bool QUADS = true>
__device__ __forceinline__ void f2_inverse(float2* s, const float2* t_lo, const float2* t_hi) {
    if (QUADS) {
        for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS) f2_quads<true>(s, t);
        __syncthreads();
    }
    for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS)
        f2_pair<true, 2>(s, t_lo, t_hi, ((t & 255) << 6) + (t >> 8), t >> 8);
    __syncthreads();
    for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS)
        f2_pair<true, 6>(s, t_lo, t_hi, ((t >> 6) << 10) + (t & 63), t & 63);
    __syncthreads();
    for (int t = threadIdx.x; t < F2_PIECES; t += F2_THREADS) f2_pair<true, 10>(s, t_lo, t_hi, t, t);
    __syncthreads();
}

// The forward transform leaves frequency f at position rev4(f) (its seven base-4 digits reversed).  Position of
// frequency -f: used for bin 0, where the spectrum of conj(z) is needed: FFT(conj z)[f] = conj(FFT(z)[-f]).
__device__ __forceinline__ int f2_rev4(int x) {
    int r = (int)(__brev((unsigned)x) >> 18);             // 14 bits reversed
    return ((r & 0x2AAA) >> 1) | ((r & 0x1555) << 1);     // bits inside each base-4 digit back in order
}
__device__ __forceinline__ int f2_mirror(int p) { return f2_rev4((F2_N - f2_rev4(p)) & (F2_N - 1)); }

// Second-level spectra of an IR: for channel c (0 = L, 1 = R) and row r (bins 0 .. 255, row 256 = bin 0's h2)
// the forward transform of the zero-padded partition sequence, digit-reversed order:
// out[(c * 257 + r) * F2_N + i].   grid = (257, 2), block = 1024.
__global__ __launch_bounds__(F2_THREADS) void k_fft2_ir(const float4* __restrict__ H, int pstride, int P, float2* __restrict__ out) {
    __shared__ float2 s[F2_LDS];
    __shared__ float2 t_lo[128], t_hi[128];
    const int row = blockIdx.x, c = blockIdx.y, bin = row == 256 ? 0 : row;
    f2_tables(t_lo, t_hi);
    for (int i = threadIdx.x; i < F2_N; i += F2_THREADS) {
        float2 v = make_float2(0.f, 0.f);
        if (i < P) {
            const float4 h = H[(size_t)bin * pstride + i];
            const float2 hc = c == 0 ? make_float2(h.x, h.y) : make_float2(h.z, h.w);
            if (bin != 0)
                v = hc;
            else  // hc = {h_dc, h_ny}: both real
                v = make_float2(row == 0 ? 0.5f * (hc.x + hc.y) : 0.5f * (hc.x - hc.y), 0.f);
        }
        s[F2_P(i)] = v;
    }
    __syncthreads();
    f2_forward(s, t_lo, t_hi);
    float2* dst = out + ((size_t)c * 257 + row) * F2_N;
    for (int i = threadIdx.x; i < F2_N; i += F2_THREADS) dst[i] = s[F2_P(i)];
}

struct Fft2Voices {
    int n;
    const float2* h0[MC_MAXV];  // second-level spectra of the voice's IR for input 1 / input 2
    const float2* h1[MC_MAXV];
    float4 g[MC_MAXV];          // {L<-in1, L<-in2, R<-in1, R<-in2}
};

// The second-level transform runs as two kernels so that each workgroup holds one sequence and nothing else
// (both are register-bound at 1024 threads):
//   k_f2_fwd : grid 256 bins x chunks x 2 inputs: window of input i -> forward transform -> stash
//   k_f2_prod: grid 256 bins x chunks x 2 channels: sum over inputs and voices of gain * spectrum * IR spectrum ->
//              inverse transform -> the valid part of the circle (its first taps - 1 outputs are discarded) to Yc
// Chunk c covers output blocks [c * chunk_t, ...) of the batch; taps = partitions swept (>= every voice's),
// chunk_t + taps - 1 <= F2_N.  stash: [(chunk * 256 + bin) * 2 + i][F2_N].
// Both grids are 1-D: the two workgroups of a (bin, chunk) - the two inputs, or the two output channels - read the
// same delay-line window / the same stash entries, so they get block ids 8 apart: same XCD (id mod 8), dispatched
// together, and the second reader finds the data in that XCD's L2.
// Sequences per (bin, chunk): uniform gains - the 2 inputs (gains applied in the product); per-slot gains - for
// every voice v and path (c, i) the sequence gain_v,c,i(slot) * X_i(slot): nseq = 4 * voices, index v * 4 + c * 2 + i.
// The chunks of a group of 8 bins follow each other, so that on every XCD the chunks of ITS bin of the group run
// together: the IR's second-level spectra of the bin and the overlap of adjacent windows are read once into that L2.
__device__ __forceinline__ void f2_decode(int id, int nz, int& bin, int& chunk, int& z) {
    const int nch = (int)gridDim.x / (MC_NB * nz);
    const int g = id / (8 * nz), w = id - g * 8 * nz;  // nz workgroups per (bin, chunk), 8 bins per group
    z = w >> 3;
    bin = ((g / nch) << 3) + (w & 7);
    chunk = g % nch;
}
struct Fft2Gains {
    const float4* row[MC_MAXV];  // per-slot gain table of each voice (null: uniform gains, nseq = 2)
};

__global__ __launch_bounds__(F2_THREADS) void k_f2_fwd(const float4* __restrict__ fdl, int ring, int slot0, int T, int chunk_t,
                                                       int taps, float2* __restrict__ stash, int nseq, Fft2Gains gg) {
    __shared__ float2 s[F2_LDS];
    __shared__ float2 t_lo[128], t_hi[128];
    constexpr int R = F2_N / F2_THREADS;
    int bin, chunk, z;
    f2_decode(blockIdx.x, nseq, bin, chunk, z);
    const int t_c0 = chunk * chunk_t, nout = min(chunk_t, T - t_c0), L = nout + taps - 1;
    const float4* fk = fdl + (size_t)bin * ring;
    const int sb = slot0 + t_c0 - (taps - 1);
    // uniform gains: z = input; per-slot gains: z = voice * 4 + channel * 2 + input
    const int i = z & 1, path = z & 3, v = z >> 2;
    const float4* grow = nullptr;
    if (nseq > 2) {
#pragma unroll
        for (int vi = 0; vi < MC_MAXV; vi++)
            if (vi == v) grow = gg.row[vi];
    }
    f2_tables(t_lo, t_hi);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int n = threadIdx.x + F2_THREADS * r;
        float2 val = make_float2(0.f, 0.f);
        if (n < L) {
            const int slot = (sb + n) & (ring - 1);
            const float4 x = fk[slot];
            val = i == 0 ? make_float2(x.x, x.y) : make_float2(x.z, x.w);
            if (grow) {
                const float4 g4 = grow[slot];
                const float g = path == 0 ? g4.x : (path == 1 ? g4.y : (path == 2 ? g4.z : g4.w));
                val.x *= g;
                val.y *= g;
            }
        }
        s[F2_P(n)] = val;
    }
    __syncthreads();
    f2_forward<false>(s, t_lo, t_hi);  // the stash holds the sequence before the last stage, see k_f2_prod
    float4* my = reinterpret_cast<float4*>(stash + (((size_t)chunk * MC_NB + bin) * nseq + z) * F2_N);
#pragma unroll
    for (int r = 0; r < R / 2; r++) {  // 16 bytes per lane: entries 2j, 2j + 1
        const int j = threadIdx.x + F2_THREADS * r;
        const float2 a = s[F2_P(2 * j)], b = s[F2_P(2 * j + 1)];
        my[j] = make_float4(a.x, a.y, b.x, b.y);
    }
}

__global__ __launch_bounds__(F2_THREADS) void k_f2_prod(const float2* __restrict__ stash, int T, int chunk_t, int taps, Fft2Voices vv,
                                                        float4* __restrict__ Yc, int ycap, int nseq) {
    __shared__ float2 s[F2_LDS];
    __shared__ float2 t_lo[128], t_hi[128];
    constexpr int R = F2_N / F2_THREADS;
    int bin, chunk, c;
    f2_decode(blockIdx.x, 2, bin, chunk, c);
    const int t_c0 = chunk * chunk_t, nout = min(chunk_t, T - t_c0);
    const float2* my = stash + ((size_t)chunk * MC_NB + bin) * nseq * F2_N;
    const bool per_slot = nseq > 2;  // the stash holds gain-weighted sequences per voice and path
    f2_tables(t_lo, t_hi);
#pragma unroll 2
    for (int r = 0; r < R / 4; r++) {  // a quad of entries 4j .. 4j + 3 per lane: the group of the transforms' last stage
        const int j = threadIdx.x + F2_THREADS * r, idx = 4 * j;
        v2f acc[4] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
        for (int q = 0; q < (bin == 0 ? 4 : 2); q++) {
            const int i = q & 1, var = q >> 1;  // var 1 (bin 0 only): the spectrum of conj(x_i) against h2
            const size_t row = ((size_t)c * 257 + (var ? 256 : bin)) * F2_N;
#pragma unroll
            for (int vi = 0; vi < MC_MAXV; vi++) {
                if (vi >= vv.n) break;
                const float2* sq = my + (size_t)(per_slot ? vi * 4 + c * 2 + i : i) * F2_N;
                // the stash holds the sequence before the last forward stage: finish it on the quad (for bin 0's second
                // pass on the mirrored quad - the frequencies -f of a quad are a quad again, in another order)
                const int m0 = var ? f2_mirror(idx) : idx;
                const float4 Sa = reinterpret_cast<const float4*>(sq + (m0 & ~3))[0];
                const float4 Sb = reinterpret_cast<const float4*>(sq + (m0 & ~3))[1];
                v2f P0 = v2f{Sa.x, Sa.y}, P1 = v2f{Sa.z, Sa.w}, P2 = v2f{Sb.x, Sb.y}, P3 = v2f{Sb.z, Sb.w};
                vx_bfly4<false>(P0, P1, P2, P3);
                v2f S[4];
                if (var) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int mk = f2_mirror(idx + k) & 3;
                        v2f t = mk == 0 ? P0 : (mk == 1 ? P1 : (mk == 2 ? P2 : P3));
                        t.y = -t.y;
                        S[k] = t;
                    }
                } else {
                    S[0] = P0, S[1] = P1, S[2] = P2, S[3] = P3;
                }
                const float2* h = i == 0 ? vv.h0[vi] : vv.h1[vi];
                const float g = per_slot ? 1.0f : (c == 0 ? (i == 0 ? vv.g[vi].x : vv.g[vi].y) : (i == 0 ? vv.g[vi].z : vv.g[vi].w));
                const float4 Ha = reinterpret_cast<const float4*>(h + row + idx)[0];
                const float4 Hb = reinterpret_cast<const float4*>(h + row + idx)[1];
                acc[0] += g * vx_mul(S[0], v2f{Ha.x, Ha.y});
                acc[1] += g * vx_mul(S[1], v2f{Ha.z, Ha.w});
                acc[2] += g * vx_mul(S[2], v2f{Hb.x, Hb.y});
                acc[3] += g * vx_mul(S[3], v2f{Hb.z, Hb.w});
            }
        }
        vx_bfly4<true>(acc[0], acc[1], acc[2], acc[3]);  // the inverse transform's first stage
#pragma unroll
        for (int k = 0; k < 4; k++) vx_st(&s[F2_P(idx + k)], acc[k]);
    }
    __syncthreads();
    f2_inverse<false>(s, t_lo, t_hi);
    const float sc = 1.0f / (float)F2_N;
    float2* dst = reinterpret_cast<float2*>(Yc + (size_t)bin * ycap + t_c0) + c;  // .xy = Y_L, .zw = Y_R
    for (int t = threadIdx.x; t < nout; t += F2_THREADS) {
        const float2 y = s[F2_P(t + taps - 1)];
        dst[2 * (size_t)t] = make_float2(y.x * sc, y.y * sc);
    }
}

// ---------------------------------------------------------------------------
// Second-level transform, fused form for uniform gains and IRs up to 5632
// partitions: transforms of 8192 points, so that the spectra of BOTH input
// sequences of a (bin, chunk) fit in LDS side by side (2 x 66.5 KB).  One
// workgroup reads the delay-line window once (16 bytes per slot: both inputs),
// transforms the two sequences in its two halves (512 threads each), forms
// Y_L and Y_R in place (every thread owns its entries of both buffers), inverts
// both and writes {Y_L, Y_R} as 16-byte entries.  No stash: per 8192-block launch
// it moves the windows (41 MB), the second-level spectra (134 MB for two chunks)
// and the sums (33 MB).
// 8192 = 4^6 x 2: radix-16 passes with quarter lengths 2048/512, 128/32, 8/2,
// then one radix-2 stage.  Position p after the forward transform holds
// frequency rev(p): the six base-4 digits of p >> 1 reversed, plus (p & 1) << 12.
// ---------------------------------------------------------------------------
#define G2_N 8192
// one pad entry per 32 elements: the 8/2 pass puts consecutive lanes on consecutive 32-element groups (stride 33)
#define G2_P(i) ((i) + ((i) >> 5))
#define G2_LDS (G2_N + G2_N / 32)
#define G2_THREADS 1024
#ifndef G2_PW
#define G2_PW 4  // window rows (of 8) requested one item ahead in the persistent form of k_g2_mac
#endif
#ifndef G2_PF
#define G2_PF 2  // groups of product entries whose spectra are loaded together
#endif
#ifndef G2_ABL
#define G2_ABL 0  // timing-only ablations of k_g2_mac (wrong results), bit flags: 1 no global memory, 2 no butterflies, 4 no LDS accesses in the passes that load AND store (scripts/gpu_abl.sh)
#endif
#ifndef G2_AHEAD
#define G2_AHEAD 0  // 1: the first G2_PF groups' spectra are requested before the forward transforms
#endif

__device__ __forceinline__ float2 g2_tw(const float2* t_lo, const float2* t_hi, int e) { return f2_mul(t_lo[e & 127], t_hi[e >> 7]); }
__device__ __forceinline__ v2f vg_tw(const float2* t_lo, const float2* t_hi, int e) { return vx_mul(vx_ld(t_lo + (e & 127)), vx_ld(t_hi + (e >> 7))); }

__device__ __forceinline__ void g2_tables(float2* t_lo, float2* t_hi) {  // w = exp(-2 pi i / 8192): w^k (128), w^(128 k) (64)
    if (threadIdx.x < 192) {
        const bool lo = threadIdx.x < 128;
        const int k = lo ? threadIdx.x : threadIdx.x - 128;
        float sn, cs;
        sincospif((lo ? (float)k : (float)(128 * k)) * (-2.0f / (float)G2_N), &sn, &cs);
        (lo ? t_lo : t_hi)[k] = make_float2(cs, sn);
    }
}

// two consecutive radix-4 stages (quarter lengths 4 Q and Q) on the 16 elements pos0 + Q m of a thread; see f2_pair
// the two radix-4 stages of a pass on a thread's 16 elements, in registers (a[m] = element pos0 + Q m)
template <bool INV, int LQ>
__device__ __forceinline__ void g2_pair_core(v2f (&a)[16], const float2* t_lo, const float2* t_hi, int j0) {
#if G2_ABL & 2
    return;
#endif
    constexpr int step1 = G2_N >> (LQ + 4), step2 = G2_N >> (LQ + 2);
    const v2f wa = vg_tw(t_lo, t_hi, j0 * step1), w = vg_tw(t_lo, t_hi, j0 * step2);
    const v2f wr[4] = {wa, vx_mul(wa, W16_1), vx_mul(wa, W16_2), vx_mul(wa, W16_3)};
    if (!INV) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            vx_bfly4<false>(a[r], a[r + 4], a[r + 8], a[r + 12]);
            vx_tw3<false>(a[r + 4], a[r + 8], a[r + 12], wr[r]);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            vx_bfly4<false>(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
            vx_tw3<false>(a[4 * g + 1], a[4 * g + 2], a[4 * g + 3], w);
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            vx_tw3<true>(a[4 * g + 1], a[4 * g + 2], a[4 * g + 3], w);
            vx_bfly4<true>(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            vx_tw3<true>(a[r + 4], a[r + 8], a[r + 12], wr[r]);
            vx_bfly4<true>(a[r], a[r + 4], a[r + 8], a[r + 12]);
        }
    }
}
// LOAD / STORE = false: the elements come from / stay in the caller's registers (k_g2_mac: the first forward pass takes
// the window rows a thread has just loaded - they ARE its elements tt + 512 m - and the last inverse pass leaves the
// outputs with the thread that stores them: two trips through LDS and two barriers less per sequence)
template <bool INV, int LQ, bool LOAD = true, bool STORE = true>
__device__ __forceinline__ void g2_pair(float2* s, const float2* t_lo, const float2* t_hi, int pos0, int j0, v2f (&a)[16]) {
    constexpr int Q = 1 << LQ;
    constexpr int stride = Q >= 32 ? Q + Q / 32 : Q;  // elements Q apart: whole pad groups apart, or inside one (see f2_pair)
    // the LDS address is recomputed in every pass: shared between the forward and the inverse transform it would stay
    // live across the whole kernel and spill
    asm volatile("" : "+v"(pos0), "+v"(j0));
    float2* p = s + G2_P(pos0);
#if G2_ABL & 4
    if (LOAD && STORE) {  // the butterflies on whatever the registers hold, kept alive without a store
#pragma unroll
        for (int m = 0; m < 16; m++) a[m] = v2f{__int_as_float(0x3f000000 | (pos0 + m)), __int_as_float(0x3f000000 | (j0 + m))};
        g2_pair_core<INV, LQ>(a, t_lo, t_hi, j0);
#pragma unroll
        for (int m = 0; m < 16; m++) asm volatile("" ::"v"(a[m]));
        return;
    }
#endif
    if (LOAD) {
#pragma unroll
        for (int m = 0; m < 16; m++) a[m] = vx_ld(p + stride * m);
    }
    g2_pair_core<INV, LQ>(a, t_lo, t_hi, j0);
    if (STORE) {
#pragma unroll
        for (int m = 0; m < 16; m++) vx_st(p + stride * m, a[m]);
    }
}
template <bool INV, int LQ>
__device__ __forceinline__ void g2_pair(float2* s, const float2* t_lo, const float2* t_hi, int pos0, int j0) {
    v2f a[16];
    g2_pair<INV, LQ, true, true>(s, t_lo, t_hi, pos0, j0, a);
}

// the radix-2 stage on adjacent pairs (its own inverse up to the factor 2): eight pairs per thread
__device__ __forceinline__ void g2_pairs2(float2* s, int tt) {
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int p = tt + 512 * r;
        const v2f a = vx_ld(&s[G2_P(2 * p)]), b = vx_ld(&s[G2_P(2 * p + 1)]);
        vx_st(&s[G2_P(2 * p)], a + b);
        vx_st(&s[G2_P(2 * p + 1)], a - b);
    }
}

// forward / inverse transform of the buffer this half of the workgroup (tt = thread within the half, 0..511) owns
// PAIRS2 = false leaves the radix-2 stage on adjacent pairs to the caller (k_g2_mac applies it to the pair of entries a
// thread owns while it forms the products, and again before the inverse: two trips through LDS less)
template <bool PAIRS2 = true>
__device__ __forceinline__ void g2_forward(float2* s, const float2* t_lo, const float2* t_hi, int tt) {
    g2_pair<false, 9>(s, t_lo, t_hi, tt, tt);  // quarter lengths 2048, 512
    __syncthreads();
    g2_pair<false, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);  // 128, 32
    __syncthreads();
    g2_pair<false, 1>(s, t_lo, t_hi, ((tt & 255) << 5) + (tt >> 8), tt >> 8);  // 8, 2
    __syncthreads();
    if (PAIRS2) {
        g2_pairs2(s, tt);
        __syncthreads();
    }
}
template <bool PAIRS2 = true>
__device__ __forceinline__ void g2_inverse(float2* s, const float2* t_lo, const float2* t_hi, int tt) {
    if (PAIRS2) {
        g2_pairs2(s, tt);
        __syncthreads();
    }
    g2_pair<true, 1>(s, t_lo, t_hi, ((tt & 255) << 5) + (tt >> 8), tt >> 8);
    __syncthreads();
    g2_pair<true, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
    __syncthreads();
    g2_pair<true, 9>(s, t_lo, t_hi, tt, tt);
    __syncthreads();
}

// position <-> frequency of the forward transform's output order, and the position of frequency -f
__device__ __forceinline__ int g2_rev6(int x) {  // six base-4 digits reversed (12 bits)
    int r = (int)(__brev((unsigned)x) >> 20);
    return ((r & 0xAAA) >> 1) | ((r & 0x555) << 1);
}
__device__ __forceinline__ int g2_mirror(int p) {
    const int f = g2_rev6(p >> 1) | ((p & 1) << 12);
    const int g = (G2_N - f) & (G2_N - 1);
    return (g2_rev6(g & 4095) << 1) | (g >> 12);
}

// second-level spectra for the 8192-point form: out[(c * 257 + row) * G2_N + i]; grid (257, 1), both channels per workgroup
__global__ __launch_bounds__(G2_THREADS) void k_g2_ir(const float4* __restrict__ H, int pstride, int P, float2* __restrict__ out) {
    __shared__ float2 s[2][G2_LDS];
    __shared__ float2 t_lo[128], t_hi[64];
    const int row = blockIdx.x, bin = row == 256 ? 0 : row;
    const int c = threadIdx.x >> 9, tt = threadIdx.x & 511;
    g2_tables(t_lo, t_hi);
    for (int i = tt; i < G2_N; i += 512) {
        float2 v = make_float2(0.f, 0.f);
        if (i < P) {
            const float4 h = H[(size_t)bin * pstride + i];
            const float2 hc = c == 0 ? make_float2(h.x, h.y) : make_float2(h.z, h.w);
            if (bin != 0)
                v = hc;
            else
                v = make_float2(row == 0 ? 0.5f * (hc.x + hc.y) : 0.5f * (hc.x - hc.y), 0.f);
        }
        s[c][G2_P(i)] = v;
    }
    __syncthreads();
    g2_forward(s[c], t_lo, t_hi, tt);
    float2* dst = out + ((size_t)c * 257 + row) * G2_N;
    for (int i = tt; i < G2_N; i += 512) dst[i] = s[c][G2_P(i)];
}

// ---------------------------------------------------------------------------
// k_g2_mac: the fused second-level transform with TWO workgroups per CU.
// The one-workgroup form above keeps both inputs' sequences in LDS side by side (133 KB): its phases - window fill,
// forward transforms, products (the spectra stream in), inverse transforms, store - cannot overlap, and its counters
// say so (profiles/r2_headline_summary.md: waves parked at s_waitcnt / barriers half of their time, VALU 43 % busy,
// LDS 26 %, 2.4 of 8 TB/s).  Here a workgroup of 512 threads owns ONE 8192-point LDS buffer (67.6 KB) and runs the two
// sequences of its item through it one after the other; what has to wait for the buffer waits in registers:
//   window -> x1 to LDS, x2 held (16 entries per thread) -> forward(x1) -> own spectrum entries of X1 to registers
//   -> x2 to LDS -> forward(x2) -> products: X1 from registers, X2 and Y_L in place in LDS, Y_R into X1's registers
//   -> inverse(Y_L) -> its valid outputs to registers -> Y_R to LDS -> inverse(Y_R) -> {Y_L, Y_R} stored 16 bytes per block.
// Two such workgroups share a CU (2 x 69 KB of LDS, 2 x 8 waves at <= 128 VGPRs), each in its own phase: one's memory
// phases run under the other's transforms.  Same arithmetic, same order of operations per entry as the one-workgroup
// form, except for bin 0.
// Bin 0 packs {DC, Nyquist} as one complex number z and needs y = h1 * z + h2 * conj(z).  The one-workgroup form
// takes the spectrum of conj(z) from mirrored entries of the spectrum of z, which here sit in other threads'
// registers; instead the item runs twice - inputs z with rows h1, then inputs conj(z) (conjugated as the window is
// read) with rows h2 (row 256), the second run adding to the sums the first one stored.  One item in 256.
// Bounds: as for the one-workgroup form above (items, window, spectra rows, sums); LDS: G2_P(8191) < G2_LDS.
// ---------------------------------------------------------------------------
#define G2B_THREADS 512
// The 8/2 pass in a wave-local mapping: lane l of wave w works on parity l >> 5 of group 32 w + (l & 31) - the 32 groups
// (1024 consecutive elements) that the same wave's 128/32 pass reads and writes.  The two passes therefore hand over
// inside a wave: LDS operations of one wave execute in order, so no workgroup barrier sits between them (#if 0: the
// barrier form, same mapping) and the waves of a workgroup drift apart instead of meeting 4 more times per item.
// Consecutive lanes are 33 entries apart as before: conflict-free.
#define G2B_POS1(tt) (((32 * ((tt) >> 6) + ((tt) & 31)) << 5) + (((tt) >> 5) & 1))
#ifndef G2B_WAVE_LOCAL
#define G2B_WAVE_LOCAL 1
#endif
#if G2B_WAVE_LOCAL
#define G2B_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define G2B_WAVE_SYNC() __syncthreads()
#endif
#ifndef G2B_AHEAD
#define G2B_AHEAD 1  // entry pairs by which the spectra loads run ahead of the products (4 loads of 16 bytes each)
#endif
#ifndef G2B_FILL
#define G2B_FILL 16  // window rows requested together (all of them: one round of memory latency)
#endif
__global__ __launch_bounds__(G2B_THREADS, 4) void k_g2_mac(const float4* __restrict__ fdl, int ring, int slot0, int T, int chunk_t,
                                                           int taps, Fft2Voices vv, float4* __restrict__ Yc, int ycap, int nitems,
                                                           CorrArgs ca, int main_grid) {
    __shared__ float2 s[G2_LDS];
    __shared__ float2 t_lo[128], t_hi[64];
    if ((int)blockIdx.x >= main_grid) {  // a workgroup that rides along: Q1/Q2 terms of 256 blocks (see CorrArgs)
        corr_terms_body((int)blockIdx.x - main_grid, reinterpret_cast<double(*)[4]>(s), ca);
        return;
    }
    const int nch = nitems >> 8;
#ifndef G2_PRIO
#define G2_PRIO 3
#endif
#if G2_PRIO  // The two workgroups of a CU sit in wave slots {0, 1} and {2, 3} of every SIMD (scripts/probes/cuid_probe.hip).  The upper pair gets issue
             // priority: whenever both want to issue, one of them goes first every time instead of taking turns - 367-372 -> 353-354 us per headline
             // launch, -2 ... -4 % for configs 2, 5 and the shipped point, same bits (profiles/r3_g2_ablation.md, section 7; -DG2_PRIO=0: without).
             // Only for launches of at least four rounds of workgroups: in a short one the disadvantaged workgroup of the last round IS the
             // launch's tail (8192 blocks: 47 -> 55 us, 32320: 105 -> 113 with it)
    if (nitems >= 2048) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid & 0xfu) >= 2u) __builtin_amdgcn_s_setprio(G2_PRIO);
    }
#endif
    g2_tables(t_lo, t_hi);
    __syncthreads();  // the first forward pass runs on registers: nothing else orders its twiddle reads after the tables
#if G2_STAMPS  // diagnostic build only: where a workgroup's time goes (s_memtime ticks = shader cycles)
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0, st_now;
    int st_items = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#define G2_STAMP(k)                                                                       \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if ((k) >= 0) st_acc[(k) < 0 ? 0 : (k)] += st_now - st_prev;                      \
        st_prev = st_now;                                                                 \
    } while (0)
#else
#define G2_STAMP(k) do { } while (0)
#endif
    constexpr int ROWS = G2_N / G2B_THREADS;        // 16 window entries per thread and sequence
    constexpr int PS = 2 * G2B_THREADS + 2 * G2B_THREADS / 32;  // ... of entry pairs 2 j, 2 (j + 512)
    for (int item = blockIdx.x; item < nitems; item += main_grid) {
        const int xq = item >> 3;
        const int bin = (xq / nch) * 8 + (item & 7), chunk = xq % nch;
        const int t_c0 = chunk * chunk_t, nout = min(chunk_t, T - t_c0), L = nout + taps - 1;
        const float4* fk = fdl + (size_t)bin * ring;
#ifdef G2_ALIAS_TEST  // timing only (wrong data): every bin's window starts 1 KB further - does the 1 MB bin stride alias in HBM?
        const int sb = slot0 + t_c0 - (taps - 1) + 64 * bin;
#else
        const int sb = slot0 + t_c0 - (taps - 1);
#endif
        for (int pass = 0; pass < (bin == 0 ? 2 : 1); pass++) {
            const float cj = pass ? -1.0f : 1.0f;  // second run of bin 0: conj(z)
            const int row = pass ? 256 : bin;
            int tt = threadIdx.x;
            asm volatile("" : "+v"(tt));
            G2_STAMP(-1);
            // ---- window: 16 bytes per slot carry both inputs.  The rows a thread loads, n = tt + 512 r, are exactly its 16
            // elements of the first forward pass: that pass runs on the registers (x1 at once, x2 when the buffer is free)
            v2f x1[ROWS], x2[ROWS];
            {
#pragma unroll
                for (int r0 = 0; r0 < ROWS; r0 += G2B_FILL) {
                    float4 x[G2B_FILL];
#pragma unroll
                    for (int r = 0; r < G2B_FILL; r++) {
                        const int n = tt + G2B_THREADS * (r0 + r);
                        x[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#if G2_ABL & 1
                        if (n < L) x[r] = make_float4(__int_as_float(0x3c000000 | n), 0.25f, -0.5f, __int_as_float(0x3c800000 | n));
#else
                        if (n < L) x[r] = fk[(sb + n) & (ring - 1)];
#endif
                    }
#pragma unroll
                    for (int r = 0; r < G2B_FILL; r++) {
                        x1[r0 + r] = v2f{x[r].x, cj * x[r].y};
                        x2[r0 + r] = v2f{x[r].z, cj * x[r].w};
                    }
                }
            }
            G2_STAMP(0);
            g2_pair<false, 9, false, true>(s, t_lo, t_hi, tt, tt, x1);  // quarter lengths 2048, 512
            __syncthreads();
            g2_pair<false, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);  // 128, 32
            G2B_WAVE_SYNC();  // a wave's 1024 elements of this pass are the 32 groups its last pass works on
            g2_pair<false, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);  // 8, 2
            __syncthreads();
            G2_STAMP(1);
            // ---- own entries of X1 (pairs 2 j, 2 j + 1, j = tt + 512 r; the radix-2 stage on the way) to registers
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
            __syncthreads();  // every thread has its X1 entries: the buffer is free for x2
            G2_STAMP(2);
            g2_pair<false, 9, false, true>(s, t_lo, t_hi, tt, tt, x2);
            __syncthreads();
            g2_pair<false, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
            G2B_WAVE_SYNC();
            g2_pair<false, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
            __syncthreads();
            G2_STAMP(3);
            // ---- products: Y_c = sum_voices g (X1 H1c + X2 H2c) on the thread's own entries; Y_L replaces X2 in LDS,
            // Y_R replaces X1 in registers.  The first voice's spectra run G2B_AHEAD entry pairs ahead of the arithmetic
            // (4 loads of 16 bytes per pair, a ring of G2B_AHEAD + 1 pairs in registers).
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
#if G2_ABL & 1
                        HLq[r % RING][i] = make_float4(__int_as_float(0x3c000000 | off), 0.5f, 0.25f, __int_as_float(0x3c400000 | off));
                        HRq[r % RING][i] = make_float4(0.5f, __int_as_float(0x3c000000 | off), __int_as_float(0x3c400000 | off), 0.25f);
#else
                        HLq[r % RING][i] = *reinterpret_cast<const float4*>(hrow[i] + off);
                        HRq[r % RING][i] = *reinterpret_cast<const float4*>(hrow[i] + (size_t)257 * G2_N * sizeof(float2) + off);
#endif
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
                    __builtin_amdgcn_sched_barrier(0);  // requests stay G2B_AHEAD pairs ahead, no further (registers)
                }
            }
            __syncthreads();
            G2_STAMP(4);
            // ---- inverse of Y_L; its last pass leaves element tt + 512 m = output block tt + 512 m - (taps - 1) in registers
            v2f yl[ROWS];
            g2_pair<true, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
            G2B_WAVE_SYNC();
            g2_pair<true, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
            __syncthreads();
            g2_pair<true, 9, true, false>(s, t_lo, t_hi, tt, tt, yl);
            G2_STAMP(5);
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
            G2_STAMP(6);
            {
                v2f yr[ROWS];
                g2_pair<true, 1>(s, t_lo, t_hi, G2B_POS1(tt), (tt >> 5) & 1);
                G2B_WAVE_SYNC();
                g2_pair<true, 5>(s, t_lo, t_hi, ((tt >> 5) << 9) + (tt & 31), tt & 31);
                __syncthreads();
                g2_pair<true, 9, true, false>(s, t_lo, t_hi, tt, tt, yr);
                G2_STAMP(7);
                // the valid part of the circle: its first taps - 1 outputs are discarded
                asm volatile("" : "+v"(tt));
                const float sc = 1.0f / (float)G2_N;
                float4* dst = Yc + (size_t)bin * ycap + t_c0;
#pragma unroll
                for (int m = 0; m < ROWS; m++) {
                    const int t = tt + G2B_THREADS * m - (taps - 1);
#if G2_ABL & 1
                    if (t >= 0 && t < nout && ycap < 0) {
#else
                    if (t >= 0 && t < nout) {
#endif
                        float4 y = make_float4(yl[m].x * sc, yl[m].y * sc, yr[m].x * sc, yr[m].y * sc);  // (second run of bin 0: h2 * conj z)
                        if (pass) {
                            const float4 o = dst[t];
                            y = make_float4(o.x + y.x, o.y + y.y, o.z + y.z, o.w + y.w);
                        }
                        dst[t] = y;
                    }
                }
            }
            __syncthreads();  // the buffer is free for the next run
            G2_STAMP(8);
#if G2_STAMPS
            st_items++;
#endif
        }
    }
#if G2_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x == 3 || blockIdx.x == 137 || blockIdx.x == 300 || blockIdx.x == 700 || blockIdx.x == 1100)) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - st_t0, dr = __builtin_amdgcn_s_memrealtime() - st_r0;
        printf("g2 wg %d: %llu cycles in %llu x 10 ns = %.0f MHz\n", (int)blockIdx.x, dt, dr, (double)dt / (double)dr * 100.0);
    }
    if (threadIdx.x == 0 && (blockIdx.x == 3 || blockIdx.x == 137 || blockIdx.x == 300 || blockIdx.x == 700 || blockIdx.x == 1100))
        printf("g2 wg %d items %d: fill %llu fwd1 %llu x1regs+x2fill %llu fwd2 %llu products %llu inv1 %llu out1+yrfill %llu inv2 %llu store %llu\n",
               (int)blockIdx.x, st_items, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5], st_acc[6], st_acc[7], st_acc[8]);
#endif
}

#ifdef MCCONV_LAB
#include "lab_kernels.hip.h"
#endif
