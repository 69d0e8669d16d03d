// fft512.hip.h — one 512-point complex FFT per 64-lane wavefront (gfx950).
//
// Replaces the cuFFT C2C calls of the reference (conv.cu:243, 367, 405, 407)
// for the partition size of the uniform-partitioned engine.  512 = 8 x 8 x 8:
// each lane holds 8 points in registers, does a radix-8 butterfly, and the two
// digit exchanges between the three stages go through a wave-private LDS tile
// whose rows are padded to 72 complex so that the transposed ds_read_b64 of a
// 32-lane group lands on 32 distinct bank pairs.  Twiddles come from a 512-entry
// table staged in LDS once per workgroup (computed in double on the host).
//
// Index algebra (n = 64 n2 + 8 n1 + n0, k = k0 + 8 k1 + 64 k2):
//   X[k] = sum_n0 w8^(n0 k2) w64^(n0 k1) [ sum_n1 w8^(n1 k1) w512^((8 n1 + n0) k0)
//            [ sum_n2 w8^(n2 k0) x[n] ] ]
#pragma once
#include <hip/hip_runtime.h>

#define FFT_N 512
#define FFT_ROW 72                       // padded row length (complex) of the exchange tile
#define FFT_WAVE_LDS (8 * FFT_ROW)       // complex entries of wave-private LDS (>= 512)

// Complex arithmetic on two-element vectors; the products and the +-j rotations are spelled out as the packed
// instructions they are (operand swizzles and sign modifiers of v_pk_mul / v_pk_fma / v_pk_add) - from scalar
// expressions the compiler re-packs them through register moves (262 instead of 144 VALU instructions for a
// radix-16 pass of the second-level transforms, kernels.hip.h).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f vx_of(float2 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ float2 vx_to(v2f a) { return make_float2(a.x, a.y); }
// a b = (a.x b.x - a.y b.y, a.x b.y + a.y b.x): both instructions in ONE asm statement - between two statements the
// compiler pads a wait state (s_nop) whenever the second reads what the first wrote; inside a statement the
// hardware's own VALU interlock orders them.  r is written by the first instruction while a and b are still needed:
// early clobber.
__device__ __forceinline__ v2f vx_mul(v2f a, v2f b) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}
// a conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
__device__ __forceinline__ v2f vx_mulc(v2f a, v2f b) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}
// a - j b = (a.x + b.y, a.y - b.x) and a + j b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2f vx_sub_j(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f vx_add_j(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + DIR j b (DIR = -1: forward transform, +1: inverse)
template <int DIR>
__device__ __forceinline__ v2f vx_rot(v2f a, v2f b) { return DIR < 0 ? vx_sub_j(a, b) : vx_add_j(a, b); }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return vx_to(vx_mul(vx_of(a), vx_of(b))); }
// multiply by -j (forward) or +j (inverse)
template <int DIR>
__device__ __forceinline__ float2 mulj(float2 a) {
    return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

// 8-point DFT in registers, y[k] = sum_r v[r] exp(DIR * 2 pi i r k / 8)
template <int DIR>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    const float h = 0.70710678118654752440f;
    v2f x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = vx_of(v[r]);
    // even / odd 4-point DFTs (w4 = DIR j)
    const v2f e0 = x[0] + x[4], e1 = x[0] - x[4], e2 = x[2] + x[6], d2 = x[2] - x[6];
    const v2f E0 = e0 + e2, E2 = e0 - e2, E1 = vx_rot<DIR>(e1, d2), E3 = vx_rot<-DIR>(e1, d2);
    const v2f o0 = x[1] + x[5], o1 = x[1] - x[5], o2 = x[3] + x[7], d3 = x[3] - x[7];
    const v2f O0 = o0 + o2, O2 = o0 - o2, O1 = vx_rot<DIR>(o1, d3), O3 = vx_rot<-DIR>(o1, d3);
    // twiddles w8^k, k = 1..3: (1 + DIR j)/sqrt2, DIR j, (-1 + DIR j)/sqrt2
    const v2f t1 = vx_rot<DIR>(O1, O1);   // O1 (1 + DIR j), scaled by h below
    const v2f t3 = vx_rot<-DIR>(O3, O3);  // O3 (1 - DIR j) = -O3 (-1 + DIR j), scaled by -h below
    x[0] = E0 + O0;
    x[4] = E0 - O0;
    x[1] = E1 + h * t1;
    x[5] = E1 - h * t1;
    x[2] = vx_rot<DIR>(E2, O2);
    x[6] = vx_rot<-DIR>(E2, O2);
    x[3] = E3 - h * t3;
    x[7] = E3 + h * t3;
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = vx_to(x[r]);
}

template <int DIR>
__device__ __forceinline__ float2 twiddle(const float2* tw, int idx) {
    float2 w = tw[idx & (FFT_N - 1)];
    if (DIR > 0) w.y = -w.y;
    return w;
}

// Ordering point between the LDS writes and reads of one exchange.  WG = true:
// workgroup barrier (every wave of the workgroup runs the transform the same
// number of times).  WG = false: only this wave takes part — LDS operations of
// one wave execute in order, so a compiler-level fence plus a drained LDS queue
// is enough and other waves may do something else meanwhile.
template <bool WG>
__device__ __forceinline__ void fft_sync() {
    if (WG) {
        __syncthreads();
    } else {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// In: v[r] = x[lane + 64 r].  Out: X[k] in natural order in lds[0..511].
// tw = LDS table exp(-2 pi i m / 512).  `lds` is wave-private (FFT_WAVE_LDS
// entries).
template <int DIR, bool WG = true>
__device__ __forceinline__ void fft512_wave(float2 (&v)[8], float2* lds, const float2* tw, int lane) {
    // stage 1: DFT over n2, twiddle w512^(lane * k0)
    dft8<DIR>(v);
#pragma unroll
    for (int k0 = 1; k0 < 8; k0++) v[k0] = cmul(v[k0], twiddle<DIR>(tw, lane * k0));
#pragma unroll
    for (int k0 = 0; k0 < 8; k0++) lds[k0 * FFT_ROW + lane] = v[k0];
    fft_sync<WG>();
    const int a = lane >> 3, n0 = lane & 7;
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) v[n1] = lds[a * FFT_ROW + n1 * 8 + n0];
    fft_sync<WG>();
    // stage 2: DFT over n1, twiddle w64^(n0 * k1)
    dft8<DIR>(v);
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) v[k1] = cmul(v[k1], twiddle<DIR>(tw, 8 * n0 * k1));
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) lds[a * FFT_ROW + k1 * 8 + n0] = v[k1];
    fft_sync<WG>();
    // lane = 8 k0 + k1 now reads its 8 consecutive n0 values
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = lds[a * FFT_ROW + n0 * 8 + m];
    fft_sync<WG>();
    // stage 3: DFT over n0 -> X[k0 + 8 k1 + 64 k2]
    dft8<DIR>(v);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) lds[a + 8 * n0 + 64 * k2] = v[k2];
    fft_sync<WG>();
}

// stage the twiddle table into LDS (all threads of the workgroup)
__device__ __forceinline__ void load_twiddles(float2* s_tw, const float2* __restrict__ g_tw) {
    for (int i = threadIdx.x; i < FFT_N; i += blockDim.x) s_tw[i] = g_tw[i];
}
