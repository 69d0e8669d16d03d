/*
 * oracle.c — CPU restatement of the limitz/cuda-audio convolution hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED (no reference
 * fixtures exist; see oracle.h header).
 *
 * Written from the behaviour of /root/reference/src/conv.cu, conv.h, wav.cu;
 * no reference text is reproduced.  Each function cites the lines it follows.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    double x, y;
} cplx;

static inline cplx c_add(cplx a, cplx b) { return (cplx){a.x + b.x, a.y + b.y}; }
static inline cplx c_sub(cplx a, cplx b) { return (cplx){a.x - b.x, a.y - b.y}; }
static inline cplx c_scale(cplx a, double s) { return (cplx){a.x * s, a.y * s}; }
static inline cplx c_conj(cplx a) { return (cplx){a.x, -a.y}; }
static inline cplx c_mul(cplx a, cplx b) { return (cplx){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
static inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_cc_defaults(orc_cc_value *v) {
    /* src/conv.h:38-49 */
    v->select = 0;
    v->predelay = 0;
    v->speed = 100;
    v->vsteps = 0;
    v->dry = 0.5f;
    v->wet = 0.5f;
    v->panDry = 0.0f;
    v->panWet = 0.0f;
    v->level = 1.0f;
}

/* ------------------------------------------------------------------ FFT -- */
/* Iterative radix-2 DIT; computes sum_n x[n] exp(sign*2*pi*i*n*k/N), i.e. the
 * unnormalised DFT cuFFT's C2C computes (conv.cu:243,367 forward = -1;
 * conv.cu:405,407 inverse = +1). */
void orc_fft(double *re, double *im, size_t n, int sign) {
    if (n < 2) return;
    /* bit reversal */
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i];
            re[i] = re[j];
            re[j] = t;
            t = im[i];
            im[i] = im[j];
            im[j] = t;
        }
    }
    /* twiddle table, computed directly (no recurrence) for accuracy */
    double *wr = (double *)malloc(sizeof(double) * (n / 2));
    double *wi = (double *)malloc(sizeof(double) * (n / 2));
    for (size_t k = 0; k < n / 2; k++) {
        double a = (double)sign * 2.0 * M_PI * (double)k / (double)n;
        wr[k] = cos(a);
        wi[k] = sin(a);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1, step = n / len;
        for (size_t i = 0; i < n; i += len) {
            for (size_t k = 0; k < half; k++) {
                double cr = wr[k * step], ci = wi[k * step];
                size_t a = i + k, b = a + half;
                double tr = re[b] * cr - im[b] * ci;
                double ti = re[b] * ci + im[b] * cr;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
        }
    }
    free(wr);
    free(wi);
}

static void fft_c(cplx *v, size_t n, int sign) {
    double *re = (double *)malloc(sizeof(double) * n);
    double *im = (double *)malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; i++) {
        re[i] = v[i].x;
        im[i] = v[i].y;
    }
    orc_fft(re, im, n, sign);
    for (size_t i = 0; i < n; i++) {
        v[i].x = re[i];
        v[i].y = im[i];
    }
    free(re);
    free(im);
}

/* --------------------------------------------------------- direct conv --- */
void orc_direct_conv(const float *x, size_t nx, const float *h, size_t nh, double *y) {
    memset(y, 0, sizeof(double) * (nx + nh - 1));
    for (size_t i = 0; i < nx; i++) {
        double xv = x[i];
        for (size_t j = 0; j < nh; j++) y[i + j] += xv * (double)h[j];
    }
}

/* ----------------------------------------------------------- wav scale --- */
void orc_wav_decode_s16(const int16_t *lr, size_t frames, float *out) {
    /* wav.cu:17-29: v / 65536 per channel (full scale = +-0.5, Q5) */
    for (size_t i = 0; i < 2 * frames; i++) out[i] = (float)lr[i] / 65536.0f;
}

void orc_wav_decode_s24(const uint8_t *b, size_t frames, float *out) {
    /* wav.cu:30-57: bytes -> top 24 bits of an int32, /256 (sign kept), / 2^24 */
    for (size_t i = 0; i < 2 * frames; i++) {
        uint32_t v = ((uint32_t)b[3 * i] << 8) | ((uint32_t)b[3 * i + 1] << 16) | ((uint32_t)b[3 * i + 2] << 24);
        int32_t s = (int32_t)v;
        s /= 256;
        out[i] = (float)s / 16777216.0f;
    }
}

/* ----------------------------------------------------------- handleCC ---- */
void orc_handle_cc(orc_cc_value *v, const uint8_t m[8], uint8_t m2, int val, size_t nb) {
    /* conv.cu:255-276; m = {select,predelay,dry,wet,speed,panDry,panWet,level} */
    if (m[0] == m2) {
        v->select = (uint64_t)val * nb / 0x80;
        v->vsteps = v->speed;
    }
    if (m[1] == m2) v->predelay = (uint64_t)val * ORC_MAX_PREDELAY / 0x80;
    if (m[2] == m2) v->dry = val / 128.0f;
    if (m[3] == m2) v->wet = val / 128.0f;
    if (m[5] == m2) v->panDry = val / 64.0f - 1;
    if (m[6] == m2) v->panWet = val / 64.0f - 1;
    if (m[7] == m2) v->level = val / 128.0f;
    if (m[4] == m2) {
        v->speed = ((uint64_t)val * 1024) / 0x80;
        if (v->vsteps > v->speed) v->vsteps = v->speed;
    }
}

static inline double pan_l(double p) { return p >= 0 ? 1 - p : 1; } /* conv.cu:386,388 */
static inline double pan_r(double p) { return p <= 0 ? 1 + p : 1; } /* conv.cu:387,389 */

/* ============================================================ refcompat === */
#define ORC_MAX_IRS 256

struct orc_ref {
    size_t N;
    int three_mult;
    orc_cc_value cc[2];
    /* conv.cu:155-179 buffers; all zero at start (Q6) */
    cplx *cin, *cin1, *cin2, *cinFFT;
    cplx *irFFT[2][2];  /* [half][L/R], N each */
    cplx *outp[2];      /* output.left/right, N + 8192 */
    cplx *resid[2];     /* residual.left/right, N + 8192 */
    cplx *tmp[2];       /* ir.left/right used as IFFT scratch (conv.cu:403) */
    cplx *irbuf[ORC_MAX_IRS]; /* _irBuffers: [H_L (N) | H_R (N)] */
    double sums[ORC_MAX_IRS][4];
    size_t nirs;
};

/* f_unpackC22R, conv.cu:47-73 — two-for-one split incl. the s==0 shortcut (Q1)
 * and the never-written N/2 entry (Q2). */
static void ref_unpack(cplx *L, cplx *R, const cplx *src, size_t N) {
    for (size_t s = 0; s < N / 2; s++) {
        size_t ia = s, ib = N - s;
        cplx va = src[ia];
        cplx vb = s ? c_conj(src[ib]) : va;
        cplx la = c_scale(c_add(va, vb), 0.5);
        cplx d = c_scale(c_sub(va, vb), -0.5);
        cplx lb = (cplx){-d.y, d.x}; /* timesj */
        L[ia] = la;
        R[ia] = lb;
        if (s) {
            L[ib] = c_conj(la);
            R[ib] = c_conj(lb);
        }
    }
}

orc_ref *orc_ref_create(size_t N, int three_mult) {
    orc_ref *r = (orc_ref *)calloc(1, sizeof(orc_ref));
    r->N = N;
    r->three_mult = three_mult;
    orc_cc_defaults(&r->cc[0]);
    orc_cc_defaults(&r->cc[1]);
    r->cin = (cplx *)calloc(N, sizeof(cplx));
    r->cin1 = (cplx *)calloc(N, sizeof(cplx));
    r->cin2 = (cplx *)calloc(N, sizeof(cplx));
    r->cinFFT = (cplx *)calloc(N, sizeof(cplx));
    for (int i = 0; i < 2; i++) {
        for (int c = 0; c < 2; c++) r->irFFT[i][c] = (cplx *)calloc(N, sizeof(cplx));
        r->outp[i] = (cplx *)calloc(N + ORC_MAX_PREDELAY, sizeof(cplx));
        r->resid[i] = (cplx *)calloc(N + ORC_MAX_PREDELAY, sizeof(cplx));
        r->tmp[i] = (cplx *)calloc(N, sizeof(cplx));
    }
    return r;
}

void orc_ref_destroy(orc_ref *r) {
    if (!r) return;
    free(r->cin);
    free(r->cin1);
    free(r->cin2);
    free(r->cinFFT);
    for (int i = 0; i < 2; i++) {
        for (int c = 0; c < 2; c++) free(r->irFFT[i][c]);
        free(r->outp[i]);
        free(r->resid[i]);
        free(r->tmp[i]);
    }
    for (size_t j = 0; j < ORC_MAX_IRS; j++) free(r->irbuf[j]);
    free(r);
}

orc_cc_value *orc_ref_cc(orc_ref *r, int half) { return &r->cc[half & 1]; }
size_t orc_ref_num_irs(const orc_ref *r) { return r->nirs; }

int orc_ref_prepare(orc_ref *r, size_t idx, const float *lr, size_t frames, size_t nframes) {
    /* conv.cu:207-253 */
    if (idx >= ORC_MAX_IRS || nframes >= r->N) return -1;
    size_t N = r->N;
    cplx *tmp = (cplx *)calloc(N, sizeof(cplx));           /* :223-227 zeroed scratch */
    cplx *buf = (cplx *)calloc(2 * N, sizeof(cplx));       /* :233 (Q6: taken as zero) */
    size_t n = frames < N - nframes ? frames : N - nframes; /* :239 truncation */
    double sl = 0, sr = 0, al = 0, ar = 0;
    for (size_t s = 0; s < n; s++) {                       /* :240 L->re, R->im */
        tmp[s].x = lr[2 * s];
        tmp[s].y = lr[2 * s + 1];
        double sg = (s & 1) ? -1.0 : 1.0;
        sl += lr[2 * s];
        sr += lr[2 * s + 1];
        al += sg * lr[2 * s];
        ar += sg * lr[2 * s + 1];
    }
    fft_c(tmp, N, -1);                                     /* :243 */
    ref_unpack(buf, buf + N, tmp, N);                      /* :246 */
    free(tmp);
    if (!r->irbuf[idx]) {
        if (idx + 1 > r->nirs) r->nirs = idx + 1;
    } else {
        free(r->irbuf[idx]);                               /* :214-215 */
    }
    r->irbuf[idx] = buf;
    r->sums[idx][0] = sl;
    r->sums[idx][1] = sr;
    r->sums[idx][2] = al;
    r->sums[idx][3] = ar;
    return 0;
}

int orc_ref_ir_sums(const orc_ref *r, size_t idx, double out[4]) {
    if (idx >= ORC_MAX_IRS || !r->irbuf[idx]) return -1;
    memcpy(out, r->sums[idx], sizeof(double) * 4);
    return 0;
}

/* f_interpolate, conv.cu:15-32 */
static void ref_interpolate(cplx *dst, const cplx *b, size_t N, size_t steps, double wet) {
    double div = (double)(steps + 5);
    for (size_t s = 0; s < N / 2; s++) {
        cplx va = dst[s];
        cplx vb = c_scale(b[s], wet);
        cplx vd = (cplx){(vb.x - va.x) / div, (vb.y - va.y) / div};
        cplx vv = c_add(va, vd);
        dst[s] = vv;
        if (s) dst[N - s] = c_conj(vv);
    }
}

/* f_pointwiseMultiplyAndScale, conv.cu:102-123 */
static void ref_mac(cplx *r, const cplx *ir1, const cplx *ir2, const cplx *a1, const cplx *a2, size_t n,
                    double s1, double s2, int three_mult) {
    for (size_t s = 0; s < n; s++) {
        cplx p1, p2;
        if (three_mult) {
            double re1 = a1[s].x * ir1[s].x - a1[s].y * ir1[s].y;
            double re2 = a2[s].x * ir2[s].x - a2[s].y * ir2[s].y;
            double im1 = (a1[s].x + a1[s].y) * (ir1[s].x + ir1[s].y) - re1;
            double im2 = (a2[s].x + a2[s].y) * (ir2[s].x + ir2[s].y) - re2;
            p1 = (cplx){re1, im1};
            p2 = (cplx){re2, im2};
        } else {
            p1 = c_mul(a1[s], ir1[s]);
            p2 = c_mul(a2[s], ir2[s]);
        }
        r[s] = c_add(c_scale(p1, s1), c_scale(p2, s2));
    }
}

void orc_ref_process(orc_ref *r, const float *in1, const float *in2, double *outL, double *outR,
                     size_t nframes) {
    /* Convolution::onProcess, conv.cu:287-466 */
    size_t N = r->N;
    orc_cc_value *c0 = &r->cc[0], *c1 = &r->cc[1];

    /* :321-328 pack z = in1 + j in2, zero padded */
    memset(r->cin, 0, sizeof(cplx) * N);
    for (size_t s = 0; s < nframes; s++) r->cin[s] = (cplx){in1[s], in2[s]};

    /* :339-353 interpolate live IR spectra towards wet * selected IR (Q7) */
    const cplx *b0 = r->irbuf[c0->select], *b1 = r->irbuf[c1->select];
    ref_interpolate(r->irFFT[0][0], b0, N, c0->vsteps, c0->wet);
    ref_interpolate(r->irFFT[0][1], b0 + N, N, c0->vsteps, c0->wet);
    if (c0->vsteps > 0) c0->vsteps--;
    ref_interpolate(r->irFFT[1][0], b1, N, c1->vsteps, c1->wet);
    ref_interpolate(r->irFFT[1][1], b1 + N, N, c1->vsteps, c1->wet);
    if (c1->vsteps > 0) c1->vsteps--;

    /* :367-371 forward FFT + two-for-one unpack into cin1, cin2 */
    memcpy(r->cinFFT, r->cin, sizeof(cplx) * N);
    fft_c(r->cinFFT, N, -1);
    ref_unpack(r->cin1, r->cin2, r->cinFFT, N);

    /* :386-401 pans, MAC with 1/N * pan * level */
    double panL1 = pan_l(c0->panWet), panR1 = pan_r(c0->panWet);
    double panL2 = pan_l(c1->panWet), panR2 = pan_r(c1->panWet);
    double invN = 1.0 / (double)N;
    ref_mac(r->outp[0], r->irFFT[0][0], r->irFFT[1][0], r->cin1, r->cin2, N, invN * panL1 * c0->level,
            invN * panL2 * c1->level, r->three_mult);
    ref_mac(r->outp[1], r->irFFT[0][1], r->irFFT[1][1], r->cin1, r->cin2, N, invN * panR1 * c0->level,
            invN * panR2 * c1->level, r->three_mult);

    /* :403-408 inverse FFTs into the scratch pair */
    for (int c = 0; c < 2; c++) {
        memcpy(r->tmp[c], r->outp[c], sizeof(cplx) * N);
        fft_c(r->tmp[c], N, +1);
    }

    /* :411-415 residual + predelay-shifted block, clamp per component (Q4, Q8);
     * predelay of half 0 for both channels */
    size_t pd = c0->predelay;
    for (int c = 0; c < 2; c++) {
        for (size_t s = 0; s < N; s++) {
            cplx v = r->resid[c][s];
            if (s >= pd) v = c_add(v, r->tmp[c][s - pd]);
            r->outp[c][s] = (cplx){clampd(v.x, -1, 1), clampd(v.y, -1, 1)};
        }
    }

    /* :418-427 dry mix, scalar added to both components (operators.h:338) */
    panL1 = pan_l(c0->panDry);
    panR1 = pan_r(c0->panDry);
    panL2 = pan_l(c1->panDry);
    panR2 = pan_r(c1->panDry);
    double sL1 = (double)c0->dry * panL1 * c0->level, sR1 = (double)c0->dry * panR1 * c0->level;
    double sL2 = (double)c1->dry * panL2 * c1->level, sR2 = (double)c1->dry * panR2 * c1->level;
    for (size_t s = 0; s < nframes; s++) {
        cplx v = r->cin[s];
        double aL = v.x * sL1 + v.y * sL2, aR = v.x * sR1 + v.y * sR2;
        r->outp[0][s].x += aL;
        r->outp[0][s].y += aL;
        r->outp[1][s].x += aR;
        r->outp[1][s].y += aR;
    }

    /* :431-437 .x of the first nframes samples goes to JACK */
    for (size_t s = 0; s < nframes; s++) {
        outL[s] = r->outp[0][s].x;
        outR[s] = r->outp[1][s].x;
    }

    /* :440-451 slide the accumulator by nframes */
    size_t len = N + ORC_MAX_PREDELAY - nframes;
    for (int c = 0; c < 2; c++) memmove(r->resid[c], r->outp[c] + nframes, sizeof(cplx) * len);
}

/* ================================================================ upols === */
#define UP_B ORC_BLOCK
#define UP_K (2 * ORC_BLOCK)
#define UP_BINS (ORC_BLOCK + 1)

typedef struct {
    size_t P, taps;
    cplx *H[2];      /* [c][p*UP_BINS + k] */
    double sums[4];  /* sigma_L, sigma_R, alpha_L, alpha_R */
} up_ir;

struct orc_upols {
    size_t n_ref;
    int compat;
    orc_cc_value cc[2];
    up_ir ir[ORC_MAX_IRS];
    size_t nirs;
    double e[2];          /* cross-fade coefficient per half (Q7) */
    size_t t;             /* blocks processed */
    size_t cap;           /* capacity of the per-block history arrays */
    cplx *X[2];           /* [i][t*UP_BINS + k] all block spectra so far */
    double *G;            /* [t*4 + c*2 + i] wet gains attached to input block t */
    double *CD, *CQ;      /* [t*2 + c] prefix sums of D_c, Q_c */
    double *wet[2];       /* [c][tau] pre-delay wet stream of THIS shard, all samples so far */
    double *wsum[2];      /* [c][tau] wet stream summed over all shards (what the post stage reads) */
    size_t pb, pe;        /* partition shard [pb, pe); pe == 0 means all */
    int pending;          /* a partial block awaits its finish */
};

orc_upols *orc_upols_create(size_t n_ref, int compat) {
    orc_upols *u = (orc_upols *)calloc(1, sizeof(orc_upols));
    u->n_ref = n_ref;
    u->compat = compat;
    orc_cc_defaults(&u->cc[0]);
    orc_cc_defaults(&u->cc[1]);
    return u;
}

void orc_upols_destroy(orc_upols *u) {
    if (!u) return;
    for (size_t j = 0; j < ORC_MAX_IRS; j++) {
        free(u->ir[j].H[0]);
        free(u->ir[j].H[1]);
    }
    free(u->X[0]);
    free(u->X[1]);
    free(u->G);
    free(u->CD);
    free(u->CQ);
    free(u->wet[0]);
    free(u->wet[1]);
    free(u->wsum[0]);
    free(u->wsum[1]);
    free(u);
}

orc_cc_value *orc_upols_cc(orc_upols *u, int half) { return &u->cc[half & 1]; }

int orc_upols_prepare(orc_upols *u, size_t idx, const float *lr, size_t frames, size_t nframes) {
    /* same truncation as conv.cu:239; partitions of 256 taps, zero-padded to
     * 512 and transformed (true spectra: no Q1/Q2 here, those are corrections) */
    if (idx >= ORC_MAX_IRS || nframes >= u->n_ref) return -1;
    up_ir *ir = &u->ir[idx];
    free(ir->H[0]);
    free(ir->H[1]);
    size_t n = frames < u->n_ref - nframes ? frames : u->n_ref - nframes;
    ir->taps = n;
    ir->P = (n + UP_B - 1) / UP_B;
    if (ir->P == 0) ir->P = 1;
    for (int c = 0; c < 2; c++) ir->H[c] = (cplx *)calloc(ir->P * UP_BINS, sizeof(cplx));
    memset(ir->sums, 0, sizeof(ir->sums));
    for (size_t s = 0; s < n; s++) {
        double sg = (s & 1) ? -1.0 : 1.0;
        ir->sums[0] += lr[2 * s];
        ir->sums[1] += lr[2 * s + 1];
        ir->sums[2] += sg * lr[2 * s];
        ir->sums[3] += sg * lr[2 * s + 1];
    }
    cplx buf[UP_K];
    for (size_t p = 0; p < ir->P; p++) {
        for (int c = 0; c < 2; c++) {
            memset(buf, 0, sizeof(buf));
            for (size_t m = 0; m < UP_B; m++) {
                size_t s = p * UP_B + m;
                if (s < n) buf[m].x = lr[2 * s + c];
            }
            fft_c(buf, UP_K, -1);
            memcpy(ir->H[c] + p * UP_BINS, buf, sizeof(cplx) * UP_BINS);
        }
    }
    if (idx + 1 > u->nirs) u->nirs = idx + 1;
    return 0;
}

static void up_grow(orc_upols *u) {
    if (u->t + 2 < u->cap) return;
    size_t nc = u->cap ? u->cap * 2 : 64;
    for (int i = 0; i < 2; i++) {
        u->X[i] = (cplx *)realloc(u->X[i], sizeof(cplx) * nc * UP_BINS);
        u->wet[i] = (double *)realloc(u->wet[i], sizeof(double) * nc * UP_B);
        memset(u->wet[i] + u->cap * UP_B, 0, sizeof(double) * (nc - u->cap) * UP_B);
        u->wsum[i] = (double *)realloc(u->wsum[i], sizeof(double) * nc * UP_B);
        memset(u->wsum[i] + u->cap * UP_B, 0, sizeof(double) * (nc - u->cap) * UP_B);
    }
    u->G = (double *)realloc(u->G, sizeof(double) * nc * 4);
    u->CD = (double *)realloc(u->CD, sizeof(double) * nc * 2);
    u->CQ = (double *)realloc(u->CQ, sizeof(double) * nc * 2);
    u->cap = nc;
}

static inline double prefix_at(const double *C, long t, int c) { return t < 0 ? 0.0 : C[(size_t)t * 2 + c]; }

static inline long floordiv(long a, long b) {
    long q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

void orc_upols_set_shard(orc_upols *u, size_t pb, size_t pe) {
    u->pb = pb;
    u->pe = pe;
}

/* first half of a block: everything up to this shard's share of the wet block
 * (pre-predelay), SURVEY Appendix B / §8(e).  This model assumes constant select
 * and predelay (Appendix B); live changes are checked against orc_ref_*. */
void orc_upols_partial(orc_upols *u, const float *in1, const float *in2, double *wetL, double *wetR) {
    up_grow(u);
    size_t t = u->t;
    const float *in[2] = {in1, in2};
    orc_cc_value *cc = u->cc;

    /* Q7: scalar coefficient of the (constant) selected IR, conv.cu:27, 345, 353 */
    for (int i = 0; i < 2; i++) {
        u->e[i] += ((double)cc[i].wet - u->e[i]) / (double)(cc[i].vsteps + 5);
        if (cc[i].vsteps > 0) cc[i].vsteps--;
    }
    /* wet gains attached to this input block: pan(panWet_i) * level_i * e_i */
    double *G = u->G + t * 4;
    for (int i = 0; i < 2; i++) {
        G[0 * 2 + i] = pan_l(cc[i].panWet) * cc[i].level * u->e[i];
        G[1 * 2 + i] = pan_r(cc[i].panWet) * cc[i].level * u->e[i];
    }

    /* forward transforms of the zero-padded blocks (overlap-add form: a block's
     * spectrum belongs to one input block only, so the per-block gains of Q7 /
     * pan / level attach to it exactly as in the reference, where the whole
     * contribution of input block t is scaled by the values current at t) */
    cplx buf[UP_K];
    double S[2] = {0, 0}, A[2] = {0, 0};
    for (int i = 0; i < 2; i++) {
        memset(buf, 0, sizeof(buf));
        for (size_t m = 0; m < UP_B; m++) {
            buf[m] = (cplx){in[i][m], 0};
            S[i] += in[i][m];
            A[i] += ((m & 1) ? -1.0 : 1.0) * in[i][m];
        }
        fft_c(buf, UP_K, -1);
        memcpy(u->X[i] + t * UP_BINS, buf, sizeof(cplx) * UP_BINS);
    }

    /* Q1/Q2 rank-1 terms of this input block (Appendix B) and their prefix sums */
    const up_ir *ir0 = &u->ir[cc[0].select], *ir1 = &u->ir[cc[1].select];
    const up_ir *irs[2] = {ir0, ir1};
    double Nr = (double)u->n_ref;
    double D[2], Q[2];
    /* reference DC term - true DC term (derivation in DESIGN.md §Q1) */
    D[0] = -(G[0 * 2 + 0] * S[1] * ir0->sums[1] + G[0 * 2 + 1] * S[1] * ir1->sums[0]) / Nr;
    D[1] = -(G[1 * 2 + 0] * S[0] * ir0->sums[1] + G[1 * 2 + 1] * S[1] * ir1->sums[1]) / Nr;
    for (int c = 0; c < 2; c++) Q[c] = -(G[c * 2 + 0] * A[0] * ir0->sums[2 + c] + G[c * 2 + 1] * A[1] * ir1->sums[2 + c]) / Nr;
    for (int c = 0; c < 2; c++) {
        u->CD[t * 2 + c] = prefix_at(u->CD, (long)t - 1, c) + (u->compat ? D[c] : 0.0);
        u->CQ[t * 2 + c] = prefix_at(u->CQ, (long)t - 1, c) + (u->compat ? Q[c] : 0.0);
    }

    /* partition x bin MAC over the frequency-domain delay line, then inverse */
    for (int c = 0; c < 2; c++) {
        cplx Y[UP_K];
        memset(Y, 0, sizeof(Y));
        for (int i = 0; i < 2; i++) {
            const up_ir *ir = irs[i];
            size_t p_lo = u->pb, p_hi = u->pe ? (u->pe < ir->P ? u->pe : ir->P) : ir->P;
            for (size_t p = p_lo; p < p_hi && p <= t; p++) {
                double g = u->G[(t - p) * 4 + c * 2 + i];
                const cplx *Hp = ir->H[c] + p * UP_BINS;
                const cplx *Xp = u->X[i] + (t - p) * UP_BINS;
                for (size_t k = 0; k < UP_BINS; k++) Y[k] = c_add(Y[k], c_scale(c_mul(Hp[k], Xp[k]), g));
            }
        }
        for (size_t k = 1; k < UP_B; k++) Y[UP_K - k] = c_conj(Y[k]);
        fft_c(Y, UP_K, +1);
        /* overlap-add: first half completes output block t, second half opens t+1 */
        for (size_t m = 0; m < UP_B; m++) {
            u->wet[c][t * UP_B + m] += Y[m].x / (double)UP_K;
            u->wet[c][(t + 1) * UP_B + m] = Y[UP_B + m].x / (double)UP_K;
        }
    }

    for (size_t m = 0; m < UP_B; m++) {
        wetL[m] = u->wet[0][t * UP_B + m];
        wetR[m] = u->wet[1][t * UP_B + m];
    }
    u->pending = 1;
}

/* second half: the wet block summed over all shards -> predelay, Q1/Q2 window
 * sums, clamp, dry mix (conv.cu:411-427) */
void orc_upols_finish(orc_upols *u, const float *in1, const float *in2, const double *wsumL, const double *wsumR,
                      double *outL, double *outR) {
    if (!u->pending) return;
    size_t t = u->t;
    orc_cc_value *cc = u->cc;
    for (size_t m = 0; m < UP_B; m++) {
        u->wsum[0][t * UP_B + m] = wsumL[m];
        u->wsum[1][t * UP_B + m] = wsumR[m];
    }
    /* predelay (half 0's, conv.cu:412,415), corrections, clamp, dry */
    long pd = (long)cc[0].predelay, N = (long)u->n_ref;
    double dgain[2][2];
    for (int i = 0; i < 2; i++) {
        dgain[0][i] = (double)cc[i].dry * pan_l(cc[i].panDry) * cc[i].level;
        dgain[1][i] = (double)cc[i].dry * pan_r(cc[i].panDry) * cc[i].level;
    }
    double *out[2] = {outL, outR};
    for (int c = 0; c < 2; c++) {
        for (size_t m = 0; m < UP_B; m++) {
            long tau = (long)(t * UP_B + m);
            double w = tau - pd >= 0 ? u->wsum[c][tau - pd] : 0.0;
            /* blocks t' with pd <= tau - t'B < N  (shift by pd, cut at N: Q8) */
            long thi = floordiv(tau - pd, UP_B), tlo = floordiv(tau - N, UP_B);
            if (thi > (long)t) thi = (long)t;
            double cd = prefix_at(u->CD, thi, c) - prefix_at(u->CD, tlo, c);
            double cq = prefix_at(u->CQ, thi, c) - prefix_at(u->CQ, tlo, c);
            double sign = ((tau - pd) & 1) ? -1.0 : 1.0;
            double v = clampd(w + cd + sign * cq, -1, 1);
            out[c][m] = v + in1[m] * dgain[c][0] + in2[m] * dgain[c][1];
        }
    }
    u->t++;
    u->pending = 0;
}

void orc_upols_process(orc_upols *u, const float *in1, const float *in2, double *outL, double *outR,
                       size_t nframes) {
    /* SURVEY Appendix B; nframes must be 256 */
    if (nframes != UP_B) return;
    double wl[UP_B], wr[UP_B];
    orc_upols_partial(u, in1, in2, wl, wr);
    orc_upols_finish(u, in1, in2, wl, wr, outL, outR);
}

/* Range evaluator: output blocks [b0, b0 + n) of the SAME partitioned form, for a stream that started cold at
 * block 0 (all state zero, Q6) and whose parameters u->cc stayed constant throughout.  in1 / in2 hold blocks
 * [0, b0 + n).  What the blocks before b0 leave behind - their cross-fade coefficients (Q7, conv.cu:27,345,353),
 * their Q1/Q2 terms (conv.cu:47-73) and the spectra the window of the first evaluated block still reaches -
 * is rebuilt without running their partition sums: O((n + pd/256) * P) instead of O(b0 * P).  Same arithmetic as
 * orc_upols_partial / orc_upols_finish above (conv.cu:392-401 sum, :403-408 inverse, :411-427 post stage);
 * test_oracle.py checks it against them and against orc_ref_process.
 * Returns 0, or -1 for a bad argument / used engine, -2 where the Q8 tail drop would act (not modelled here:
 * longest IR + 255 + predelay > n_ref). */
static int upols_range(orc_upols *u, const float *in1, const float *in2, size_t b0, size_t n, double *outL, double *outR,
                       int settled) {
    if (!u || !in1 || !in2 || !n || u->t != 0) return -1;
    orc_cc_value cc[2] = {u->cc[0], u->cc[1]};
    const up_ir *irs[2] = {&u->ir[cc[0].select], &u->ir[cc[1].select]};
    if (!irs[0]->H[0] || !irs[1]->H[0]) return -1;
    const float *in[2] = {in1, in2};
    const size_t total = b0 + n;
    const long pd = (long)cc[0].predelay, N = (long)u->n_ref;
    size_t tmax = irs[0]->taps > irs[1]->taps ? irs[0]->taps : irs[1]->taps;
    if (u->compat && tmax + 255 + (size_t)pd > u->n_ref) return -2;
    /* settled: in1 / in2 start somewhere in a stream whose cross-fade has converged (e_i = wet_i for every block
     * given and for every earlier block that still matters): block 0 of the buffers must then lie beyond the reach
     * of the first evaluated block - one reference length + the predelay (Q1/Q2 windows, conv.cu:89-100) and the
     * longest IR + the predelay (partition sums) */
    if (settled && (b0 * UP_B < (size_t)N + (size_t)pd + 2 * UP_B || b0 * UP_B < tmax + (size_t)pd + 3 * UP_B)) return -3;

    /* per input block: wet gains (Q7 recurrence), Q1/Q2 terms and their prefix sums - cheap, from block 0 */
    double *G = (double *)malloc(sizeof(double) * total * 4);
    double *CD = (double *)malloc(sizeof(double) * total * 2), *CQ = (double *)malloc(sizeof(double) * total * 2);
    double e[2] = {settled ? (double)cc[0].wet : 0.0, settled ? (double)cc[1].wet : 0.0};
    if (settled) cc[0].vsteps = cc[1].vsteps = 0;
    const double Nr = (double)u->n_ref;
    for (size_t t = 0; t < total; t++) {
        double *g = G + t * 4;
        for (int i = 0; i < 2; i++) {
            e[i] += ((double)cc[i].wet - e[i]) / (double)(cc[i].vsteps + 5);
            if (cc[i].vsteps > 0) cc[i].vsteps--;
            g[0 * 2 + i] = pan_l(cc[i].panWet) * cc[i].level * e[i];
            g[1 * 2 + i] = pan_r(cc[i].panWet) * cc[i].level * e[i];
        }
        double S[2] = {0, 0}, A[2] = {0, 0};
        for (int i = 0; i < 2; i++)
            for (size_t m = 0; m < UP_B; m++) {
                const double x = in[i][t * UP_B + m];
                S[i] += x;
                A[i] += ((m & 1) ? -1.0 : 1.0) * x;
            }
        double D[2], Q[2];
        D[0] = -(g[0 * 2 + 0] * S[1] * irs[0]->sums[1] + g[0 * 2 + 1] * S[1] * irs[1]->sums[0]) / Nr;
        D[1] = -(g[1 * 2 + 0] * S[0] * irs[0]->sums[1] + g[1 * 2 + 1] * S[1] * irs[1]->sums[1]) / Nr;
        for (int c = 0; c < 2; c++) Q[c] = -(g[c * 2 + 0] * A[0] * irs[0]->sums[2 + c] + g[c * 2 + 1] * A[1] * irs[1]->sums[2 + c]) / Nr;
        for (int c = 0; c < 2; c++) {
            CD[t * 2 + c] = prefix_at(CD, (long)t - 1, c) + (u->compat ? D[c] : 0.0);
            CQ[t * 2 + c] = prefix_at(CQ, (long)t - 1, c) + (u->compat ? Q[c] : 0.0);
        }
    }

    /* first block whose segment is needed: the wet sample b0 * 256 - pd lies in block fb, whose first half also
     * takes the second half of block fb - 1 */
    long fb = floordiv((long)(b0 * UP_B) - pd, UP_B) - 1;
    const size_t m0 = fb > 0 ? (size_t)fb : 0;
    size_t Pmax = 1;
    for (int i = 0; i < 2; i++) {
        size_t ph = u->pe ? (u->pe < irs[i]->P ? u->pe : irs[i]->P) : irs[i]->P;
        if (ph > Pmax) Pmax = ph;
    }
    const size_t x0 = m0 + 1 > Pmax ? m0 + 1 - Pmax : 0; /* first block whose spectrum some evaluated window reaches */
    const size_t nx = total - x0, nm = total - m0;
    cplx *X[2];
    for (int i = 0; i < 2; i++) X[i] = (cplx *)malloc(sizeof(cplx) * nx * UP_BINS);
    double *seg = (double *)malloc(sizeof(double) * nm * 2 * UP_K); /* [t - m0][c][512] */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long q = 0; q < (long)(nx * 2); q++) {
        const size_t t = x0 + (size_t)q / 2;
        const int i = (int)(q & 1);
        cplx buf[UP_K];
        memset(buf, 0, sizeof(buf));
        for (size_t m = 0; m < UP_B; m++) buf[m] = (cplx){in[i][t * UP_B + m], 0};
        fft_c(buf, UP_K, -1);
        memcpy(X[i] + (t - x0) * UP_BINS, buf, sizeof(cplx) * UP_BINS);
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (long q = 0; q < (long)(nm * 2); q++) {
        const size_t t = m0 + (size_t)q / 2;
        const int c = (int)(q & 1);
        cplx Y[UP_K];
        memset(Y, 0, sizeof(Y));
        for (int i = 0; i < 2; i++) {
            const up_ir *ir = irs[i];
            size_t p_lo = u->pb, p_hi = u->pe ? (u->pe < ir->P ? u->pe : ir->P) : ir->P;
            for (size_t p = p_lo; p < p_hi && p <= t; p++) {
                const double g = G[(t - p) * 4 + c * 2 + i];
                const cplx *Hp = ir->H[c] + p * UP_BINS;
                const cplx *Xp = X[i] + (t - p - x0) * UP_BINS;
                for (size_t k = 0; k < UP_BINS; k++) Y[k] = c_add(Y[k], c_scale(c_mul(Hp[k], Xp[k]), g));
            }
        }
        for (size_t k = 1; k < UP_B; k++) Y[UP_K - k] = c_conj(Y[k]);
        fft_c(Y, UP_K, +1);
        double *s = seg + ((t - m0) * 2 + c) * UP_K;
        for (size_t m = 0; m < UP_K; m++) s[m] = Y[m].x / (double)UP_K;
    }
    /* overlap-add, predelay, Q1/Q2 window sums, clamp, dry mix: as orc_upols_finish */
    double dgain[2][2];
    for (int i = 0; i < 2; i++) {
        dgain[0][i] = (double)cc[i].dry * pan_l(cc[i].panDry) * cc[i].level;
        dgain[1][i] = (double)cc[i].dry * pan_r(cc[i].panDry) * cc[i].level;
    }
    double *out[2] = {outL, outR};
    for (int c = 0; c < 2; c++)
        for (size_t t = b0; t < total; t++)
            for (size_t m = 0; m < UP_B; m++) {
                const long tau = (long)(t * UP_B + m), ws = tau - pd;
                double w = 0.0;
                if (ws >= 0) {
                    const size_t wb = (size_t)ws / UP_B, wm = (size_t)ws % UP_B;
                    /* wb >= m0 + 1 unless the stream itself starts there (m0 == 0) */
                    w = seg[((wb - m0) * 2 + c) * UP_K + wm];
                    if (wb > m0) w += seg[((wb - 1 - m0) * 2 + c) * UP_K + UP_B + wm];
                }
                long thi = floordiv(tau - pd, UP_B), tlo = floordiv(tau - N, UP_B);
                if (thi > (long)t) thi = (long)t;
                const double cd = prefix_at(CD, thi, c) - prefix_at(CD, tlo, c);
                const double cq = prefix_at(CQ, thi, c) - prefix_at(CQ, tlo, c);
                const double sign = ((tau - pd) & 1) ? -1.0 : 1.0;
                const double v = clampd(w + cd + sign * cq, -1, 1);
                out[c][(t - b0) * UP_B + m] = v + in1[t * UP_B + m] * dgain[c][0] + in2[t * UP_B + m] * dgain[c][1];
            }
    free(seg);
    free(X[0]);
    free(X[1]);
    free(G);
    free(CD);
    free(CQ);
    return 0;
}

int orc_upols_range(orc_upols *u, const float *in1, const float *in2, size_t b0, size_t n, double *outL, double *outR) {
    return upols_range(u, in1, in2, b0, n, outL, outR, 0);
}

/* The same for an excerpt of a long-running stream in steady state: in1 / in2 begin at some block of the stream
 * at which (and for one reference length + predelay before which) the cross-fade coefficients had reached wet_i;
 * b0 counts from the start of the buffers and must lie beyond the reach stated above (-3 otherwise). */
int orc_upols_range_settled(orc_upols *u, const float *in1, const float *in2, size_t b0, size_t n, double *outL,
                            double *outR) {
    return upols_range(u, in1, in2, b0, n, outL, outR, 1);
}

/* =============================================================== cpu32 ==== */
/* float32 uniform-partition overlap-save, the timed CPU baseline ("port").
 * SoA spectra, bin-major, partitions stored reversed next to a doubled ring so
 * the partition sum is a contiguous dot product. */
struct orc_cpu32 {
    size_t P;
    float *Hre[4], *Him[4]; /* path = c*2+i : [k*P + q], q = P-1-p */
    float *Xre[2], *Xim[2]; /* [k*2P + slot] doubled ring */
    float prev[2][UP_B];
    size_t t;
    float twr[UP_K / 2], twi[UP_K / 2];
};

static void fft512_f32(float *re, float *im, const float *twr, const float *twi, int sign) {
    const size_t n = UP_K;
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            float t = re[i];
            re[i] = re[j];
            re[j] = t;
            t = im[i];
            im[i] = im[j];
            im[j] = t;
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1, step = n / len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < half; k++) {
                float cr = twr[k * step], ci = sign < 0 ? twi[k * step] : -twi[k * step];
                size_t a = i + k, b = a + half;
                float tr = re[b] * cr - im[b] * ci, ti = re[b] * ci + im[b] * cr;
                re[b] = re[a] - tr;
                im[b] = im[a] - ti;
                re[a] += tr;
                im[a] += ti;
            }
    }
}

orc_cpu32 *orc_cpu32_create(const float *lr0, const float *lr1, size_t frames) {
    orc_cpu32 *c = (orc_cpu32 *)calloc(1, sizeof(orc_cpu32));
    size_t P = (frames + UP_B - 1) / UP_B;
    if (!P) P = 1;
    c->P = P;
    for (size_t k = 0; k < UP_K / 2; k++) {
        double a = -2.0 * M_PI * (double)k / (double)UP_K;
        c->twr[k] = (float)cos(a);
        c->twi[k] = (float)sin(a);
    }
    for (int j = 0; j < 4; j++) {
        c->Hre[j] = (float *)calloc(UP_BINS * P, sizeof(float));
        c->Him[j] = (float *)calloc(UP_BINS * P, sizeof(float));
    }
    for (int i = 0; i < 2; i++) {
        c->Xre[i] = (float *)calloc(UP_BINS * 2 * P, sizeof(float));
        c->Xim[i] = (float *)calloc(UP_BINS * 2 * P, sizeof(float));
    }
    const float *lrs[2] = {lr0, lr1};
    float re[UP_K], im[UP_K];
    for (int i = 0; i < 2; i++)
        for (int ch = 0; ch < 2; ch++)
            for (size_t p = 0; p < P; p++) {
                memset(re, 0, sizeof(re));
                memset(im, 0, sizeof(im));
                for (size_t m = 0; m < UP_B; m++) {
                    size_t s = p * UP_B + m;
                    if (s < frames) re[m] = lrs[i][2 * s + ch];
                }
                fft512_f32(re, im, c->twr, c->twi, -1);
                int path = ch * 2 + i;
                for (size_t k = 0; k < UP_BINS; k++) {
                    c->Hre[path][k * P + (P - 1 - p)] = re[k];
                    c->Him[path][k * P + (P - 1 - p)] = im[k];
                }
            }
    return c;
}

void orc_cpu32_destroy(orc_cpu32 *c) {
    if (!c) return;
    for (int j = 0; j < 4; j++) {
        free(c->Hre[j]);
        free(c->Him[j]);
    }
    for (int i = 0; i < 2; i++) {
        free(c->Xre[i]);
        free(c->Xim[i]);
    }
    free(c);
}

size_t orc_cpu32_partitions(const orc_cpu32 *c) { return c->P; }

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_cpu32_process(orc_cpu32 *c, const float *in1, const float *in2, float *outL, float *outR,
                       size_t nblocks, const float g[4], const float d[4], int nthreads) {
    const size_t P = c->P;
    (void)nthreads;
    for (size_t b = 0; b < nblocks; b++) {
        const float *x1 = in1 + b * UP_B, *x2 = in2 + b * UP_B;
        /* packed forward transform z = w1 + j w2 of the sliding windows */
        float re[UP_K], im[UP_K];
        for (size_t m = 0; m < UP_B; m++) {
            re[m] = c->prev[0][m];
            im[m] = c->prev[1][m];
            re[UP_B + m] = x1[m];
            im[UP_B + m] = x2[m];
        }
        memcpy(c->prev[0], x1, sizeof(float) * UP_B);
        memcpy(c->prev[1], x2, sizeof(float) * UP_B);
        fft512_f32(re, im, c->twr, c->twi, -1);
        size_t slot = c->t % P;
        for (size_t k = 0; k < UP_BINS; k++) {
            size_t kk = (UP_K - k) % UP_K;
            float ar = re[k], ai = im[k], br = re[kk], bi = -im[kk];
            float x1r = 0.5f * (ar + br), x1i = 0.5f * (ai + bi);
            float dr = 0.5f * (ar - br), di = 0.5f * (ai - bi);
            float x2r = di, x2i = -dr; /* -j * d */
            float *X;
            X = c->Xre[0] + k * 2 * P;
            X[slot] = X[slot + P] = x1r;
            X = c->Xim[0] + k * 2 * P;
            X[slot] = X[slot + P] = x1i;
            X = c->Xre[1] + k * 2 * P;
            X[slot] = X[slot + P] = x2r;
            X = c->Xim[1] + k * 2 * P;
            X[slot] = X[slot + P] = x2i;
        }
        /* partition x bin MAC: window base so that entry q holds X[t-(P-1-q)] */
        size_t base = (slot + 1) % P;
        float Yre[2][UP_K], Yim[2][UP_K];
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : omp_get_max_threads())
#endif
        for (long k = 0; k < (long)UP_BINS; k++) {
            for (int ch = 0; ch < 2; ch++) {
                float sr = 0, si = 0;
                for (int i = 0; i < 2; i++) {
                    int path = ch * 2 + i;
                    const float *hr = c->Hre[path] + (size_t)k * P, *hi = c->Him[path] + (size_t)k * P;
                    const float *xr = c->Xre[i] + (size_t)k * 2 * P + base, *xi = c->Xim[i] + (size_t)k * 2 * P + base;
                    float ar = 0, ai = 0;
#ifdef _OPENMP
#pragma omp simd reduction(+ : ar, ai)
#endif
                    for (size_t q = 0; q < P; q++) {
                        ar += hr[q] * xr[q] - hi[q] * xi[q];
                        ai += hr[q] * xi[q] + hi[q] * xr[q];
                    }
                    sr += g[path] * ar;
                    si += g[path] * ai;
                }
                Yre[ch][k] = sr;
                Yim[ch][k] = si;
            }
        }
        /* packed inverse: W = Y_L + j Y_R with Hermitian extension */
        for (size_t k = 0; k < UP_BINS; k++) {
            re[k] = Yre[0][k] - Yim[1][k];
            im[k] = Yim[0][k] + Yre[1][k];
            if (k && k < UP_B) {
                re[UP_K - k] = Yre[0][k] + Yim[1][k];
                im[UP_K - k] = -Yim[0][k] + Yre[1][k];
            }
        }
        fft512_f32(re, im, c->twr, c->twi, +1);
        for (size_t m = 0; m < UP_B; m++) {
            float wl = re[UP_B + m] * (1.0f / UP_K), wr = im[UP_B + m] * (1.0f / UP_K);
            outL[b * UP_B + m] = wl + x1[m] * d[0] + x2[m] * d[1];
            outR[b * UP_B + m] = wr + x1[m] * d[2] + x2[m] * d[3];
        }
        c->t++;
    }
}
