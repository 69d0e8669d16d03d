/*
 * oracle.h — CPU restatement of the limitz/cuda-audio convolution hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it, and only as the checker / the timed CPU baseline.  The product
 * (cuda_audio_amd + libmcconv.so) never links or imports this code.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path, it is CUDA+cuFFT and cannot be built or
 * run here, so this restatement is pinned only by (1) exact time-domain
 * convolution, (2) an independent numpy/pocketfft restatement
 * (oracle/refcompat_np.py) and (3) the A==B identity of SURVEY Appendix A/B.
 *
 * Every function cites the reference lines it follows (paths under
 * /root/reference).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_PREDELAY 8192 /* src/conv.h:26-28 CONV_MAX_PREDELAY */
#define ORC_BLOCK 256         /* partition size of the partitioned restatement */

/* mirrors Convolution::CC::value, src/conv.h:38-49 (same defaults) */
typedef struct {
    uint64_t select;   /* [0, nIR)   */
    uint64_t predelay; /* [0, 8192]  */
    uint64_t speed;    /* [0, 1024]  */
    uint64_t vsteps;
    float dry, wet, panDry, panWet, level;
} orc_cc_value;

void orc_cc_defaults(orc_cc_value *v);

/* ---- plain DFT (what cuFFT's C2C computes, unnormalised; sign=-1 forward,
 * +1 inverse; src/conv.cu:243,367,405,407).  n must be a power of two. */
void orc_fft(double *re, double *im, size_t n, int sign);

/* ---- config 1: direct O(N*M) time-domain convolution, y has nx+nh-1 taps */
void orc_direct_conv(const float *x, size_t nx, const float *h, size_t nh, double *y);

/* ---- WAV sample scaling of src/wav.cu:17-57 (Q5): s16 / 65536, s24 / 2^24 */
void orc_wav_decode_s16(const int16_t *lr, size_t frames, float *out_lr);
void orc_wav_decode_s24(const uint8_t *bytes, size_t frames, float *out_lr);

/* ---- "refcompat": the reference's single-FFT algorithm, float64, state
 * machine with the reference's buffers (src/conv.cu:142-195, 207-253, 287-466).
 * three_mult != 0 uses the literal (wrong) 3-multiply product of
 * conv.cu:117-120; 0 uses the true complex product (identical real output). */
typedef struct orc_ref orc_ref;
orc_ref *orc_ref_create(size_t fftSize, int three_mult);
void orc_ref_destroy(orc_ref *r);
/* Convolution::prepare, conv.cu:207-253; lr = interleaved L,R floats as
 * WavFile would hold them (already scaled); nframes default 1024. */
int orc_ref_prepare(orc_ref *r, size_t idx, const float *lr, size_t frames, size_t nframes);
orc_cc_value *orc_ref_cc(orc_ref *r, int half); /* mutable, like public cc[] */
size_t orc_ref_num_irs(const orc_ref *r);
/* handleCC, conv.cu:255-276 (message/CC-number matching done by caller ids) */
void orc_handle_cc(orc_cc_value *v, const uint8_t ccmap[8], uint8_t m2, int val, size_t nb);
/* Convolution::onProcess, conv.cu:287-466.  outputs are doubles. */
void orc_ref_process(orc_ref *r, const float *in1, const float *in2, double *outL, double *outR,
                     size_t nframes);
/* IR sums used by the Q1/Q2 closed forms: sigma = sum h, alpha = sum h(-1)^m
 * over the truncated IR; out[0..3] = sigma_L, sigma_R, alpha_L, alpha_R */
int orc_ref_ir_sums(const orc_ref *r, size_t idx, double out[4]);

/* ---- "upols": uniform-partitioned overlap-save form,
 * B = 256, K = 512, float64, with the Q1/Q2 rank-1 corrections, Q7 ramp,
 * predelay, clamp and dry mix.  Must equal refcompat to rounding. */
typedef struct orc_upols orc_upols;
orc_upols *orc_upols_create(size_t n_ref, int compat /*apply Q1/Q2*/);
void orc_upols_destroy(orc_upols *u);
int orc_upols_prepare(orc_upols *u, size_t idx, const float *lr, size_t frames, size_t nframes);
orc_cc_value *orc_upols_cc(orc_upols *u, int half);
void orc_upols_process(orc_upols *u, const float *in1, const float *in2, double *outL, double *outR,
                       size_t nframes);
/* sharded form: partition range [pb, pe) (pe = 0: all); partial gives
 * this shard's 256-sample wet block (pre-predelay), finish takes the sum over shards */
void orc_upols_set_shard(orc_upols *u, size_t pb, size_t pe);
void orc_upols_partial(orc_upols *u, const float *in1, const float *in2, double *wetL, double *wetR);
void orc_upols_finish(orc_upols *u, const float *in1, const float *in2, const double *wsumL, const double *wsumR,
                      double *outL, double *outR);
/* range evaluator: output blocks [b0, b0 + n) of a stream that started cold at block 0 with the constant
 * parameters now in orc_upols_cc(); in1 / in2 hold blocks [0, b0 + n); the blocks before b0 contribute their
 * gains, Q1/Q2 terms and spectra but their own partition sums are not run (a fresh engine only; honours
 * orc_upols_set_shard).  0 = ok, -1 bad argument, -2 the Q8 tail drop would act (not modelled). */
int orc_upols_range(orc_upols *u, const float *in1, const float *in2, size_t b0, size_t n, double *outL, double *outR);
/* ... for an excerpt of a long stream in steady state (cross-fade converged: e_i = wet_i throughout): the buffers
 * begin anywhere in the stream, b0 counts from their start and must exceed n_ref + predelay and the longest IR +
 * predelay by two blocks (-3 otherwise) */
int orc_upols_range_settled(orc_upols *u, const float *in1, const float *in2, size_t b0, size_t n, double *outL,
                            double *outR);

/* ---- CPU baseline ("port"): float32 uniform-partition overlap-save, OpenMP
 * over bins, steady-state hot path only (fwd FFT, partition x bin MAC for the
 * 2x2 path matrix, inverse FFT, dry mix).  Timed by bench.py. */
typedef struct orc_cpu32 orc_cpu32;
orc_cpu32 *orc_cpu32_create(const float *lr0, const float *lr1, size_t frames);
void orc_cpu32_destroy(orc_cpu32 *c);
size_t orc_cpu32_partitions(const orc_cpu32 *c);
/* gains: wet gain per path g[c][i] (c=L,R; i=in1,in2), dry gain d[c][i] */
void orc_cpu32_process(orc_cpu32 *c, const float *in1, const float *in2, float *outL, float *outR,
                       size_t nblocks, const float g[4], const float d[4], int nthreads);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
