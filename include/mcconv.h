/*
 * mcconv.h — C ABI of the MI355X partitioned-convolution reverb engine
 * (libmcconv.so, hand-written HIP for gfx950).
 *
 * This is the drop-in boundary for the reference's hot path: the entry points
 * are what a binding for limitz/cuda-audio's `Convolution` class needs, one
 * per reference interface (citations are file:line under the reference's src/):
 *
 *   mc_create            <- Convolution::Convolution(name, fftSize)   conv.h:52, conv.cu:142-195
 *   mc_destroy           <- (reference never frees; conv.h:53-54)
 *   mc_load_ir           <- Convolution::prepare(idx, wav, nframes)    conv.h:63, conv.cu:207-253
 *   mc_set_params /
 *   mc_get_params        <- public Convolution::cc[2].value            conv.h:33-50 (written by main.cu:49-70)
 *   mc_handle_cc         <- Convolution::onMidiMessage / handleCC      conv.h:65, conv.cu:255-285
 *   mc_process           <- Convolution::onProcess(nframes)            conv.h:59, conv.cu:287-466
 *   mc_avg_runtime_ms    <- Convolution::avgRuntime()                  conv.h:61, conv.cu:454-462
 *   mc_process_batch*    <- the same per-block path run over T consecutive
 *                           blocks in one call (throughput mode; new)
 *   mc_partial_* / mc_finish_* <- the two halves of a batch around the
 *                           cross-GPU sum when IR partitions are sharded (new)
 *
 * Plain pointers and sizes only; no C++ or torch types.  Every function
 * returns 0 on success or a negative mc_status, never throws, and leaves a
 * message for mc_last_error() (thread-local).  The caller keeps ownership of
 * every pointer it passes; the engine owns all device memory.
 *
 * Threading (mirrors SURVEY.md §8b): mc_process* is single-caller per engine;
 * mc_set_params / mc_handle_cc may be called from another thread (values are
 * sampled once at the start of each process call); mc_load_ir must not run
 * concurrently with mc_process*.
 */
#ifndef MCCONV_H
#define MCCONV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC_ABI_VERSION 1
#define MC_BLOCK 256          /* frames per internal block (and the default JACK period) */
#define MC_MAX_PREDELAY 8192  /* conv.h:26-28 */
#define MC_MAX_SPEED 1024     /* conv.h:22-24 */

typedef enum {
    MC_OK = 0,
    MC_ERR_ARG = -1,      /* bad argument / unsupported size */
    MC_ERR_HIP = -2,      /* HIP runtime error (message has the call) */
    MC_ERR_STATE = -3,    /* call not valid in this state (e.g. no IR loaded) */
    MC_ERR_NOMEM = -4
} mc_status;

typedef struct mc_engine mc_engine;

typedef struct {
    uint32_t struct_size;   /* sizeof(mc_config), for ABI evolution */
    int32_t device;         /* HIP device ordinal; -1 = current device */
    uint64_t n_ref;         /* the reference's fftSize (conv.h:52): IR truncation n_ref-1024
                               (conv.cu:239), Q1/Q2 window and 1/n_ref factors */
    uint32_t max_batch;     /* largest nblocks accepted by the device-buffer batch calls (1..1048576; the engine's rings and
                               scratch are sized by it: about 33 KB of device memory per block) */
    uint32_t max_partitions;/* 0 = derive from n_ref: ceil((n_ref-1024)/256) */
    uint32_t compat;        /* 1 = bug-compatible with conv.cu (DC/Nyquist terms Q1/Q2);
                               0 = plain linear convolution */
    uint32_t part_begin;    /* IR-partition shard [part_begin, part_end) computed by this */
    uint32_t part_end;      /* engine; 0,0 = all partitions (single GPU) */
    uint32_t stream_threshold; /* batches shorter than this use the streaming MAC kernel
                               (0 = default) */
    uint32_t precision;     /* 0 = fp32 spectra; 1 = fp16 storage of IR spectra and delay line for the
                               partition sweep (fp32 products and sums; the streaming kernel: single periods and batches
                               below 12288 blocks).  Longer settled batches take the overlap-save form in either precision
                               (its spectra are built from the fp32 taps: fp32 accuracy there) */
    uint32_t period;        /* JACK period the host will call mc_process with: 0/256, 512 or 1024 frames.  The
                               reference's per-call semantics (cross-fade step, DC/Nyquist and tail-drop windows)
                               follow this size; internally a period is 1, 2 or 4 blocks of 256.  Batch calls
                               take multiples of period/256 blocks. */
    uint32_t pipeline;      /* 1 = pipelined batches: mc_process_batch_device / _slice_device return once the MAC of
                               the batch is queued; its inverse transforms and post stage run on a second stream
                               under the MAC of the next batch.  The outputs of a call are complete, in the order of
                               the engine's stream, only after mc_fence (or mc_sync); its inputs must stay valid
                               until then.  Other calls drain the pipeline first.  0 = every call is complete in
                               stream order (default). */
    uint32_t form;          /* 0 = uniform-partitioned engine (default).  1 = the path in the reference's own shape
                               (BASELINE config 2): one n_ref-point transform per call, its live IR spectra stepped bin
                               by bin (conv.cu:15-32) and an n_ref-long running accumulator clamped at +-1 every call
                               (conv.cu:89-100) - the same samples while nothing saturates, the reference's samples
                               beyond.  n_ref <= 1048576, whole IR on one engine, fp32; no slices / shards / pipeline.
                               The environment variable MCCONV_FORM=single|partitioned overrides the field. */
    uint32_t reserved;
} mc_config;

/* mirrors Convolution::CC::value (conv.h:38-49); same defaults via mc_default_params */
typedef struct {
    uint64_t select;    /* index of the loaded IR used by this half */
    uint64_t predelay;  /* samples, [0, 8192]; half 0's value is used for both channels (conv.cu:412,415) */
    uint64_t speed;     /* cross-fade length in blocks, [0, 1024] */
    uint64_t vsteps;    /* remaining cross-fade steps (set to speed by a select CC) */
    float dry, wet, panDry, panWet, level;
} mc_cc_value;

typedef struct {
    uint64_t launches;      /* MAC-kernel launches timed so far */
    uint64_t blocks;        /* blocks those launches processed */
    double total_ms;        /* sum of the durations of those launches: HIP events around the launch; for the sweep of a
                               single period (a few microseconds) the kernel's own time stamps, first workgroup start to
                               last workgroup end */
    double last_ms;
    uint32_t resident;      /* 1 = the last launch used the resident (batch) kernel */
    uint32_t partitions;    /* partitions swept per block by the last launch */
    uint32_t fast_levels;   /* how the last long batch summed its partitions: 0 = direct-form MAC, 1..3 = fast-FIR form
                               with that many nested levels ((3/4)^levels of the multiply-adds); second-level transform
                               along the block axis instead of the MAC: 254 = its fused 8192-point form (one launch,
                               k_g2_mac: long batches with one set of gains where the overlap-save form does not apply),
                               255 = its split 16384-point form (k_f2_fwd + k_f2_prod: per-slot gains, IRs over 5632
                               partitions); 253 = overlap-save segments of 512 x 8192 frames (k_os_cols, k_os_rows, k_os_out:
                               the default for whole or block-sliced batches of >= 12288 blocks with one set of gains
                               outside the Q8 regime; `partitions` then holds the blocks of history per segment) */
    uint32_t reserved;
} mc_kernel_stats;

uint32_t mc_abi_version(void);
const char *mc_last_error(void);
void mc_default_config(mc_config *cfg);
void mc_default_params(mc_cc_value *v);

int mc_create(const mc_config *cfg, mc_engine **out);
void mc_destroy(mc_engine *e);
int mc_reset(mc_engine *e); /* zero all signal state (delay line, tails, cross-fade) */
/* change the JACK period (256, 512 or 1024 frames; see mc_config.period); resets the signal state */
int mc_set_period(mc_engine *e, uint32_t nframes);

/* lr: interleaved L,R float frames as WavFile holds them (already scaled, wav.cu Q5);
 * nframes: the reference's third prepare() argument (1024). Host pointer. */
int mc_load_ir(mc_engine *e, uint64_t idx, const float *lr, uint64_t frames, uint64_t nframes);
int mc_num_irs(const mc_engine *e);
/* out[0..3] = sum h_L, sum h_R, sum h_L(-1)^m, sum h_R(-1)^m of the truncated IR; out[4] = taps, out[5] = partitions */
int mc_ir_info(const mc_engine *e, uint64_t idx, double out[6]);

int mc_set_params(mc_engine *e, int half, const mc_cc_value *v);
int mc_get_params(const mc_engine *e, int half, mc_cc_value *v);
/* handleCC (conv.cu:255-276): ccmap = controller numbers {select,predelay,dry,wet,speed,panDry,panWet,level} */
int mc_handle_cc(mc_engine *e, int half, const uint8_t ccmap[8], uint8_t controller, int value);

/* One JACK period: host buffers in, host buffers out, returns when the output
 * is in outL/outR (like onProcess, which blocks on the GPU; conv.cu:455). */
int mc_process(mc_engine *e, const float *in1, const float *in2, float *outL, float *outR, uint64_t nframes);
/* nblocks consecutive periods, host buffers of nblocks*256 floats: the reference's per-block copies between JACK's
 * host buffers and the device (conv.cu:321-328, 431-437) for a whole run of blocks.  Any nblocks >= 1 (a multiple of
 * period/256); long runs are cut into chunks inside (parameters are sampled per chunk).  Pageable buffers (what JACK
 * hands the reference) go through the engine's pinned staging buffer, one chunk at a time.  Buffers in pinned host
 * memory - mc_host_alloc, hipHostMalloc or hipHostRegister, all four of them - are read and written by the DMA engines
 * directly: copy-in, kernels and copy-out of consecutive chunks overlap on three streams (PCIe-bound).  Returns when
 * the output is complete in outL / outR. */
int mc_process_batch(mc_engine *e, const float *in1, const float *in2, float *outL, float *outR, uint64_t nblocks);
/* pinned host memory for mc_process_batch buffers (hipHostMalloc; no reference equivalent - JACK owns its buffers) */
void *mc_host_alloc(size_t bytes);
void mc_host_free(void *p);
/* the same with device-resident buffers (16-byte aligned, as hipMalloc and block-granular slices of it are);
 * asynchronous on the engine's stream */
int mc_process_batch_device(mc_engine *e, const float *d_in1, const float *d_in2, float *d_outL, float *d_outR,
                            uint64_t nblocks);
/* Block-sliced operation - scaling batch throughput over GPUs without a data-path collective.  The output blocks
 * of a batch are independent given the input, so G engines (one per GPU) are fed the SAME batch and each finishes
 * `count` output blocks starting at block `first` of it into d_outL/d_outR (count*256 floats each).  An engine
 * transforms only the input blocks its own windows can reach (its slice, n_ref + 8192 frames before it and the
 * same distance before the next call's slice); the partition sums and inverse transforms run over the slice plus
 * the <= 33 blocks before it that the overlap-add and the predelay reach back to (count + reach-back <=
 * max_batch).  Consequences: `first` must be the same in every call, and an engine that has been called with a
 * proper slice accepts only sliced calls, no predelay change and at most three IRs cross-fading per half until
 * mc_reset (MC_ERR_STATE otherwise).
 * first = 0, count = nblocks is mc_process_batch_device. */
int mc_process_batch_slice_device(mc_engine *e, const float *d_in1, const float *d_in2, float *d_outL, float *d_outR,
                                  uint64_t nblocks, uint64_t first, uint64_t count);
/* Sharded operation: d_partial receives this engine's share of the wet signal,
 * 2*nblocks*256 floats ([L | R], overlap-added and already shifted by the predelay);
 * after the caller has summed the partials of all shards (RCCL reduce / all-reduce),
 * mc_finish_batch_device applies Q1/Q2 terms, Q8, clamp and dry mix.  Every shard must
 * see the same input and parameters.  A predelay change, or a select that makes a fourth
 * IR cross-fade in one half, re-renders the engine's history and must not arrive while a
 * batch is between its partial and its finish (MC_ERR_STATE).
 * Up to two batches may be between their partial and their finish (finishes
 * retire batches in order), so the reduce of batch k can overlap the MAC of
 * batch k+1.  A rank that does not need the output (non-root of a reduce)
 * passes NULL for d_wet_sum, d_outL and d_outR: the batch is retired, nothing runs (such a shard keeps no
 * Q1/Q2 history, so an engine should either always or never finish with output). */
int mc_partial_batch_device(mc_engine *e, const float *d_in1, const float *d_in2, float *d_partial, uint64_t nblocks);
int mc_finish_batch_device(mc_engine *e, const float *d_in1, const float *d_in2, const float *d_wet_sum,
                           float *d_outL, float *d_outR, uint64_t nblocks);
/* The finish after a REDUCE-SCATTER instead of a reduce: every shard receives the sum of the partials for ITS run of
 * blocks [first, first + count) only - d_wet_sum_slice = [L | R], count * 256 floats each (e.g. one reduce-scatter per
 * channel over the [nblocks * 256] channel halves of the partials) - and finishes those blocks into d_outL / d_outR
 * (count * 256 floats each).  d_in1 / d_in2 are the whole batch's inputs, as passed to mc_partial_batch_device.  Every
 * shard then keeps the Q1/Q2 history (it transforms the whole input anyway), no rank is a root, and each link carries
 * 1/N of what a reduce to one root funnels into that root.  first and count are multiples of period/256.  Retires the
 * batch like mc_finish_batch_device; an engine should use one of the two finishes throughout.  No reference
 * equivalent (the reference runs on one device, gpu.cu:38-90). */
int mc_finish_batch_slice_device(mc_engine *e, const float *d_in1, const float *d_in2, const float *d_wet_sum_slice,
                                 float *d_outL, float *d_outR, uint64_t nblocks, uint64_t first, uint64_t count);

int mc_sync(mc_engine *e);
/* pipelined engines: the engine's stream waits for everything issued so far (no host synchronisation) */
int mc_fence(mc_engine *e);
/* ... for everything except the most recently issued batch (so that the consumer of batch k - 1 can be queued
 * behind batch k without stalling batch k + 1 on the post stage of k) */
int mc_fence_older(mc_engine *e);
/* hip_stream is a hipStream_t.  NULL selects the engine's own stream, which is NON-BLOCKING: it is not ordered
 * with HIP's default stream.  A caller whose buffers are produced or consumed on the default stream (PyTorch's
 * current stream unless one is set) must pass that stream explicitly - MC_STREAM_DEFAULT, HIP's hipStreamLegacy
 * handle - or order the two with mc_sync / events. */
#define MC_STREAM_DEFAULT ((void *)1)
int mc_set_stream(mc_engine *e, void *hip_stream);
void *mc_get_stream(mc_engine *e);
double mc_avg_runtime_ms(const mc_engine *e);      /* mean ms per mc_process call after 10 warm-ups */
int mc_enable_kernel_timing(mc_engine *e, int on); /* HIP events around the MAC kernel */
int mc_get_kernel_stats(mc_engine *e, mc_kernel_stats *out, int reset);
uint64_t mc_algorithmic_bytes_per_block(const mc_engine *e); /* (4 paths + 2 inputs) * P * 2048 */
uint64_t mc_blocks_processed(const mc_engine *e);
/* Batch length (blocks, <= at_most and <= max_batch) that suits the loaded IRs best.  Batches of at least 12288 blocks and
 * one segment run as overlap-save segments of 16384 - P16 blocks (P16 = partitions of the longest loaded IR rounded up
 * to 16): whole segments waste nothing.  Below that the sum over partitions is a second-level transform in chunks of
 * 8192 - P16 + 1 blocks (16384 - P16 + 1 for IRs over 2560 partitions): whole chunks minus one block (the reach-back
 * of a block slice), a multiple of 8.  at_most shorter than a chunk is returned as it is (rounded down to 8).
 * No reference equivalent (batch calls are new). */
uint64_t mc_preferred_batch(const mc_engine *e, uint64_t at_most);

/* Diagnostics (tests only): copy `bytes` from an engine-owned device buffer to host.
 * which: 0 = IR spectra of IR `idx` (float4 [256][pstride]), 1 = delay line
 * (float4 [256][ring]), 2 = MAC output (float4 [256][max_batch]), 3 = segments,
 * 4 = wet ring, 5 = Q1/Q2 prefix ring (double [rc][4]); host-side words, no stream access: 6 = JACK-path counters
 * {parked periods used, gave up on their own, told to give up} (3 x uint64), 7 / 8 = generation of the parameter
 * pair the last process call sampled / that was published last (uint64), 9 = batches in the Q8 regime by the form
 * their cut terms took {k_drop_fft, forward transforms, time-domain tiles} and JACK periods whose cut terms came with the
 * launch before theirs (4 x uint64), 10 = batch launches by the form
 * of their partition sums {fused, split second-level transform, resident MAC} (3 x uint64), 11 = overlap-save form {batches that
 * took it, builds of its spectra} (2 x uint64), 12 / 13 / 14 = its row buffer and spectra (float4), 15 = 1 when the library is the lab
 * build (-DMCCONV_LAB: measurement switches and alternative kernels), 16 = 256-frame JACK tails by the form partition 0 took
 * {frequency domain, time domain}, counted by the kernel (2 x uint32; read behind the stream).  dims[0..3] receive
 * {pstride, ring, max_batch, wet ring length} when non-null. */
int mc_debug_read(mc_engine *e, int which, uint64_t idx, void *dst, uint64_t offset_bytes, uint64_t bytes,
                  uint64_t dims[4]);

#ifdef __cplusplus
}
#endif
#endif
