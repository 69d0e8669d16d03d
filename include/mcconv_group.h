/*
 * mcconv_group.h - C ABI of the multi-GPU driver (libmcconv_rccl.so; links libmcconv.so and librccl.so).
 *
 * BASELINE's north star: "IR partitions shard across the 8 GPUs of one node with an RCCL sum-reduce of per-GPU partial
 * output blocks over xGMI; the C++ host ... calling the HIP kernels through a thin C-ABI".  The reference itself runs on ONE
 * device (gpu.cu:38-90 picks it; main.cu:31-39 multiplies Convolution instances on it), so a group has no reference
 * interface to replace: it is what `Convolution` would call if it were given more than one device, and mirrors the
 * single-engine entry points of mcconv.h one for one:
 *
 *   mc_group_create        <- mc_create          one engine per listed device, each owning a run of the IR partitions
 *                                                 (multiples of 16, as cuda_audio_amd/sharded.py shard_bounds cuts them)
 *   mc_group_load_ir       <- mc_load_ir          (Convolution::prepare, conv.cu:207-253) on every engine
 *   mc_group_set_params    <- mc_set_params       (Convolution::cc[2].value, conv.h:33-50) on every engine
 *   mc_group_process_batch <- mc_process_batch    host buffers in and out; inside, one host thread per device:
 *                                                 H2D of the batch, mc_partial_batch_device, the exchange,
 *                                                 mc_finish_batch_slice_device, D2H of the rank's run of blocks
 *
 * The exchange is ncclReduceScatter(float, sum) per channel on the engines' streams when the batch splits into equal runs
 * of whole periods (every rank finishes nblocks / n blocks; each link carries 1/n of the partials), else ncclReduce to rank 0,
 * which finishes the whole batch.  Listing one device more than once (tests on a one-GPU box: "virtual ranks") replaces
 * RCCL - which refuses duplicate devices in one communicator - by a sum kernel on that device; mc_group_exchange says which.
 * Same error convention as mcconv.h (0 / negative mc_status, message through mc_last_error of the calling thread).
 */
#ifndef MCCONV_GROUP_H
#define MCCONV_GROUP_H
#include "mcconv.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mc_group mc_group;

/* cfg: as for mc_create (part_begin / part_end / device are overwritten per rank); devices: ndev HIP device ordinals.
 * cfg->reserved = 1 with ONE device: that rank still goes partial -> RCCL (a communicator of one) -> slice finish, so that the
 * collective's code path runs on a one-GPU box (tests); 0 = a group of one is the plain engine. */
int mc_group_create(const mc_config *cfg, const int32_t *devices, uint32_t ndev, mc_group **out);
void mc_group_destroy(mc_group *g);
uint32_t mc_group_size(const mc_group *g);
/* the engine of rank r (introspection, tests); owned by the group */
mc_engine *mc_group_engine(mc_group *g, uint32_t rank);
/* partitions [begin, end) of rank r */
int mc_group_shard(const mc_group *g, uint32_t rank, uint32_t *begin, uint32_t *end);
int mc_group_load_ir(mc_group *g, uint64_t idx, const float *lr, uint64_t frames, uint64_t nframes);
int mc_group_set_params(mc_group *g, int half, const mc_cc_value *v);
/* nblocks * 256 frames per channel, host pointers; returns when the output is complete */
int mc_group_process_batch(mc_group *g, const float *in1, const float *in2, float *outL, float *outR, uint64_t nblocks);
/* "rccl", "device-sum" (duplicate devices) or "none" (one rank) */
const char *mc_group_exchange(const mc_group *g);
/* message of the last failed mc_group_* call on this thread (also covers errors raised on the rank threads) */
const char *mc_group_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
