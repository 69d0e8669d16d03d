// mcconv.hip — engine state + the C ABI of include/mcconv.h.
//
// One mc_engine corresponds to one reference `Convolution` object
// (conv.h:30-86): it owns the IR bank (_irBuffers, conv.h:77), the
// frequency-domain delay line that replaces the reference's whole-IR spectra
// and N-long overlap-add accumulator (conv.h:72-75), the per-half parameters
// (cc[2].value, conv.h:33-50) and the running-mean timer (conv.h:61,79-80).
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/mcconv.h"
#include "params_handoff.h"
#include "kernels.hip.h"
#include "ossave.hip.h"
#include "singlefft.hip.h"

// Environment switches.  The default library reads ten - MCCONV_FORM, _OS, _FFT2, _FFT2_FUSED, _FFA_LEVELS (which form sums the
// partitions), _TD_FFT (Q8 cut terms in the time domain), _NO_PARK, _PARK_MS, _NO_SPIN, _BAR_IO (the JACK path's waiting and I/O) -
// all of which select paths a caller can also reach through mc_config or that the tests compare bit for bit.  Every other switch
// of rounds 1-3 selects a measured-and-lost alternative or a measurement and exists only under -DMCCONV_LAB (scripts/build_variant.sh).
#ifdef MCCONV_LAB
#define LAB_ENV(name) std::getenv(name)
#else
#define LAB_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(MC_ERR_HIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr int kMaxIrs = 256;
constexpr int kMixIrs = 4;  // merged spectra of deselected IRs: [half][buffer], table entries kMaxIrs ..
constexpr int kStageBufs = 4;
constexpr int kEvPool = 1024;
constexpr int kPipe = 2;
constexpr int kStampSlots = 64;  // timed launches between two drains whose kernels leave their own time stamps

inline uint32_t next_pow2(uint64_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline double pan_l(double p) { return p >= 0 ? 1 - p : 1; }  // conv.cu:386
inline double pan_r(double p) { return p <= 0 ? 1 + p : 1; }  // conv.cu:387

struct IrEntry {
    float4* d_H = nullptr;
    float4* d_Hp[3] = {nullptr, nullptr, nullptr};  // fast-FIR components of the partition sequence: 3 / 9 / 27 arrays,
    bool hp_valid[3] = {false, false, false};       // built on the first batch that selects that level (ensure_hp)
    float2* d_H2 = nullptr;  // second-level spectra [2 ch][257 rows][F2_N], built on first use (k_fft2_ir)
    bool h2_valid = false;
    float2* d_G2 = nullptr;  // second-level spectra for the fused 8192-point form [2 ch][257 rows][G2_N]
    bool g2_valid = false;
    float2* d_h = nullptr;  // time-domain taps {L, R} (Q8 pass)
    float4* d_Htail = nullptr;  // Q8 pass, frequency-domain form: the last partitions' spectra partition-major [P - tail_p0][256] (ensure_htail)
    int tail_p0 = 0;
    bool tail_valid = false;
    float2* d_S = nullptr;   // single-transform form (mc_config.form = 1): [H_L | H_R], n_ref / 2 bins each
    uint2* d_H16 = nullptr;  // fp16 copy of the spectra, scaled by scale16 (precision = fp16)
    float scale16 = 1.f;
    uint64_t taps = 0;
    int P = 0;
    double sums[4] = {0, 0, 0, 0};
};

}  // namespace


struct BatchCtx {
    int T = 0;
    uint64_t t0 = 0;
    int pstride = 1;
    uint64_t predelay = 0;
    VoiceSums vs;
    int vir[2][MC_MAXV];  // IR index per half and voice (-1 = none)
    int slot = 0;
    int first = 0, count = 0;  // output blocks this engine finishes (the whole batch unless block-sliced)
    int need_a0 = 0, need_a1 = 0, need_b0 = 0;  // blocks the front half transformed: [a0, a1) and [b0, T)
    uint64_t win0 = 0;         // first absolute sample whose segments the front half computed
    bool wet_ready = false;    // the front half overlap-added the window into the wet ring (k_inv_wet)
    bool corr_done = false;    // the Q1/Q2 prefix sums of the batch are final (they rode along with the front half's launches)
    bool drop_done = false;    // Q8 regime: the front half summed the cut terms of the whole batch into d_dropbuf (k_drop_fft)
    int out_from = -1;         // >= 0: the front half finished the output of blocks >= out_from itself (k_inv_wet<true>)
};

// One voice as the MAC sees it
struct ActiveVoice {
    int v;                 // gain-table row
    const IrEntry *ir0, *ir1;
    int p_end;             // partitions to sweep (multiple of 16), before sharding
    bool uniform;          // every slot of the window carries the same gains
    float4 ugain;
};

// Sample the parameters, advance the cross-fade, stage the per-block table of a
// batch of T blocks starting at block t_front, and describe the batch in `ctx`.
struct Staged {
    BatchCtx ctx;
    BlockParams* d_ptab;
    float4* d_sums;
    BlockParams first;  // host copy of the first block's parameters
    ActiveVoice act[MC_MAXV];
    int nact = 0;
};


// JACK path: a period launched one call ahead, parked on its doorbell (process_one)
struct JackPre {
    bool valid = false;
    uint64_t block = 0;   // the block it will finish
    unsigned seq = 0;     // its doorbell / completion value
    mc_cc_value cc[2];    // the parameters it was staged with
    Staged st;
    int pm = 1;           // blocks per period it was staged for
    bool carries_sweep = false;  // its kernel also runs the sweep of block + 1 ...
    int carried_vir[2];          // ... for this IR pair
};

// state of the single-transform form (singlefft.hip.h / singlefft_host.hip.h)
struct SfState {
    int N = 0, M = 0, AT = 1;
    float2* d_live = nullptr;  // [half][ch][N/2] live IR spectra (the reference's irFFT, conv.h:72); bin d + M c at [d][c]
    float2* d_W = nullptr;     // [M][512] packed output spectrum
    float2* d_T = nullptr;     // [512][M] between the inverse passes; IR preparation: the packed taps, then U
    float2* d_Z = nullptr;     // [N] IR preparation
    float* d_acc = nullptr;    // [ch][N] accumulators (conv.h:74 residual): a ring, `base` = slot of the next output frame
    unsigned* d_ctr = nullptr;
    unsigned base = 0;
    float* d_io[4] = {nullptr, nullptr, nullptr, nullptr};  // staging of host-buffer batches, io_cap frames each
    size_t io_cap = 0;
    size_t lds_bytes = 0;
    bool stockham = false;  // MCCONV_SF_STOCKHAM: pass 2 of the inverse through the LDS transform whatever the size
};

struct mc_engine {
    mc_config cfg;
    int device = 0;
    SfState* sf = nullptr;  // != null: the engine runs the reference's single-transform form (singlefft.hip.h)
    hipStream_t own_stream = nullptr, stream = nullptr;
    int Tcap = 0;  // blocks the scratch buffers and rings are sized for: >= Tmax, and enough to re-render history in few launches
    int Tmax = 0, Pcap = 0, Pstride = 0, ring = 0, sr = 0, wr = 0, rc = 0, nchunk = 2, Tstream = 0;
    int stream_threshold = 0;
    int pm = 1;  // blocks per reference call (JACK period / 256): 1, 2 or 4
    int stream_nt = 256;
    IrEntry irs[kMaxIrs + kMixIrs];
    int mix_buf[2] = {0, 0};  // which of its two merged-IR entries half i used last
    int nirs = 0;

    uint2* d_fdl16 = nullptr;  // fp16 mirror of the delay line (precision = fp16)
    bool half = false;
    float4* d_Yc = nullptr;  // [256][Tcap] partition sums combined from the fast-FIR components
    float2* d_stash = nullptr;  // second-level transform: spectra parked between the transforms and the products
    size_t stash_chunks = 0;  // capacity of d_stash in sequences per bin (chunks x sequences per chunk)
    // Pipelined batches (mc_config.pipeline): the inverse transforms and the post stage of batch k run on a second
    // stream under the MAC of batch k + 1.  Scratch that both touch is double-buffered by batch parity.
    bool pipelined = false;
    hipStream_t post_stream = nullptr;
    hipEvent_t ev_mac[2][2], ev_corr[2], ev_post[2];
    bool post_pending[2] = {false, false};
    float4 *d_Ybuf[2] = {nullptr, nullptr}, *d_Ycbuf[2] = {nullptr, nullptr}, *d_partbuf[2] = {nullptr, nullptr};
    float4 *d_tail = nullptr, *d_tailbuf[2] = {nullptr, nullptr};  // chunk partials of the last 2^levels blocks (fast-FIR form)
    float4 *d_fdl = nullptr, *d_slotgain = nullptr, *d_Y = nullptr, *d_part = nullptr, *d_sums = nullptr;
    float *d_seg = nullptr, *d_wet = nullptr;
    double* d_cring = nullptr;
    double* d_ctot = nullptr;   // [ceil(Tmax/256)][4] chunk totals of the Q1/Q2 prefix sums
    // predelay epochs: what blocks played under earlier predelays still owe, by absolute output sample
    float *d_res_mac = nullptr, *d_res_fix = nullptr;  // [2][rr] each
    int rr = 0;
    uint64_t res_end = 0;     // the rings hold output samples < res_end
    uint64_t epoch_b0 = 0;    // first block of the live epoch
    uint64_t cur_delay = 0;   // its predelay
    float* d_xhist = nullptr;   // [2][xr] input history (Q8 pass)
    float4* d_gring = nullptr;  // [MC_MAXV][rc] wet gains of past blocks (Q8 pass)
    int xr = 0;
    BlockParams* d_ptab = nullptr;
    float2* d_tw = nullptr;
    float* d_io[4] = {nullptr, nullptr, nullptr, nullptr};  // in1, in2, outL, outR staging for host-pointer calls
    int Thost = 0;                                          // blocks per chunk of a host-buffer batch staged through h_io
    // host-buffer batches whose buffers are pinned (mc_host_alloc, hipHostMalloc, hipHostRegister): chunks of Tdev blocks,
    // H2D / compute / D2H on three streams, two chunks in flight; device staging allocated on first use
    // (three staging sets: with two, copy-in waiting for the kernels of chunk k - 2 and the kernels waiting for the copy-out of
    // chunk k - 2 lock the three streams into taking turns - 2.9 instead of 1.5 ms per 32320-block chunk, scripts/pcie_pipeline_probe.py)
    float* d_pio[3][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    int Tdev = 0;
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    hipEvent_t ev_h2d[3] = {nullptr, nullptr, nullptr}, ev_comp[3] = {nullptr, nullptr, nullptr}, ev_d2h[3] = {nullptr, nullptr, nullptr};
    float* h_io = nullptr;                                  // pinned mirror of d_io, 4 * Thost * 256
    float* hd_io = nullptr;                                 // device-side address of h_io (mapped, zero-copy)
    unsigned* h_flag = nullptr;                             // completion word of the single-period path (mapped)
    unsigned* hd_flag = nullptr;
    unsigned flag_seq = 0;
    unsigned* d_done_ctr = nullptr;  // workgroups of the period's last kernel that have finished
    bool spin_wait = true;
    BlockParams* h_ptab[kStageBufs] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ptab_ev[kStageBufs];
    bool ptab_ev_used[kStageBufs] = {false, false, false, false};
    int ptab_next = 0;

    // parameters: written by any thread (mc_set_params, mc_handle_cc), sampled at the start of a process call without a
    // lock (params_handoff.h); last_gen = the generation of the pair the last process call ran on (mc_debug_read item 7)
    ParamHandoff ph;
    uint64_t last_gen = 0;

    // Cross-fade state.  The reference keeps live spectra irFFT_i and pulls them towards
    // wet_i * H_sel_i every block: irFFT += (wet H_sel - irFFT)/(vsteps+5) (conv.cu:27, 339-353).
    // irFFT_i is therefore always sum_j c_ij H_j with c_ij <- c_ij (1 - 1/(v+5)) + [j == sel_i] wet_i/(v+5):
    // one coefficient per IR that has been selected recently.  coef[i][v] is that coefficient for
    // the IR in voice slot v of half i.
    struct VoiceSlot {
        int ir = -1;           // IR index, -1 = free
        double coef = 0.0;
        uint64_t last_nz = 0;  // last block whose gain for this slot was non-zero
        bool ever = false;
    } voice[2][MC_MAXV];
    float last_g[MC_MAXV][4];
    uint64_t gain_change_block[MC_MAXV] = {0, 0, 0};  // block at which voice v's gains last changed
    uint64_t last_nz_voice[MC_MAXV] = {0, 0, 0};
    bool voice_ever[MC_MAXV] = {false, false, false};

    uint64_t t_abs = 0;    // blocks finished (front + back done)
    uint64_t t_front = 0;  // blocks whose front half (FFT, MAC, inverse, overlap-add) has been issued
    // up to kPipe batches may sit between their front and back halves (sharded
    // operation overlaps the cross-GPU reduce of batch k with the MAC of batch k+1)
    typedef ::BatchCtx BatchCtx;
    BatchCtx pipe[2];
    int pipe_head = 0, pipe_count = 0;
    uint64_t batch_seq = 0;
    // speculative MAC of the next single block (partitions >= 1 do not depend on the next input)
    bool speculate = true, spec_valid = false;
    // Q8 regime, JACK path: the cut terms a parked launch summed for the period after its own (SweepArgs.drop_next): valid for that block while
    // predelay, epoch and the voices' IRs are what they were (everything else they depend on is at least n_ref frames old)
    struct DropSpec {
        bool valid = false;
        uint64_t block = 0, predelay = 0, epoch_b0 = 0;
        int vir[2][MC_MAXV];
        const float* buf = nullptr;
    } dspec;
    bool carry_drop = true;  // MCCONV_CARRY_DROP=0: always a launch of its own (k_drop_period_fft) (measurement)
    uint64_t spec_block = 0;
    int spec_vir[2][MC_MAXV];
    int spec_nact = 0;
    hipEvent_t ev_tail = nullptr;
    float4* d_part_jack[2] = {nullptr, nullptr};  // JACK path: the sweep's partials, double-buffered by block parity
    bool td_fft = true;                           // Q8 regime, batches: the cut terms in the frequency domain (MCCONV_TD_FFT=0: time-domain tiles)
    float2* d_dropbuf = nullptr;                  // batches, Q8 regime: the cut terms of a batch's blocks [Tmax][256] {L, R} (k_drop_fft), allocated by the first such batch
    float* d_drop[2] = {nullptr, nullptr};        // JACK path, Q8 regime: a period's tail-drop terms [2][1024] (k_drop_period), by period parity
    JackPre pre;                     // the period parked one call ahead
    bool park = true;                // MCCONV_NO_PARK=1: every period launched when it arrives
    unsigned long long park_ticks = 10000000ull;  // a parked tail gives up after this many 100 MHz ticks (100 ms; MCCONV_PARK_MS)
    unsigned long long* h_bell = nullptr;         // mapped: {sequence number, command} the parked tail polls
    unsigned long long* hd_bell = nullptr;
    // JACK path on a large-BAR system: doorbell and period input live in fine-grained DEVICE memory that the CPU writes
    // straight through the BAR (posted writes), so the parked tail polls and reads locally instead of over PCIe
    // (MCCONV_BAR_IO=0: mapped host memory as before).  16 floats = the doorbell's line, then in1, in2 (room for 1024 frames each).
    float* d_bar = nullptr;
    bool bar_io = false;
    // Tagged I/O of the 256-frame JACK path (TailArgs::in_gran / out_gran): input granules at byte 16384 of d_bar, output granules
    // in mapped host memory.  On where the BAR path is (MCCONV_TAGGED_IO=0: doorbell + completion word as in round 2).
    bool tio = false;
    bool tio_long = false;  // ... also for 512 / 1024-frame periods (MCCONV_TAGGED_IO=2; measured slower there: twice the bytes over the link)
    unsigned long long* h_gran = nullptr;   // [2][256] output granules {value, sequence number}
    unsigned long long* hd_gran = nullptr;
    bool host_out_direct = true;  // MCCONV_HOST_OUT_DIRECT=0: pinned-buffer batches copy their output out instead of storing it to the host
    unsigned* h_exited = nullptr;                 // mapped: sequence number of a parked tail that gave up on its own
    unsigned* hd_exited = nullptr;
    // how often a parked period was used / gave up on its own (host away > park_ms) / was told to give up: mc_debug_read item 6
    uint64_t n_park_hit = 0, n_park_timeout = 0, n_park_cancel = 0;
    unsigned* d_tailform = nullptr;  // 256-frame tails by the form partition 0 took {frequency domain, time domain}, counted by the kernel (mc_debug_read item 16)
    int tail_form = 0;               // lab build, MCCONV_TAIL_FORM=td|fd: one form for every 256-frame tail (0: the first look decides)
    uint64_t n_drop_carried = 0;  // JACK path, Q8 regime: periods whose cut terms came with the launch before theirs (mc_debug_read item 9, fourth word)
    uint64_t n_mac_form[3] = {0, 0, 0};  // batches whose partition sums took the fused / split second-level transform / the resident MAC (mc_debug_read item 10)
    uint64_t n_drop_fft = 0, n_drop_ahead = 0, n_drop_tiles = 0;  // Q8 regime: batches by the form their cut terms took (mc_debug_read item 9)
#ifdef MC_JACK_TRACE
    double tr_launch = 0, tr_flag = 0, tr_total = 0, tr_kernel = 0, tr_gap = 0, tr_out = 0, tr_clk = 0, tr_c[3] = {0, 0, 0};
    unsigned long long tr_prev_end = 0;
    long tr_n = 0;
#endif
    bool fft2 = true;     // long batches: second-level transform along the block axis instead of the MAC
    bool fft2_fused = true;  // ... in the fused 8192-point form where it applies
    bool debug_addr = false;  // MCCONV_DEBUG_ADDR: print the device ranges k_g2_mac touches at its first launch (fault triage)
    int g2_grid = 1 << 30;   // workgroups of k_g2_mac, capped by the number of (bin, chunk) items (default: one per item); MCCONV_G2_GRID
    int g2_pmax = 5632;      // longest block-axis convolution (partitions) the fused 8192-point form takes (MCCONV_G2_PMAX):
                             // measured crossover with the split 16384-point form ~5700 (30 s IRs, P = 5168: 0.25 vs 0.27 ms
                             // for a third more blocks)
    int64_t fft2_work = 300000;  // blocks x partitions from which a batch takes the fused second-level form (MCCONV_FFT2_WORK)
    int g2_pmin = 16;        // shortest block-axis convolution (partitions, uniform gains) of an unsharded engine that takes the
                             // second-level transform (MCCONV_G2_PMIN; round 1: 256)
    bool corr_ride = true;   // MCCONV_CORR_RIDE=0: the Q1/Q2 prefix steps as launches of their own (measurement)
    bool drop_ahead = true;  // MCCONV_DROP_AHEAD=0: Q8 regime: every cut term through k_drop_fft, none summed by the forward transforms (measurement)
    bool fuse_drop = true;   // MCCONV_FUSE_DROP=0: Q8 regime: the output through k_post<3> even where the inverse transforms could finish it (measurement)
    bool fuse_out = true;    // MCCONV_FUSE_OUT=0: the output always through k_post (measurement)
    unsigned* d_cticket = nullptr;  // ticket counter of the riding prefix-sum workgroups (see CorrArgs)
    unsigned* d_cflag = nullptr;    // [ceil(Tmax/256)] launch sequence number per published chunk total
    unsigned cticket_base = 0, cflag_seq = 0;
    // k_g2_duo (round 3): one 1024-thread workgroup per CU whose halves run the items' phases in lockstep, one phase apart.
    // Measured equal to k_g2_mac at 20 chunks per bin and slower below (profiles/r3_g2_ablation.md): NOT the default;
    // MCCONV_G2_DUO=1 selects it for launches of at least _DUO_MINCH chunks per bin; _DUO_GRID: its workgroups (a multiple
    // of 8, default = the CUs)
    bool g2_duo = false;
    int g2_duo_minch = 3, g2_duo_grid = 256;
    bool g2_wide = false;    // MCCONV_G2_WIDE=1: the one-workgroup-per-CU form of the kernel (1024 threads, both sequences in LDS)
    // Long settled batches as overlap-save segments of 512 x 8192 frames (ossave.hip.h): whole batches on one fp32 engine whose
    // window carries one set of gains, outside the Q8 regime.  Buffers and the spectra of the sounding (IR set, gains) are
    // made by the first batch that takes the form.
    bool os_hold = false;       // set around a batch whose output pointers are mapped HOST memory (mc_process_batch with pinned buffers)
    bool os_on = true;          // MCCONV_OS=0: such batches through the second-level transform as before (measurement)
    int os_min_blocks = 12288;  // shortest batch that takes the form (a segment costs the same however little of it is used)
    float4* d_os_T = nullptr;   // [segments][256 row pairs][8192] {row k1, row 512 - k1} between the passes
    size_t os_T_segs = 0;
    float4* d_os_part = nullptr;  // [segments][512 tiles][512] the tiles' sixteenths of the blocks' sums {S1, S2, A1, A2}
    size_t os_part_segs = 0;
    float4 *d_os_SP = nullptr, *d_os_SP0 = nullptr;  // spectra A, B of the cached key, in the row pass's order
    float* d_os_planes = nullptr;  // gain-weighted taps, four planes of os_planes_n floats
    size_t os_planes_n = 0;
    struct OsKey {
        bool valid = false;
        int n = 0;
        int ir0[MC_MAXV], ir1[MC_MAXV];
        float g[MC_MAXV][4];
        uint64_t gen = 0;
    } os_key;
    uint64_t ir_gen = 0;                  // counts changes of any IR's taps (load, reload, merge)
    uint64_t n_os[2] = {0, 0};            // batches that took the form, spectra builds (mc_debug_read item 11)
    // the small launches of such a batch - prefix sums, the tail's delay-line slots, the last block's segment - run on a side
    // stream beside the three passes (MCCONV_OS_SIDE=0: in line, measurement)
    bool os_side = true;
    hipStream_t os_stream = nullptr;
    hipEvent_t os_ev[3] = {nullptr, nullptr, nullptr};
    int ffa_levels = 3;   // resident MAC in fast-FIR form (up to this many nested levels) when batch and IR are long enough
    bool sliced = false;  // block-sliced calls keep no wet / segment history outside their slices
    bool inv_to_wet = true;  // whole-batch path: k_inv_wet + ring-reading k_post (MCCONV_INV_WET=0: k_inv + segment ring)
    int slice_first = -1;  // ... and transform only what their windows reach: the slice start must not move
    bool uniform_valid[2] = {false, false};
    BlockParams uniform_bp[2];

    // avgRuntime (conv.cu:454-462): first 10 calls discarded
    double runtime_ms = 0;
    int nruns = -10;

    // kernel timing
    bool ktiming = false;
    hipEvent_t kev[kEvPool][2];
    uint32_t kev_blocks[kEvPool];
    bool kev_stamped[kEvPool];            // the launch carries in-kernel time stamps (a single period's sweep)
    unsigned long long* d_stamps = nullptr;  // [kStampSlots][MC_STAMP_WGS][2] {start, end} per workgroup in 100 MHz ticks
    int kev_n = 0;
    bool kev_created = false;
    mc_kernel_stats ks;
};

namespace {

size_t y_capacity(const mc_engine* e) { return std::max<size_t>(27 * ((size_t)e->Tcap / 8 + 256) + 1024, 8192); }  // blocks per bin

void host_twiddles(std::vector<float2>& tw) {
    tw.resize(FFT_N);
    for (int m = 0; m < FFT_N; m++) {
        double a = -2.0 * M_PI * (double)m / (double)FFT_N;
        tw[m] = make_float2((float)cos(a), (float)sin(a));
    }
}

void ring_bell(mc_engine* e, unsigned seq, unsigned command) {
    // the period was written before: release order makes it visible to the tail that acquires the doorbell
    if (e->bar_io) {
        _mm_sfence();  // the period's write-combined stores leave before the doorbell's
        __atomic_store_n(reinterpret_cast<unsigned long long*>(e->d_bar), ((unsigned long long)command << 32) | seq, __ATOMIC_RELEASE);
        _mm_sfence();  // ... and the doorbell does not wait in the write-combining buffer
        return;
    }
    __atomic_store_n(e->h_bell, ((unsigned long long)command << 32) | seq, __ATOMIC_RELEASE);
}

// JACK path: a period launched ahead is parked on the engine's stream (process_one).  Before anything else may use the
// stream, or change what the parked tail has already loaded, it is told to give up.  It has written nothing; the next
// mc_process launches that period again.
void unpark(mc_engine* e) {
    // (the cut terms a parked launch summed for the period after its own belong to that launch: without it - told to give up,
    // or timed out and launched again, which rewrites the same buffer with ANOTHER block's terms - they are not to be trusted)
    e->dspec.valid = false;
    if (e->pre.valid) {
        ring_bell(e, e->pre.seq, 1);
        e->pre.valid = false;
        e->n_park_cancel++;
    }
}
int leave_jack_path(mc_engine* e) {
    unpark(e);
    return MC_OK;
}

hipError_t reset_stamps(mc_engine* e) {
    return hipMemset(e->d_stamps, 0, sizeof(unsigned long long) * 2 * MC_STAMP_WGS * kStampSlots);
}

int drain_kernel_events(mc_engine* e) {
    if (!e->kev_n) return MC_OK;
    unpark(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<unsigned long long> stamps;
    bool any_stamped = false;
    for (int i = 0; i < e->kev_n; i++) any_stamped = any_stamped || e->kev_stamped[i];
    if (any_stamped && e->d_stamps) {
        stamps.resize((size_t)2 * MC_STAMP_WGS * kStampSlots);
        HIP_TRY(hipMemcpy(stamps.data(), e->d_stamps, sizeof(unsigned long long) * stamps.size(), hipMemcpyDeviceToHost));
        HIP_TRY(reset_stamps(e));
    }
    for (int i = 0; i < e->kev_n; i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->kev[i][0], e->kev[i][1]));
        if (e->kev_stamped[i] && !stamps.empty() && i < kStampSlots) {
            unsigned long long lo = ~0ull, hi = 0;
            const unsigned long long* sp = stamps.data() + (size_t)2 * MC_STAMP_WGS * i;
            for (int w = 0; w < MC_STAMP_WGS; w++)
                if (sp[2 * w]) {
                    lo = std::min(lo, sp[2 * w]);
                    hi = std::max(hi, sp[2 * w + 1]);
                }
            if (hi > lo) ms = (float)((double)(hi - lo) * 1e-5);  // 100 MHz ticks -> ms
        }
        e->kev_stamped[i] = false;
        e->ks.launches++;
        e->ks.blocks += e->kev_blocks[i];
        e->ks.total_ms += ms;
        e->ks.last_ms = ms;
    }
    e->kev_n = 0;
    return MC_OK;
}

// Pipelined mode: nothing of an earlier batch is still running on the post stream
int drain_post(mc_engine* e) {
    if (!e->pipelined) return MC_OK;
    if (e->post_pending[0] || e->post_pending[1]) HIP_TRY(hipStreamSynchronize(e->post_stream));
    e->post_pending[0] = e->post_pending[1] = false;
    return MC_OK;
}

// Pipelined mode: the engine's stream waits for everything issued on the post stream so far
int fence_post(mc_engine* e) {
    if (!e->pipelined) return MC_OK;
    for (int par = 0; par < 2; par++)
        if (e->post_pending[par]) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_post[par], 0));
    return MC_OK;
}

int zero_state(mc_engine* e) {
    {
        int rc = leave_jack_path(e);
        if (rc) return rc;
    }
    {
        int rc = drain_post(e);
        if (rc) return rc;
    }
    HIP_TRY(hipMemsetAsync(e->d_fdl, 0, sizeof(float4) * (size_t)MC_NB * e->ring, e->stream));
    if (e->d_fdl16) HIP_TRY(hipMemsetAsync(e->d_fdl16, 0, sizeof(uint2) * (size_t)MC_NB * e->ring, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_slotgain, 0, sizeof(float4) * (size_t)MC_MAXV * e->ring, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_seg, 0, sizeof(float) * (size_t)e->sr * 2 * FFT_N, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_wet, 0, sizeof(float) * 2 * (size_t)e->wr, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_cring, 0, sizeof(double) * 4 * (size_t)e->rc, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_xhist, 0, sizeof(float) * 2 * (size_t)e->xr, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_gring, 0, sizeof(float4) * (size_t)MC_MAXV * e->rc, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_res_mac, 0, sizeof(float) * 2 * (size_t)e->rr, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_res_fix, 0, sizeof(float) * 2 * (size_t)e->rr, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->res_end = e->epoch_b0 = e->cur_delay = 0;
    e->sliced = false;
    e->slice_first = -1;
    for (int i = 0; i < 2; i++)
        for (int v = 0; v < MC_MAXV; v++) e->voice[i][v] = mc_engine::VoiceSlot();
    for (int v = 0; v < MC_MAXV; v++) {
        for (int c = 0; c < 4; c++) e->last_g[v][c] = 0.f;
        e->gain_change_block[v] = 0;
        e->last_nz_voice[v] = 0;
        e->voice_ever[v] = false;
    }
    e->t_abs = e->t_front = 0;
    e->spec_valid = e->dspec.valid = false;
    e->uniform_valid[0] = e->uniform_valid[1] = false;
    e->pipe_head = e->pipe_count = 0;
    return MC_OK;
}

int retire_epoch(mc_engine* e, uint64_t new_delay, bool force);

// An IR's spectra changed (load, reload, mix): everything derived from them - the second-level spectra and the fast-FIR
// components - is rebuilt on next use.  (Round 2 built the fast-FIR components eagerly at every load: 60 MB and three
// launches per 10 s IR for a form that only MCCONV_FFT2=0 selects.)
void invalidate_derived(IrEntry& ir) {
    ir.h2_valid = ir.g2_valid = ir.tail_valid = false;
    ir.hp_valid[0] = ir.hp_valid[1] = ir.hp_valid[2] = false;
}
// The last partitions of an IR's spectra, partition-major, for k_drop_fft: every workgroup of that launch reads the same few
// partitions of all 256 bins, and in the bin-major bank those entries lie Pstride * 16 bytes apart (one 128-byte line fetched per
// 16 bytes used, all of them in one L2 channel).  Worth 2-4 % of the step (scripts/gpu_q8_pd.sh, MCCONV_HTAIL=0 turns it off): the launch
// is bound by its latency per workgroup, not by these reads.  kTailSpan partitions cover every predelay the controllers can reach
// (8192 frames = 32 partitions, + 2); a term outside falls back to the bank.
constexpr int kTailSpan = 48;
int ensure_htail(mc_engine* e, const IrEntry* irc) {
    IrEntry& ir = *const_cast<IrEntry*>(irc);
    static const bool off = LAB_ENV("MCCONV_HTAIL") && std::atoi(LAB_ENV("MCCONV_HTAIL")) == 0;
    if (ir.tail_valid || !ir.d_H || off) return MC_OK;
    if (!ir.d_Htail) HIP_TRY(hipMalloc(&ir.d_Htail, sizeof(float4) * (size_t)kTailSpan * MC_NB));
    ir.tail_p0 = std::max(0, ir.P - kTailSpan);
    hipLaunchKernelGGL(k_h_tail, dim3(kTailSpan), dim3(MC_NB), 0, e->stream, (const float4*)ir.d_H, e->Pstride, ir.tail_p0, ir.P, ir.d_Htail);
    HIP_TRY(hipGetLastError());
    ir.tail_valid = true;
    return MC_OK;
}
// whether the fast-FIR form of level lvl (1..3) exists for this engine at all
bool hp_possible(const mc_engine* e, int lvl) {
    return !e->half && e->Pstride >= 128 && (e->Pstride >> lvl) >= 32;  // the fp16 MAC streams; small engines never use the fast form
}
// fast-FIR components of an IR's spectra (level lvl) for the resident MAC, built on first use
int ensure_hp(mc_engine* e, const IrEntry* irc, int lvl) {
    IrEntry& ir = *const_cast<IrEntry*>(irc);
    if (ir.hp_valid[lvl - 1]) return MC_OK;
    const size_t n = (size_t)(lvl == 1 ? 3 : (lvl == 2 ? 9 : 27)) * MC_NB * (e->Pstride >> lvl);
    if (!ir.d_Hp[lvl - 1]) HIP_TRY(hipMalloc(&ir.d_Hp[lvl - 1], sizeof(float4) * n));
    hipLaunchKernelGGL(k_polyphase, dim3(2048), dim3(256), 0, e->stream, ir.d_H, ir.d_Hp[lvl - 1], e->Pstride, lvl);
    HIP_TRY(hipGetLastError());
    ir.hp_valid[lvl - 1] = true;
    return MC_OK;
}

// second-level spectra of an IR (transform of its partition sequence along the block axis), built on first use
int ensure_fft2(mc_engine* e, const IrEntry* irc) {
    IrEntry& ir = *const_cast<IrEntry*>(irc);
    if (ir.h2_valid) return MC_OK;
    if (!ir.d_H2) HIP_TRY(hipMalloc(&ir.d_H2, sizeof(float2) * (size_t)2 * 257 * F2_N));
    // a partition shard transforms its own run of the partition sequence, H[pb .. pe)
    const int pb = std::min<int>((int)e->cfg.part_begin, ir.P), pe = e->cfg.part_end ? std::min<int>((int)e->cfg.part_end, ir.P) : ir.P;
    hipLaunchKernelGGL(k_fft2_ir, dim3(257, 2), dim3(F2_THREADS), 0, e->stream, ir.d_H + pb, e->Pstride, std::min(std::max(pe - pb, 0), F2_N), ir.d_H2);
    HIP_TRY(hipGetLastError());
    ir.h2_valid = true;
    return MC_OK;
}

int ensure_g2(mc_engine* e, const IrEntry* irc) {
    IrEntry& ir = *const_cast<IrEntry*>(irc);
    if (ir.g2_valid) return MC_OK;
    if (!ir.d_G2) HIP_TRY(hipMalloc(&ir.d_G2, sizeof(float2) * (size_t)2 * 257 * G2_N));
    const int pb = std::min<int>((int)e->cfg.part_begin, ir.P), pe = e->cfg.part_end ? std::min<int>((int)e->cfg.part_end, ir.P) : ir.P;
    hipLaunchKernelGGL(k_g2_ir, dim3(257), dim3(G2_THREADS), 0, e->stream, ir.d_H + pb, e->Pstride, std::min(std::max(pe - pb, 0), G2_N), ir.d_G2);
    HIP_TRY(hipGetLastError());
    ir.g2_valid = true;
    return MC_OK;
}

// More IRs are cross-fading in half i than there are voices.  The reference's live spectrum is sum_j c_j H_j and
// every deselected coefficient decays by the same factor per block (f_interpolate, conv.cu:27), so all of the
// half's current IRs - the one being deselected included - merge into ONE spectrum M = sum_j c_j H_j that carries
// on with coefficient 1.  The blocks played so far were weighted IR by IR: everything they still owe is rendered
// into the residual rings first (as for a predelay change), the live pipeline restarts from silence, and the
// half's voices become {M} plus free slots.
int consolidate_voices(mc_engine* e, int i) {
    int rc = retire_epoch(e, e->cur_delay, true);
    if (rc) return rc;
    mc_engine::VoiceSlot* vs = e->voice[i];
    const int buf = 1 - e->mix_buf[i];
    const int midx = kMaxIrs + i * 2 + buf;
    IrEntry& M = e->irs[midx];
    const size_t nH = (size_t)MC_NB * e->Pstride * 4, nh_cap = (size_t)e->cfg.n_ref * 2;
    if (!M.d_H) HIP_TRY(hipMalloc(&M.d_H, sizeof(float) * nH));
    if (!M.d_h) HIP_TRY(hipMalloc(&M.d_h, sizeof(float) * nh_cap));
    MixSrc sH, sh;
    std::memset(&sH, 0, sizeof(sH));
    std::memset(&sh, 0, sizeof(sh));
    double sums[4] = {0, 0, 0, 0}, bound16 = 0.0;
    uint64_t taps = 1;
    int P = 1;
    for (int v = 0; v < MC_MAXV; v++) {
        if (vs[v].ir < 0 || vs[v].coef == 0.0 || !e->irs[vs[v].ir].d_H) continue;
        const IrEntry& src = e->irs[vs[v].ir];
        sH.p[v] = reinterpret_cast<const float*>(src.d_H);
        sH.n[v] = nH;
        sH.c[v] = (float)vs[v].coef;
        sh.p[v] = reinterpret_cast<const float*>(src.d_h);
        sh.n[v] = (size_t)src.taps * 2;
        sh.c[v] = (float)vs[v].coef;
        for (int c = 0; c < 4; c++) sums[c] += vs[v].coef * src.sums[c];
        taps = std::max(taps, src.taps);
        P = std::max(P, src.P);
        bound16 += std::fabs(vs[v].coef) * 16384.0 / (double)src.scale16;  // |H_j| < 2^14 / scale16_j
    }
    hipLaunchKernelGGL(k_mix, dim3(2048), dim3(256), 0, e->stream, reinterpret_cast<float*>(M.d_H), nH, sH);
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(256), 0, e->stream, reinterpret_cast<float*>(M.d_h), (size_t)taps * 2, sh);
    HIP_TRY(hipGetLastError());
    invalidate_derived(M);
    e->ir_gen++;
    std::memcpy(M.sums, sums, sizeof(sums));
    M.taps = taps;
    M.P = P;
    if (e->half) {
        int ex = 0;
        if (bound16 > 0.0) std::frexp(bound16, &ex);
        M.scale16 = std::ldexp(1.0f, 13 - ex);
        if (!M.d_H16) HIP_TRY(hipMalloc(&M.d_H16, sizeof(uint2) * (nH / 4)));
        hipLaunchKernelGGL(k_to_half, dim3(1024), dim3(256), 0, e->stream, M.d_H, M.d_H16, nH / 4, M.scale16);
        HIP_TRY(hipGetLastError());
    }
    for (int v = 0; v < MC_MAXV; v++) vs[v] = mc_engine::VoiceSlot();
    vs[0].ir = midx;
    vs[0].coef = 1.0;
    e->mix_buf[i] = buf;
    return MC_OK;
}

// voice slot of half i that holds IR `ir`; allocates a free (fully retired) slot when it is new
int voice_slot_for(mc_engine* e, int i, int ir, uint64_t block, int* err) {
    mc_engine::VoiceSlot* vs = e->voice[i];
    for (int v = 0; v < MC_MAXV; v++)
        if (vs[v].ir == ir) return v;
    for (int pass = 0; pass < 2; pass++) {
        // a slot may be reused once its last non-zero gain has left every window (longest IR = Pcap blocks)
        for (int v = 0; v < MC_MAXV; v++)
            if (vs[v].ir < 0 || (vs[v].coef == 0.0 && (!vs[v].ever || vs[v].last_nz + (uint64_t)e->Pcap + 1 < block))) {
                vs[v] = mc_engine::VoiceSlot();
                vs[v].ir = ir;
                return v;
            }
        if (pass == 0) {
            const int rc = consolidate_voices(e, i);
            if (rc) {
                *err = rc;
                return 0;
            }
        }
    }
    *err = fail(MC_ERR_STATE, "no voice slot");
    return 0;
}

// Build the per-block parameter table for T blocks starting at block e->t_front (host, double).
// Advances the cross-fade coefficients exactly like f_interpolate + vsteps-- (conv.cu:27, 339-353).
// Returns pstride (0 = one entry serves all blocks).
int build_params(mc_engine* e, int T, mc_cc_value (&cc)[2], BlockParams** out_tab, int* out_n, int* err) {
    BlockParams* tab = e->h_ptab[e->ptab_next];
    bool all_same = true;
    for (int t = 0; t < T; t++) {
        const uint64_t blk = e->t_front + (uint64_t)t;
        BlockParams& bp = tab[t];
        if (t == 1) {
            // steady state (every cross-fade coefficient has reached its target): the remaining blocks of the batch
            // repeat block 0 - one table entry serves all, whatever the batch length
            bool settled = true;
            for (int i = 0; i < 2; i++) {
                const int sv = voice_slot_for(e, i, (int)cc[i].select, blk, err);
                for (int v = 0; v < MC_MAXV; v++) {
                    const mc_engine::VoiceSlot& s = e->voice[i][v];
                    if (s.ir >= 0 && s.coef != ((v == sv) ? (double)cc[i].wet : 0.0)) settled = false;
                }
            }
            if (settled) {
                const uint64_t last = e->t_front + (uint64_t)T - 1;
                for (int v = 0; v < MC_MAXV; v++) {
                    const float* g = tab[0].g[v];
                    if (g[0] != 0.f || g[1] != 0.f || g[2] != 0.f || g[3] != 0.f) {
                        e->last_nz_voice[v] = last;
                        for (int i = 0; i < 2; i++)
                            if (g[0 * 2 + i] != 0.f || g[1 * 2 + i] != 0.f) e->voice[i][v].last_nz = last;
                    }
                }
                break;
            }
        }
        std::memset(&bp, 0, sizeof(bp));
        // the reference advances its live spectra once per onProcess call: with periods of 512 / 1024 frames the
        // 2 / 4 internal blocks of a call share the call's coefficients
        for (int i = 0; i < 2 && (blk % (uint64_t)e->pm) == 0; i++) {
            const double wet = (double)cc[i].wet;
            const double div = (double)(cc[i].vsteps + 5);
            const int sv = voice_slot_for(e, i, (int)cc[i].select, blk, err);
            if (*err) return 0;
            for (int v = 0; v < MC_MAXV; v++) {
                mc_engine::VoiceSlot& s = e->voice[i][v];
                if (s.ir < 0) continue;
                const double target = (v == sv) ? wet : 0.0;
                s.coef += (target - s.coef) / div;
                // settle: the recurrence converges geometrically; snap once the distance is below double
                // resolution of the reference's float spectra (keeps steady-state tables bit-identical)
                if (std::fabs(target - s.coef) <= 1e-300 + 4e-16 * std::fabs(target) || (target == 0.0 && std::fabs(s.coef) < 1e-14))
                    s.coef = target;
            }
            if (cc[i].vsteps > 0) cc[i].vsteps--;
        }
        for (int i = 0; i < 2; i++) {
            const double lvl = (double)cc[i].level;
            const double pl = pan_l((double)cc[i].panWet), pr = pan_r((double)cc[i].panWet);
            for (int v = 0; v < MC_MAXV; v++) {
                mc_engine::VoiceSlot& s = e->voice[i][v];
                const double c = s.ir >= 0 ? s.coef : 0.0;
                bp.G[v][0 * 2 + i] = pl * lvl * c;
                bp.G[v][1 * 2 + i] = pr * lvl * c;
                bp.g[v][0 * 2 + i] = (float)bp.G[v][0 * 2 + i];
                bp.g[v][1 * 2 + i] = (float)bp.G[v][1 * 2 + i];
                if (bp.g[v][0 * 2 + i] != 0.f || bp.g[v][1 * 2 + i] != 0.f) {
                    s.last_nz = blk;
                    s.ever = true;
                }
            }
            bp.d[0 * 2 + i] = (float)((double)cc[i].dry * pan_l((double)cc[i].panDry) * lvl);
            bp.d[1 * 2 + i] = (float)((double)cc[i].dry * pan_r((double)cc[i].panDry) * lvl);
        }
        for (int v = 0; v < MC_MAXV; v++) {
            if (std::memcmp(e->last_g[v], bp.g[v], sizeof(bp.g[v])) != 0) {
                std::memcpy(e->last_g[v], bp.g[v], sizeof(bp.g[v]));
                e->gain_change_block[v] = blk;
            }
            if (bp.g[v][0] != 0.f || bp.g[v][1] != 0.f || bp.g[v][2] != 0.f || bp.g[v][3] != 0.f) {
                e->last_nz_voice[v] = blk;
                e->voice_ever[v] = true;
            }
        }
        if (t > 0 && std::memcmp(&tab[t], &tab[0], sizeof(BlockParams)) != 0) all_same = false;
    }
    *out_tab = tab;
    *out_n = all_same ? 1 : T;
    return all_same ? 0 : 1;
}

const IrEntry* any_ir(const mc_engine* e) {
    for (int i = 0; i < kMaxIrs; i++)
        if (e->irs[i].d_H) return &e->irs[i];
    return nullptr;
}

// cc[i].value as the call sees it (conv.cu reads the public fields once per onProcess)
int sample_params(mc_engine* e, mc_cc_value (&cc)[2]) {
    e->last_gen = e->ph.sample(cc);
    for (int i = 0; i < 2; i++) {
        if (cc[i].select >= (uint64_t)kMaxIrs || !e->irs[cc[i].select].d_H)
            return fail(MC_ERR_STATE, "half %d selects IR %llu which is not loaded", i, (unsigned long long)cc[i].select);
    }
    if (cc[0].predelay > MC_MAX_PREDELAY) return fail(MC_ERR_ARG, "predelay %llu > %d", (unsigned long long)cc[0].predelay, MC_MAX_PREDELAY);
    return MC_OK;
}

// The voices that can still be heard at block e->t_front, as the MAC sees them
int collect_voices(const mc_engine* e, const BlockParams* first, ActiveVoice* act, int (&vir)[2][MC_MAXV], VoiceSums* vs) {
    if (vs) std::memset(vs, 0, sizeof(*vs));
    const IrEntry* fallback = any_ir(e);
    int nact = 0;
    for (int v = 0; v < MC_MAXV; v++) {
        const IrEntry* ir[2];
        for (int i = 0; i < 2; i++) {
            const int idx = e->voice[i][v].ir;
            vir[i][v] = (idx >= 0 && e->irs[idx].d_H) ? idx : -1;
            ir[i] = vir[i][v] >= 0 ? &e->irs[idx] : nullptr;
            if (ir[i] && vs) {
                for (int c = 0; c < 2; c++) {
                    vs->sig[v][i][c] = ir[i]->sums[c];
                    vs->alp[v][i][c] = ir[i]->sums[2 + c];
                }
            }
        }
        if (!ir[0] && !ir[1]) continue;
        ActiveVoice a;
        a.v = v;
        a.ir0 = ir[0] ? ir[0] : (ir[1] ? ir[1] : fallback);  // a missing half has zero gains; any valid spectra do
        a.ir1 = ir[1] ? ir[1] : a.ir0;
        a.p_end = round_up(std::max(ir[0] ? ir[0]->P : 0, ir[1] ? ir[1]->P : 0), 16);
        // sounding in this batch's windows?  (last non-zero gain not older than the sweep)
        if (!e->voice_ever[v] || e->last_nz_voice[v] + (uint64_t)a.p_end < e->t_front) continue;
        a.uniform = first && e->gain_change_block[v] + (uint64_t)a.p_end <= e->t_front;
        a.ugain = first ? make_float4(first->g[v][0], first->g[v][1], first->g[v][2], first->g[v][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        act[nact++] = a;
    }
    return nact;
}

int stage_params(mc_engine* e, int T, mc_cc_value (&cc)[2], Staged* st) {
    const int bslot = (int)(e->batch_seq % kPipe);
    BlockParams* d_ptab = e->d_ptab + (size_t)bslot * e->Tmax;

    BlockParams* tab;
    int ntab;
    // reuse of a pinned staging buffer: wait until its previous upload has run
    if (e->ptab_ev_used[e->ptab_next]) HIP_TRY(hipEventSynchronize(e->ptab_ev[e->ptab_next]));
    int berr = MC_OK;
    const uint64_t sampled_vsteps[2] = {cc[0].vsteps, cc[1].vsteps};  // (build_params counts its copy down as it goes)
    const int pstride = build_params(e, T, cc, &tab, &ntab, &berr);
    if (berr) return berr;
    for (int i = 0; i < 2; i++) {
        // vsteps counts down on the engine's copy too (conv.cu:345,353): a compare-exchange against the value this call
        // sampled, so that a select which arrived meanwhile (vsteps = speed, conv.cu:261) is not undone
        const uint64_t used = std::min<uint64_t>(sampled_vsteps[i], (uint64_t)(T / e->pm));
        e->ph.count_down(i, sampled_vsteps[i], used);
        cc[i].vsteps = sampled_vsteps[i] - used;  // (a second staging in the same call - the period parked ahead - goes on from here)
    }
    bool need_upload = true;
    if (pstride == 0 && e->uniform_valid[bslot] && std::memcmp(&e->uniform_bp[bslot], &tab[0], sizeof(BlockParams)) == 0)
        need_upload = false;
    if (need_upload) {
        HIP_TRY(hipMemcpyAsync(d_ptab, tab, sizeof(BlockParams) * (size_t)ntab, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipEventRecord(e->ptab_ev[e->ptab_next], e->stream));
        e->ptab_ev_used[e->ptab_next] = true;
        e->ptab_next = (e->ptab_next + 1) % kStageBufs;
        e->uniform_valid[bslot] = (pstride == 0);
        if (pstride == 0) e->uniform_bp[bslot] = tab[0];
    }
    st->first = tab[0];
    st->d_ptab = d_ptab;
    st->d_sums = e->d_sums + (size_t)bslot * e->Tmax;
    st->ctx.T = T;
    st->ctx.t0 = e->t_front;
    st->ctx.pstride = pstride;
    st->ctx.predelay = cc[0].predelay;
    st->ctx.slot = bslot;
    st->nact = collect_voices(e, &tab[0], st->act, st->ctx.vir, &st->ctx.vs);
    return MC_OK;
}

// Q8 pass descriptor: enabled only when some contribution is shifted past n_ref by the predelay
TailDrop make_taildrop(const mc_engine* e, const int (&vir)[2][MC_MAXV], uint64_t predelay) {
    TailDrop td;
    std::memset(&td, 0, sizeof(td));
    const IrEntry* fb = any_ir(e);
    uint64_t lmax = 0;
    td.nv = MC_MAXV;
    for (int v = 0; v < MC_MAXV; v++) {
        const IrEntry* a = vir[0][v] >= 0 ? &e->irs[vir[0][v]] : nullptr;
        const IrEntry* b = vir[1][v] >= 0 ? &e->irs[vir[1][v]] : nullptr;
        td.h0[v] = a ? a->d_h : (fb ? fb->d_h : nullptr);
        td.h1[v] = b ? b->d_h : (fb ? fb->d_h : nullptr);
        td.L0[v] = a ? (int)a->taps : 0;
        td.L1[v] = b ? (int)b->taps : 0;
        lmax = std::max<uint64_t>(lmax, std::max<uint64_t>(td.L0[v], td.L1[v]));
    }
    td.lmax = (int)lmax;
    // the reference transforms a whole call (pm blocks) at once: its contribution is taps + 256 pm - 1 frames long, and the cut applies to what
    // passes n_ref frames after the START of the call - the last block of a call loses terms (pm - 1) blocks earlier than the first
    // (tests/fuzz/fuzz_q8.py found the condition written for calls of one block: 1024-frame periods with taps + 1023 + predelay > n_ref >= taps + 255 + predelay)
    td.on = (e->cfg.compat && lmax + (uint64_t)(MC_B * e->pm - 1) + predelay > e->cfg.n_ref) ? 1 : 0;
    td.xhist = e->d_xhist;
    td.xr = e->xr;
    td.gring = e->d_gring;
    // the frequency-domain form (k_drop_fft ahead of k_post<3>): fp32 engines whose delay line still holds the blocks the dropped terms come
    // from - up to n_ref / 256 + 2 blocks before the oldest block of the batch (the ring is Tmax + that much or more for every engine that is
    // not created with a batch limit far below its fft size).  MCCONV_TD_FFT=0 (read when the engine is created): the time-domain tiles inside k_post<1> (profiles/r3_shipped_defaults.md
    // has both: 0.40 + 0.21 ms against 0.90 ms per 125 000 blocks at the shipped operating point, 2.5 against 22 ms per step at predelay 8192)
    for (int v = 0; v < MC_MAXV; v++) {
        const IrEntry* a = vir[0][v] >= 0 ? &e->irs[vir[0][v]] : nullptr;
        const IrEntry* b = vir[1][v] >= 0 ? &e->irs[vir[1][v]] : nullptr;
        td.H0s[v] = a ? a->d_H : (fb ? fb->d_H : nullptr);
        td.H1s[v] = b ? b->d_H : (fb ? fb->d_H : nullptr);
        td.P0[v] = a ? a->P : 0;
        td.P1[v] = b ? b->P : 0;
        td.Ht0[v] = a && a->tail_valid ? a->d_Htail : nullptr;
        td.Ht1[v] = b && b->tail_valid ? b->d_Htail : nullptr;
        td.tp0[v] = a && a->tail_valid ? a->tail_p0 : 0;
        td.tp1[v] = b && b->tail_valid ? b->tail_p0 : 0;
    }
    td.pstride_ir = e->Pstride;
    td.fdl = e->d_fdl;
    td.slotgain = e->d_slotgain;
    td.ring = e->ring;
    td.g_tw = e->d_tw;
    td.fft = (e->td_fft && !e->half && (uint64_t)e->ring >= (uint64_t)e->Tmax + e->cfg.n_ref / MC_B + 40) ? 1 : 0;
    return td;
}

// Q8 regime, frequency-domain form (batches): the voices' last partitions partition-major and the buffer of the cut terms, both made by
// the first batch that needs them (on the engine's stream, ahead of k_drop_fft)
int prepare_drop_fft(mc_engine* e, const int (&vir)[2][MC_MAXV], uint64_t predelay) {
    uint64_t lmax = 0;
    for (int h = 0; h < 2; h++)
        for (int v = 0; v < MC_MAXV; v++)
            if (vir[h][v] >= 0) lmax = std::max<uint64_t>(lmax, e->irs[vir[h][v]].taps);
    if (!(e->cfg.compat && lmax + (uint64_t)(MC_B * e->pm - 1) + predelay > e->cfg.n_ref) || e->half || !e->td_fft) return MC_OK;
    for (int h = 0; h < 2; h++)
        for (int v = 0; v < MC_MAXV; v++)
            if (vir[h][v] >= 0) {
                const int rc_t = ensure_htail(e, &e->irs[vir[h][v]]);
                if (rc_t != MC_OK) return rc_t;
            }
    if (!e->d_dropbuf) HIP_TRY(hipMalloc(&e->d_dropbuf, sizeof(float2) * (size_t)e->Tmax * MC_B));
    return MC_OK;
}

// Whether every output block of a batch (calls of one block) loses exactly ONE (kappa, partition) term, the same for every voice that
// loses anything - then k_fwd<true> sums the cut terms (DropAhead).  The terms of block b: kappa in {0, 1, 2 if predelay % 256},
// partitions p >= n_ref / 256 - predelay / 256 - kappa of every voice's IRs (k_drop_fft).
int plan_drop_ahead(mc_engine* e, const int (&vir)[2][MC_MAXV], uint64_t predelay, DropAhead* da, int* shift) {
    *shift = -1;
    int rc = prepare_drop_fft(e, vir, predelay);
    if (rc != MC_OK) return rc;
    const TailDrop td = make_taildrop(e, vir, predelay);
    if (!td.on || !td.fft || !e->d_dropbuf) return MC_OK;
    const int N = (int)(e->cfg.n_ref / MC_B), a = (int)(predelay >> 8), c = (int)(predelay & 255);
    int kap = -1, part = -1;
    for (int v = 0; v < td.nv; v++) {
        const int pmax = std::max(td.P0[v], td.P1[v]);
        for (int kappa = 0; kappa < (c ? 3 : 2); kappa++)
            for (int p = std::max(N - a - kappa, 0); p < pmax; p++) {
                if (kap >= 0 && (kap != kappa || part != p)) return MC_OK;  // a second term
                kap = kappa, part = p;
            }
    }
    if (kap < 0) return MC_OK;
    for (int v = 0; v < td.nv; v++) {
        da->Ht0[v] = part < td.P0[v] ? (td.Ht0[v] && part >= td.tp0[v] ? td.Ht0[v] + (size_t)(part - td.tp0[v]) * MC_NB : nullptr) : nullptr;
        da->Ht1[v] = part < td.P1[v] ? (td.Ht1[v] && part >= td.tp1[v] ? td.Ht1[v] + (size_t)(part - td.tp1[v]) * MC_NB : nullptr) : nullptr;
        if ((part < td.P0[v] && !da->Ht0[v]) || (part < td.P1[v] && !da->Ht1[v])) return MC_OK;  // (no partition-major copy: k_drop_fft reads the bank)
    }
    da->nv = td.nv;
    da->kappa = kap;
    da->c = c;
    da->shift = kap + a + part;
    da->drop = e->d_dropbuf;
    *shift = da->shift;
    return MC_OK;
}

// JACK path in the Q8 regime: the tail-drop terms of the period that starts at block blk (pm blocks), ahead of its tail kernel
// on the engine's stream (they depend on blocks at least n_ref frames old only).  Returns the buffer the tail reads, or null.
const float* launch_drop_period(mc_engine* e, const TailDrop& td, uint64_t blk, uint64_t predelay) {
    if (!td.on) return nullptr;
    float* dst = e->d_drop[(blk / (uint64_t)e->pm) & 1];
    const int64_t blo = (int64_t)e->epoch_b0;
    if (td.fft) {  // in the frequency domain: a partition sum over the last partitions and one inverse transform per block (k_drop_period_fft)
        if (e->pm == 1)
            hipLaunchKernelGGL(k_drop_period_fft<1>, dim3(1), dim3(64), 0, e->stream, td, dst, (int64_t)blk, (int64_t)predelay, (int64_t)e->cfg.n_ref, blo);
        else if (e->pm == 2)
            hipLaunchKernelGGL(k_drop_period_fft<2>, dim3(1), dim3(128), 0, e->stream, td, dst, (int64_t)blk, (int64_t)predelay, (int64_t)e->cfg.n_ref, blo);
        else
            hipLaunchKernelGGL(k_drop_period_fft<4>, dim3(1), dim3(256), 0, e->stream, td, dst, (int64_t)blk, (int64_t)predelay, (int64_t)e->cfg.n_ref, blo);
        return dst;
    }
    if (e->pm == 1)
        hipLaunchKernelGGL(k_drop_period<1>, dim3(1), dim3(256), 0, e->stream, td, dst, (int64_t)blk, (int64_t)predelay, (int64_t)e->cfg.n_ref, e->rc, blo);
    else if (e->pm == 2)
        hipLaunchKernelGGL(k_drop_period<2>, dim3(1), dim3(256), 0, e->stream, td, dst, (int64_t)blk, (int64_t)predelay, (int64_t)e->cfg.n_ref, e->rc, blo);
    else
        hipLaunchKernelGGL(k_drop_period<4>, dim3(1), dim3(256), 0, e->stream, td, dst, (int64_t)blk, (int64_t)predelay, (int64_t)e->cfg.n_ref, e->rc, blo);
    return dst;
}

// shard of a voice's partition range [p_begin, p_end), multiples of 16
void partition_range(const mc_engine* e, int p_hi, int* p_begin, int* p_end) {
    int pb = (int)e->cfg.part_begin, pe = e->cfg.part_end ? std::min<int>((int)e->cfg.part_end, p_hi) : p_hi;
    if (pb > pe) pb = pe;
    *p_begin = pb;
    *p_end = pe;
}

void launch_mac_stream(mc_engine* e, const ActiveVoice& a, int p_lo, int p_hi, int T, int slot0, int nsum, int ch_off,
                       float4* dst = nullptr, unsigned long long* stamps = nullptr) {
    if (!dst) dst = e->d_part;
    const hipStream_t st = e->stream;
    const int nt = e->stream_nt;
    const int span = p_hi - p_lo;
    const int chunk = round_up(std::max(1, (span + e->nchunk - 1) / e->nchunk), 64);
    const dim3 grid(MC_NB, e->nchunk, T);
    const float4* sg = e->d_slotgain + (size_t)a.v * e->ring;
    const bool half = e->half && a.ir0->d_H16 && a.ir1->d_H16;
    const void* h0 = half ? (const void*)a.ir0->d_H16 : (const void*)a.ir0->d_H;
    const void* h1 = half ? (const void*)a.ir1->d_H16 : (const void*)a.ir1->d_H;
    const void* fd = half ? (const void*)e->d_fdl16 : (const void*)e->d_fdl;
    const float2 inv = make_float2(1.0f / (a.ir0->scale16 * FDL16_SCALE), 1.0f / (a.ir1->scale16 * FDL16_SCALE));
#define MC_LAUNCH_STREAM(U, NT, H)                                                                                        \
    hipLaunchKernelGGL((k_mac_stream<U, NT, H>), grid, dim3(NT), 0, st, h0, h1, e->Pstride, p_lo, p_hi, chunk, fd, sg, \
                       e->ring, slot0, dst, nsum, ch_off, a.ugain, inv, stamps)
#define MC_LAUNCH_STREAM_H(U, NT) \
    do {                          \
        if (half)                 \
            MC_LAUNCH_STREAM(U, NT, true); \
        else                      \
            MC_LAUNCH_STREAM(U, NT, false); \
    } while (0)
    if (nt == 512) {
        if (a.uniform) MC_LAUNCH_STREAM_H(true, 512); else MC_LAUNCH_STREAM_H(false, 512);
    } else {
        if (a.uniform) MC_LAUNCH_STREAM_H(true, 256); else MC_LAUNCH_STREAM_H(false, 256);
    }
#undef MC_LAUNCH_STREAM_H
#undef MC_LAUNCH_STREAM
}

// Where k_inv finds the partition sums of a batch
struct MacOut {
    const float4* ysrc;
    int64_t sk, stt, sc;
    int nsum;
    int swept;
    bool resident;
    int lvl;            // fast-FIR levels of the main part (0 = direct form)
    int64_t ffa_plane;  // elements between the component sequences (k_inv combines them)
    int main_n;         // blocks described by the fields above; the remaining tail_n blocks of the batch come from
    int tail_n;         // the streaming kernel:
    const float4* tail_ysrc;
    int64_t tail_sk, tail_stt;
    int tail_nsum;
    bool corr_done;  // the Q1/Q2 prefix sums rode along with the launch (k_g2_mac) and are final when it ends
};

// inverse transforms of the blocks whose partition sums `mo` describes, into the segment ring from block `b0` -
// or, to_wet, overlap-added straight into the wet ring (only the last block of a launch then stays in the segment ring)
// out != null: the launches also finish the output of their blocks (k_inv_wet<true>, see OutArgs)
void launch_inv(mc_engine* e, const MacOut& mo, uint64_t b0, hipStream_t st, bool to_wet = false, const OutArgs* out = nullptr) {
    auto inv = [&](const float4* y, int64_t sk, int64_t stt, int nsum, int64_t sc, int n, uint64_t b) {
        const int seg0 = (int)(b & (uint64_t)(e->sr - 1));
        if (to_wet) {
            OutArgs oa;
            std::memset(&oa, 0, sizeof(oa));
            const dim3 grid((n + IW_NEW - 1) / IW_NEW);
            if (out) {
                oa = *out;
                oa.blk0 = (int)((int64_t)b - out->tabs0);
                hipLaunchKernelGGL(k_inv_wet<true>, grid, dim3(IW_THREADS), 0, st, y, sk, stt, nsum, sc, n, e->d_seg, e->sr, seg0, e->d_wet,
                                   e->wr, (int64_t)b * MC_B, e->d_tw, oa);
            } else
                hipLaunchKernelGGL(k_inv_wet<false>, grid, dim3(IW_THREADS), 0, st, y, sk, stt, nsum, sc, n, e->d_seg, e->sr, seg0, e->d_wet,
                                   e->wr, (int64_t)b * MC_B, e->d_tw, oa);
        } else
            hipLaunchKernelGGL(k_inv, dim3((n + FWD_TILE - 1) / FWD_TILE), dim3(XF_THREADS), 0, st, y, sk, stt, nsum, sc, n, e->d_seg, e->sr,
                               seg0, e->d_tw);
    };
    if (mo.main_n > 0 && mo.lvl > 0) {
        const int S = 1 << mo.lvl;
        const dim3 cgrid(((mo.main_n + S - 1) / S + 255) / 256, MC_NB);
        if (mo.lvl == 1)
            hipLaunchKernelGGL(k_ffa_combine<1>, cgrid, dim3(256), 0, st, mo.ysrc, mo.ffa_plane, (int)mo.sk, mo.main_n, e->d_Yc, e->Tcap);
        else if (mo.lvl == 2)
            hipLaunchKernelGGL(k_ffa_combine<2>, cgrid, dim3(256), 0, st, mo.ysrc, mo.ffa_plane, (int)mo.sk, mo.main_n, e->d_Yc, e->Tcap);
        else
            hipLaunchKernelGGL(k_ffa_combine<3>, cgrid, dim3(256), 0, st, mo.ysrc, mo.ffa_plane, (int)mo.sk, mo.main_n, e->d_Yc, e->Tcap);
        inv(e->d_Yc, (int64_t)e->Tcap, (int64_t)1, 1, (int64_t)0, mo.main_n, b0);
    } else if (mo.main_n > 0) {
        inv(mo.ysrc, mo.sk, mo.stt, mo.nsum, mo.sc, mo.main_n, b0);
    }
    if (mo.tail_n > 0) inv(mo.tail_ysrc, mo.tail_sk, mo.tail_stt, mo.tail_nsum, (int64_t)1, mo.tail_n, b0 + (uint64_t)mo.main_n);
}

// Taps of the convolution along the block axis that this engine sums: the longest sounding IR's partitions, or the
// engine's shard of them - a shard [pb, pe) is a (pe - pb)-tap convolution whose window starts pb slots earlier.
int block_axis_taps(const mc_engine* e, const ActiveVoice* act, int nact, int* pb_out) {
    int taps = 0, pb0 = (int)e->cfg.part_begin;
    for (int a = 0; a < nact; a++) {
        int pb, pe;
        partition_range(e, act[a].p_end, &pb, &pe);
        taps = std::max(taps, pe - pb);
    }
    if (pb_out) *pb_out = pb0;
    return taps;
}

// Will launch_mac_batch take the second-level transform for this batch?
bool fft2_applies(const mc_engine* e, const ActiveVoice* act, int nact, bool per_slot_gains, int T) {
    // both gain layouts are handled (uniform: 2 sequences per bin, fused form; per slot: 4 per voice, split form)
    bool per_slot = per_slot_gains;
    for (int a = 0; a < nact; a++) per_slot = per_slot || !act[a].uniform;
    if (!(T >= e->stream_threshold && !e->half) || !e->fft2 || nact <= 0) return false;
    const int taps = block_axis_taps(e, act, nact, nullptr);
    if (taps > F2_N / 2) return false;
    const bool shard = e->cfg.part_begin || e->cfg.part_end;
    if (!per_slot && e->fft2_fused && taps <= e->g2_pmax)
        // Uniform gains, fused form: a launch costs ~39 us per chunk of 8192 - taps + 1 blocks whatever the taps; the direct
        // MAC ~10 us + 0.078 ns per block and partition (and never less than ~40 us for the longest IRs: one workgroup
        // sweeps its partitions in turn).  Measured (P = 1728 / 345 / 32): the transform wins from 128 / ~900 / ~9000
        // blocks on - a product of ~300 000 (MCCONV_FFT2_WORK)
        return taps >= (shard ? 16 : e->g2_pmin) && (int64_t)T * taps >= e->fft2_work;
    // per-slot gains (or the fused form switched off): the split form, two launches and a stash
    if (T < 768) return false;
    if (shard) return taps >= 16 && (int64_t)T * taps >= 1300000;
    return taps >= 256;
}

// Partition x bin MAC of T blocks starting at delay-line slot `slot0` for the given voices.
// per_slot_gains: the batch's blocks (or the window) do not share one set of gains.
// ride != null: the Q1/Q2 prefix sums may ride along with the launch (mo->corr_done says whether they did)
int launch_mac_batch(mc_engine* e, const ActiveVoice* act, int nact, bool per_slot_gains, int T, int slot0, MacOut* mo,
                     const CorrArgs* ride = nullptr) {
    mo->corr_done = false;
    mo->resident = T >= e->stream_threshold && !e->half;
    mo->swept = 0;
    mo->sc = 1;
    mo->lvl = 0;
    mo->ffa_plane = 0;
    mo->main_n = T;
    mo->tail_n = 0;
    // Fast-FIR form of the convolution along the block axis.  With the polyphase components Xe[n] = X[2n],
    // Xo[n] = X[2n+1], He[q] = H[2q], Ho[q] = H[2q+1] (indices relative to the batch start):
    //   Y[2n] = (He*Xe)[n] + (Ho*Xo)[n-1],   Y[2n+1] = ((He+Ho)*(Xe+Xo))[n] - (He*Xe)[n] - (Ho*Xo)[n]
    // - three convolutions with half the taps and half the outputs: 3/4 of the multiply-adds; applied to each of
    // the three again (level 2): nine convolutions at a quarter of the rate, 9/16.  One launch of the resident
    // kernel covers all components; k_inv combines them.  Every component is computed for sequence indices
    // -1 .. T/2^L - 2, which covers the first T - 2^L blocks of the batch; the last 2^L blocks go through the
    // streaming kernel, so that no tile is spent on one extra output.
    // Long batches whose window carries one set of gains: the convolution along the block axis as a circular
    // convolution per (bin, chunk of blocks) with a second-level transform of length F2_N (k_f2_fwd, k_f2_prod); the IRs'
    // partition sequences are transformed once (k_fft2_ir).  O(log) instead of O(P) work per output block.
    if (fft2_applies(e, act, nact, per_slot_gains, T)) {
        int pb0 = 0;
        const int pmax = block_axis_taps(e, act, nact, &pb0);  // taps of this engine's (shard of the) convolution
        slot0 = (slot0 - pb0) & (e->ring - 1);                  // a shard's window starts pb slots earlier
        {
            Fft2Voices vv;
            std::memset(&vv, 0, sizeof(vv));
            for (int a = 0; a < nact; a++) {
                int rc = ensure_fft2(e, act[a].ir0);
                if (!rc) rc = ensure_fft2(e, act[a].ir1);
                if (rc) return rc;
                vv.h0[a] = act[a].ir0->d_H2;
                vv.h1[a] = act[a].ir1->d_H2;
                vv.g[a] = act[a].ugain;
            }
            vv.n = nact;
            // one set of gains over the whole window: transform the two inputs and weight the products; otherwise
            // transform gain(slot) x input for every voice and path
            bool per_slot = per_slot_gains;
            for (int a = 0; a < nact; a++) per_slot = per_slot || !act[a].uniform;
            if (!per_slot && e->fft2_fused && pmax <= e->g2_pmax) {
                // fused form: both inputs' 8192-point spectra side by side in LDS, no stash (k_g2_mac)
                for (int a = 0; a < nact; a++) {
                    int rc = ensure_g2(e, act[a].ir0);
                    if (!rc) rc = ensure_g2(e, act[a].ir1);
                    if (rc) return rc;
                    vv.h0[a] = act[a].ir0->d_G2;
                    vv.h1[a] = act[a].ir1->d_G2;
                }
                const int chunk_t = G2_N - pmax + 1;
                const int nch = (T + chunk_t - 1) / chunk_t;
                if (e->debug_addr) {
                    // every address the kernel forms lies in one of these ranges (see the bounds argument at k_g2_mac)
                    e->debug_addr = false;
                    fprintf(stderr, "mcconv k_g2_mac: T %d chunk_t %d taps %d items %d grid %d\n  fdl  [%p, %p)\n  Yc   [%p, %p) (written: %d of %d entries per bin)\n",
                            T, chunk_t, pmax, MC_NB * nch, std::min(MC_NB * nch, e->g2_grid), (void*)e->d_fdl, (void*)(e->d_fdl + (size_t)MC_NB * e->ring),
                            (void*)e->d_Yc, (void*)(e->d_Yc + (size_t)MC_NB * e->Tcap), T, e->Tcap);
                    for (int a = 0; a < nact; a++)
                        fprintf(stderr, "  G2[%d] in1 [%p, %p) in2 [%p, %p)\n", a, (void*)vv.h0[a], (void*)(vv.h0[a] + (size_t)2 * 257 * G2_N),
                                (void*)vv.h1[a], (void*)(vv.h1[a] + (size_t)2 * 257 * G2_N));
                }
#ifdef MCCONV_LAB  // the measured alternatives of k_g2_mac (lab_kernels.hip.h)
                if (e->g2_duo && !e->g2_wide && nch >= e->g2_duo_minch && e->g2_grid == (1 << 30)) {
                    // the lockstep form: persistent, two workers (the halves of a workgroup) per CU; the Q1/Q2 terms do not ride
                    // along (a rider would need a CU of its own: the launches of their own follow, as for every long batch)
                    const int grid = std::max(8, std::min(e->g2_duo_grid, 256) & ~7);
                    static const int solo = LAB_ENV("MCCONV_G2_DUO_SOLO") ? std::atoi(LAB_ENV("MCCONV_G2_DUO_SOLO")) : 0;  // (measurement: one group works alone)
                    const int wpl = (grid >> 3) * (solo ? 1 : 2), nxq = (MC_NB * nch) >> 3;
                    hipLaunchKernelGGL(k_g2_duo, dim3(grid), dim3(G2D_THREADS), 0, e->stream, e->d_fdl, e->ring, slot0, T, chunk_t, pmax, vv,
                                       e->d_Yc, e->Tcap, MC_NB * nch, (nxq + wpl - 1) / wpl, solo);
                } else if (e->g2_wide)  // the one-workgroup-per-CU form (MCCONV_G2_WIDE=1)
                    hipLaunchKernelGGL(k_g2_mac_wide, dim3(std::min(MC_NB * nch, e->g2_grid)), dim3(G2_THREADS), 0, e->stream, e->d_fdl,
                                       e->ring, slot0, T, chunk_t, pmax, vv, e->d_Yc, e->Tcap, MC_NB * nch);
                else
#endif
                {
                    CorrArgs ca;
                    std::memset(&ca, 0, sizeof(ca));
                    if (ride && ride->nchunks > 0) {
                        ca = *ride;
                        mo->corr_done = true;
                    }
                    const int main_grid = std::min(MC_NB * nch, e->g2_grid);
                    static const int dyn_lds = LAB_ENV("MCCONV_G2_DYNLDS") ? std::atoi(LAB_ENV("MCCONV_G2_DYNLDS")) : 0;  // (measurement: extra LDS per workgroup, > 22 KB leaves one workgroup per CU)
                    hipLaunchKernelGGL(k_g2_mac, dim3(main_grid + ca.nchunks), dim3(G2B_THREADS), dyn_lds, e->stream, e->d_fdl, e->ring, slot0, T,
                                       chunk_t, pmax, vv, e->d_Yc, e->Tcap, MC_NB * nch, ca, main_grid);
                }
                mo->ysrc = e->d_Yc;
                mo->sk = e->Tcap;
                mo->stt = 1;
                mo->nsum = 1;
                mo->sc = 0;
                mo->swept = pmax;
                mo->lvl = -2;  // second-level transform, fused form
                e->n_mac_form[0]++;
                return MC_OK;
            }
            const int nseq = per_slot ? 4 * nact : 2;
            Fft2Gains gg;
            std::memset(&gg, 0, sizeof(gg));
            if (per_slot)
                for (int a = 0; a < nact; a++) gg.row[a] = e->d_slotgain + (size_t)act[a].v * e->ring;
            const int chunk_t = F2_N - pmax + 1;
            const dim3 grid(MC_NB, (T + chunk_t - 1) / chunk_t);
            const size_t need = (size_t)grid.y * nseq;
            if (e->stash_chunks < need) {
                HIP_TRY(hipStreamSynchronize(e->stream));
                if (e->d_stash) (void)hipFree(e->d_stash);
                e->d_stash = nullptr;
                HIP_TRY(hipMalloc(&e->d_stash, sizeof(float2) * need * MC_NB * F2_N));
                e->stash_chunks = need;
            }
            hipLaunchKernelGGL(k_f2_fwd, dim3(grid.x * grid.y * nseq), dim3(F2_THREADS), 0, e->stream, e->d_fdl, e->ring, slot0, T,
                               chunk_t, pmax, e->d_stash, nseq, gg);
            hipLaunchKernelGGL(k_f2_prod, dim3(grid.x * grid.y * 2), dim3(F2_THREADS), 0, e->stream, e->d_stash, T, chunk_t, pmax, vv,
                               e->d_Yc, e->Tcap, nseq);
            mo->ysrc = e->d_Yc;
            mo->sk = e->Tcap;
            mo->stt = 1;
            mo->nsum = 1;
            mo->sc = 0;
            mo->swept = pmax;
            mo->lvl = -1;  // marks the second-level transform in the kernel statistics
            e->n_mac_form[1]++;
            return MC_OK;
        }
    }
    int lvl = 0;
    if (mo->resident && e->ffa_levels > 0 && e->cfg.part_begin == 0 && e->cfg.part_end == 0 && nact > 0) {
        int pmin = 1 << 30;
        const bool have = hp_possible(e, 1);
        for (int a = 0; a < nact; a++) pmin = std::min(pmin, act[a].p_end);
        auto ready = [&](int l) { return hp_possible(e, l); };
        if (have && e->ffa_levels >= 3 && T >= 8192 && T % 8 == 0 && pmin >= 1024 && ready(3)) lvl = 3;
        else if (have && e->ffa_levels >= 2 && T >= 4096 && T % 4 == 0 && pmin >= 512 && ready(2)) lvl = 2;
        else if (have && T >= 2048 && T % 2 == 0 && pmin >= 256) lvl = 1;
    }
    if (lvl) {
        for (int a = 0; a < nact; a++) {  // the components of this level, built on the first batch that takes it
            int rc = ensure_hp(e, act[a].ir0, lvl);
            if (!rc) rc = ensure_hp(e, act[a].ir1, lvl);
            if (rc) return rc;
        }
        const int S = 1 << lvl, ncomp = lvl == 1 ? 3 : (lvl == 2 ? 9 : 27);
        const int nh = T / S, ph = e->Pstride >> lvl;
        const int tiles = (nh + 255) / 256;
        const int tcap = tiles * 256;
        const size_t plane = (size_t)MC_NB * tcap;
        int launched = 0;
        for (int a = 0; a < nact; a++) {
            const ActiveVoice& av = act[a];
            const int q_end = round_up((av.p_end + S - 1) / S, 16);
            const float4* sg = e->d_slotgain + (size_t)av.v * e->ring;
            const dim3 grid(MC_NB * tiles * ncomp);
            if (av.uniform && !per_slot_gains)
                hipLaunchKernelGGL(k_mac_resident<false>, grid, dim3(256), 0, e->stream, av.ir0->d_Hp[lvl - 1], av.ir1->d_Hp[lvl - 1],
                                   ph, 0, q_end, e->d_fdl, e->ring, slot0, nh, av.ugain, sg, e->d_Y, tcap, launched ? 1 : 0, 1, q_end,
                                   lvl, tiles, (int64_t)plane);
            else
                hipLaunchKernelGGL(k_mac_resident<true>, grid, dim3(256), 0, e->stream, av.ir0->d_Hp[lvl - 1], av.ir1->d_Hp[lvl - 1],
                                   ph, 0, q_end, e->d_fdl, e->ring, slot0, nh, av.ugain, sg, e->d_Y, tcap, launched ? 1 : 0, 1, q_end,
                                   lvl, tiles, (int64_t)plane);
            launched++;
            mo->swept = std::max(mo->swept, av.p_end);
        }
        mo->ysrc = e->d_Y;
        mo->sk = tcap;
        mo->stt = 1;
        mo->nsum = 1;
        mo->sc = 0;
        mo->lvl = lvl;
        e->n_mac_form[2]++;
        mo->ffa_plane = (int64_t)plane;
        mo->main_n = T - S;
        // the last S blocks: streaming kernel, one set of chunk partials per voice
        mo->tail_n = S;
        mo->tail_nsum = nact * e->nchunk;
        for (int a = 0; a < nact; a++) {
            ActiveVoice av = act[a];
            if (per_slot_gains) av.uniform = false;
            launch_mac_stream(e, av, 0, av.p_end, S, (slot0 + T - S) & (e->ring - 1), mo->tail_nsum, a * e->nchunk, e->d_tail);
        }
        mo->tail_ysrc = e->d_tail;
        mo->tail_sk = mo->tail_nsum;
        mo->tail_stt = (int64_t)MC_NB * mo->tail_nsum;
        return MC_OK;
    }
    if (mo->resident) {
        e->n_mac_form[2]++;
        // short batches: split the partition range so that the launch has ~2048 workgroups
        const int tiles = (T + 255) / 256;
        const int psplit = std::max(1, std::min(8, 8 / tiles));
        const int tcap = psplit > 1 ? tiles * 256 : e->Tcap;  // plane stride of Y (planes summed by k_inv)
        int launched = 0;
        for (int a = 0; a < nact; a++) {
            const ActiveVoice& av = act[a];
            int p_begin, p_end;
            partition_range(e, av.p_end, &p_begin, &p_end);
            if (p_end <= p_begin) continue;
            const int pchunk = round_up((p_end - p_begin + psplit - 1) / psplit, 16);
            const dim3 grid(MC_NB * tiles * psplit);
            const float4* sg = e->d_slotgain + (size_t)av.v * e->ring;
            if (av.uniform && !per_slot_gains)
                hipLaunchKernelGGL(k_mac_resident<false>, grid, dim3(256), 0, e->stream, av.ir0->d_H, av.ir1->d_H, e->Pstride,
                                   p_begin, p_end, e->d_fdl, e->ring, slot0, T, av.ugain, sg, e->d_Y, tcap, launched ? 1 : 0,
                                   psplit, pchunk, 0, tiles, (int64_t)0);
            else
                hipLaunchKernelGGL(k_mac_resident<true>, grid, dim3(256), 0, e->stream, av.ir0->d_H, av.ir1->d_H, e->Pstride,
                                   p_begin, p_end, e->d_fdl, e->ring, slot0, T, av.ugain, sg, e->d_Y, tcap, launched ? 1 : 0,
                                   psplit, pchunk, 0, tiles, (int64_t)0);
            launched++;
            mo->swept = std::max(mo->swept, p_end - p_begin);
        }
        if (!launched) HIP_TRY(hipMemsetAsync(e->d_Y, 0, sizeof(float4) * (size_t)MC_NB * tcap * psplit, e->stream));
        mo->ysrc = e->d_Y;
        mo->sk = tcap;
        mo->stt = 1;
        mo->nsum = psplit;
        mo->sc = (int64_t)MC_NB * tcap;
    } else {
        // streaming: one set of chunk partials per sounding voice, all added by k_inv
        int nv = 0;
        ActiveVoice list[MC_MAXV];
        int pb[MC_MAXV], pe[MC_MAXV];
        for (int a = 0; a < nact; a++) {
            partition_range(e, act[a].p_end, &pb[nv], &pe[nv]);
            if (pe[nv] <= pb[nv]) continue;
            list[nv] = act[a];
            if (per_slot_gains) list[nv].uniform = false;
            nv++;
        }
        mo->nsum = std::max(1, nv) * e->nchunk;
        if (!nv) HIP_TRY(hipMemsetAsync(e->d_part, 0, sizeof(float4) * (size_t)T * MC_NB * mo->nsum, e->stream));
        for (int a = 0; a < nv; a++) {
            launch_mac_stream(e, list[a], pb[a], pe[a], T, slot0, mo->nsum, a * e->nchunk);
            mo->swept = std::max(mo->swept, pe[a] - pb[a]);
        }
        mo->ysrc = e->d_part;
        mo->sk = mo->nsum;
        mo->stt = (int64_t)MC_NB * mo->nsum;
    }
    return MC_OK;
}

Retired make_retired(const mc_engine* e) {
    Retired r;
    r.mac = e->d_res_mac;
    r.fix = e->d_res_fix;
    r.rr = e->rr;
    r.end = (int64_t)e->res_end;
    r.b0 = (int64_t)e->epoch_b0;
    return r;
}

int zero_ring_range(mc_engine* e, float* ring, uint64_t from, uint64_t to) {
    // samples [from, to) of a [2][rr] ring indexed by absolute sample; to - from <= rr
    if (to <= from) return MC_OK;
    const uint64_t rr = (uint64_t)e->rr;
    const uint64_t a = from & (rr - 1), n = to - from;
    const uint64_t n1 = std::min(n, rr - a);
    for (int c = 0; c < 2; c++) {
        HIP_TRY(hipMemsetAsync(ring + (size_t)c * rr + a, 0, sizeof(float) * n1, e->stream));
        if (n > n1) HIP_TRY(hipMemsetAsync(ring + (size_t)c * rr, 0, sizeof(float) * (n - n1), e->stream));
    }
    return MC_OK;
}

// The predelay of the next call differs from the live epoch's.  The reference shifts every call's contribution
// by the predelay current at that call (conv.cu:411-415), so the blocks played so far keep their old offset:
// render what they still owe (partition sums over silent input, Q1/Q2 window terms, Q8 drops) into the
// residual rings and restart the live pipeline from silence under the new predelay.
int retire_epoch(mc_engine* e, uint64_t new_delay, bool force = false) {
    if (new_delay == e->cur_delay && !force) return MC_OK;
    if (e->t_front == e->epoch_b0) {  // nothing has been played under the old value
        e->cur_delay = new_delay;
        return MC_OK;
    }
    if (e->pipe_count) return fail(MC_ERR_STATE, "predelay change / voice merge while a batch awaits mc_finish_batch_device");
    if (e->sliced) return fail(MC_ERR_STATE, "predelay change / voice merge on a block-sliced engine (mc_reset first)");
    {
        int rc = drain_post(e);  // the rings below are rewritten
        if (!rc) rc = leave_jack_path(e);
        if (rc) return rc;
    }
    const uint64_t b0 = e->t_front, bs = e->epoch_b0, d_old = e->cur_delay;
    // the latest old call started at block b0 - pm; the reference cuts its contribution n_ref samples later
    const uint64_t new_end = (b0 - (uint64_t)e->pm) * MC_B + e->cfg.n_ref;
    const uint64_t pending_from = std::max<uint64_t>(e->res_end, b0 * MC_B);
    int rc = zero_ring_range(e, e->d_res_mac, pending_from, new_end);
    if (!rc) rc = zero_ring_range(e, e->d_res_fix, pending_from, new_end);
    if (rc) return rc;

    ActiveVoice act[MC_MAXV];
    int vir[2][MC_MAXV];
    const int nact = collect_voices(e, nullptr, act, vir, nullptr);
    int F = 0;  // blocks over which the old blocks still ring: the longest sounding IR
    for (int a = 0; a < nact; a++) F = std::max(F, act[a].p_end);
    // nothing beyond the cut is kept
    F = (int)std::min<uint64_t>((uint64_t)F, (new_end - b0 * MC_B + MC_B - 1) / MC_B);
    if (d_old)
        hipLaunchKernelGGL(k_flush_ring, dim3((unsigned)((d_old + 255) / 256)), dim3(256), 0, e->stream, e->d_wet, e->wr,
                           (int64_t)(b0 * MC_B), (int64_t)d_old, e->d_res_mac, e->rr, (int64_t)new_end);
    uint64_t tv = b0;
    while (F > 0) {
        int Tc = std::min(F, e->Tcap);
        if (!(Tc >= e->stream_threshold && !e->half)) Tc = std::min(Tc, e->Tstream);
        const int slot0 = (int)(tv & (uint64_t)(e->ring - 1));
        // silent blocks into the delay line (n_frames = 0: the inputs are never read)
        hipLaunchKernelGGL(k_fwd<false>, dim3((Tc + FWD_TILE - 1) / FWD_TILE), dim3(XF_THREADS), 0, e->stream, (const float*)nullptr,
                           (const float*)nullptr, 1, (int64_t)0, Tc, e->d_fdl, e->ring, slot0, (const BlockParams*)nullptr, 0,
                           (float4*)nullptr, (float4*)nullptr, e->d_tw, e->d_fdl16, (float*)nullptr, 0, (float4*)nullptr, 0,
                           (int64_t)0, 0, Tc, Tc, 0, 0, DropAhead(), 0);
        MacOut mo;
        rc = launch_mac_batch(e, act, nact, true, Tc, slot0, &mo);
        if (rc) return rc;
        launch_inv(e, mo, tv, e->stream);
        hipLaunchKernelGGL(k_flush_ola, dim3(Tc), dim3(256), 0, e->stream, e->d_seg, e->sr, (int64_t)tv, (int64_t)d_old,
                           e->d_res_mac, e->rr, (int64_t)new_end);
        tv += (uint64_t)Tc;
        F -= Tc;
    }
    {
        const uint64_t n = new_end - b0 * MC_B;
        hipLaunchKernelGGL(k_flush_fix, dim3((unsigned)((n + MC_B - 1) / MC_B)), dim3(256), 0, e->stream, e->d_cring, e->rc,
                           e->d_res_fix, e->rr, (int64_t)(b0 * MC_B), (int64_t)new_end, (int64_t)bs, (int64_t)b0 - 1, (int64_t)d_old,
                           (int64_t)e->cfg.n_ref, (int)e->cfg.compat, make_taildrop(e, vir, d_old), e->pm);
    }
    HIP_TRY(hipGetLastError());
    // the live pipeline restarts from silence: the old blocks are accounted for
    HIP_TRY(hipMemsetAsync(e->d_fdl, 0, sizeof(float4) * (size_t)MC_NB * e->ring, e->stream));
    if (e->d_fdl16) HIP_TRY(hipMemsetAsync(e->d_fdl16, 0, sizeof(uint2) * (size_t)MC_NB * e->ring, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_seg, 0, sizeof(float) * (size_t)e->sr * 2 * FFT_N, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_wet, 0, sizeof(float) * 2 * (size_t)e->wr, e->stream));
    e->res_end = new_end;
    e->epoch_b0 = b0;
    e->cur_delay = new_delay;
    e->spec_valid = e->dspec.valid = false;
    return MC_OK;
}

// The chunks of 256 blocks the Q1/Q2 prefix sums exist for (CorrArgs): the runs [need_a0, need_a1) and [need_b0, T) of a
// block-sliced rank, everything for a whole batch.  The last block of the batch is always covered (the next call's base).
void corr_chunks(CorrArgs* ca, int T, int need_a0, int need_a1, int need_b0) {
    ca->need_a0 = need_a0;
    ca->need_a1 = need_a1;
    ca->need_b0 = need_b0;
    ca->run0 = need_a0 & ~(CORR_CHUNK - 1);
    ca->nrun0 = need_a1 > ca->run0 ? (std::min(need_a1, T) - ca->run0 + CORR_CHUNK - 1) / CORR_CHUNK : 0;
    const int end0 = ca->run0 + ca->nrun0 * CORR_CHUNK;
    ca->run1 = std::max(std::min(need_b0, T - 1) & ~(CORR_CHUNK - 1), end0);
    const int nrun1 = T > ca->run1 ? (T - ca->run1 + CORR_CHUNK - 1) / CORR_CHUNK : 0;
    ca->nchunks = ca->nrun0 + nrun1;
}

// ---------------------------------------------------------------------------
// Long settled batches as overlap-save segments (ossave.hip.h).
// ---------------------------------------------------------------------------
// Can this whole batch take the form?  One engine finishing its own output, one set of gains over the batch and the
// window before it, no retired predelay epoch ringing out, the segments' history inside the live epoch; in the Q8 regime only
// the shipped shape (every output block loses ONE term of ONE source block: the forward transforms sum the cut terms, q8_ok).
bool os_applies(const mc_engine* e, const Staged& st, int count, bool slice, const float* d_in1, const float* d_in2, const float* d_outL,
                const float* d_outR, bool q8_ok, int* ovl_blocks) {
    if (!e->os_on || e->os_hold || e->pipelined || !e->fuse_out || !e->inv_to_wet || (e->sliced && !slice)) return false;
    if (e->cfg.part_begin || e->cfg.part_end || !d_outL || !d_outR || count < e->os_min_blocks || st.ctx.pstride != 0 || st.nact <= 0) return false;
    // (mc_config.stream_threshold asks for the literal MAC below it; engines with fp16 storage keep it for the partition sweep of
    // single periods and short batches - the spectra of this form come from the fp32 taps whatever the storage of the sweep)
    if (!e->half && count < e->stream_threshold) return false;
    if ((reinterpret_cast<uintptr_t>(d_in1) | reinterpret_cast<uintptr_t>(d_in2) | reinterpret_cast<uintptr_t>(d_outL) | reinterpret_cast<uintptr_t>(d_outR)) & 15) return false;
    int pmax = 0;
    for (int a = 0; a < st.nact; a++) {
        if (!st.act[a].uniform) return false;
        pmax = std::max(pmax, st.act[a].p_end);
    }
    // a block-sliced engine's first segment also carries the blocks whose Q1/Q2 terms its windows reach (their sums come from the column pass)
    const int ovl = slice ? std::max(pmax, (int)(e->cfg.n_ref / MC_B) + MC_MAX_PREDELAY / MC_B + 2) : pmax;
    if (pmax <= 0 || (int64_t)ovl * MC_B > OS_N / 2) return false;  // (a segment at least half new frames)
    if (e->res_end > e->t_front * MC_B) return false;
    if (e->epoch_b0 != 0 && e->epoch_b0 + (uint64_t)pmax > e->t_front) return false;
    if (!q8_ok) return false;  // (Q8 regime in a shape whose cut terms the forward transforms cannot sum themselves)
    *ovl_blocks = ovl;
    return true;
}

// buffers for nseg segments and the spectra of the batch's (IR set, gains)
int ensure_os(mc_engine* e, const Staged& st, int nseg) {
    const size_t seg_el = (size_t)OS_ITEMS * OS_N2;
    const size_t want = (size_t)std::max(nseg, 2);  // (the spectra are built in two segments' worth of it)
    if (e->os_T_segs < want) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->d_os_T) (void)hipFree(e->d_os_T);
        e->d_os_T = nullptr;
        e->os_T_segs = 0;
        HIP_TRY(hipMalloc(&e->d_os_T, sizeof(float4) * seg_el * want));
        e->os_T_segs = want;
    }
    if (e->os_part_segs < (size_t)nseg) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->d_os_part) (void)hipFree(e->d_os_part);
        e->d_os_part = nullptr;
        e->os_part_segs = 0;
        HIP_TRY(hipMalloc(&e->d_os_part, sizeof(float4) * (size_t)OS_TPS * OS_N1 * (size_t)nseg));
        e->os_part_segs = (size_t)nseg;
    }
    if (!e->os_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&e->os_stream, hipStreamNonBlocking));
        for (int i = 0; i < 3; i++) HIP_TRY(hipEventCreateWithFlags(&e->os_ev[i], hipEventDisableTiming));
    }
    if (!e->d_os_SP) HIP_TRY(hipMalloc(&e->d_os_SP, sizeof(float4) * seg_el * 2));
    if (!e->d_os_SP0) HIP_TRY(hipMalloc(&e->d_os_SP0, sizeof(float4) * 2 * OS_N2));
    mc_engine::OsKey key;
    OsMix mix;
    std::memset(&mix, 0, sizeof(mix));
    key.valid = true;
    key.n = st.nact;
    key.gen = e->ir_gen;
    int64_t lmax = 4;
    for (int a = 0; a < MC_MAXV; a++) {
        key.ir0[a] = key.ir1[a] = -1;
        key.g[a][0] = key.g[a][1] = key.g[a][2] = key.g[a][3] = 0.f;
        if (a >= st.nact) continue;
        const ActiveVoice& av = st.act[a];
        key.ir0[a] = (int)(av.ir0 - e->irs);
        key.ir1[a] = (int)(av.ir1 - e->irs);
        key.g[a][0] = av.ugain.x, key.g[a][1] = av.ugain.y, key.g[a][2] = av.ugain.z, key.g[a][3] = av.ugain.w;
        mix.h0[a] = av.ir0->d_h;
        mix.h1[a] = av.ir1->d_h;
        mix.L0[a] = (int)av.ir0->taps;
        mix.L1[a] = (int)av.ir1->taps;
        mix.g[a] = av.ugain;
        lmax = std::max<int64_t>(lmax, std::max<int64_t>(mix.L0[a], mix.L1[a]));
    }
    mix.n = st.nact;
    const mc_engine::OsKey& k0 = e->os_key;
    if (k0.valid && k0.n == key.n && k0.gen == key.gen && !std::memcmp(k0.ir0, key.ir0, sizeof(key.ir0)) &&
        !std::memcmp(k0.ir1, key.ir1, sizeof(key.ir1)) && !std::memcmp(k0.g, key.g, sizeof(key.g)))
        return MC_OK;
    const int64_t np = (lmax + 3) & ~(int64_t)3;
    if (e->os_planes_n < (size_t)np) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (e->d_os_planes) (void)hipFree(e->d_os_planes);
        e->d_os_planes = nullptr;
        e->os_planes_n = 0;
        HIP_TRY(hipMalloc(&e->d_os_planes, sizeof(float) * 4 * (size_t)np));
        e->os_planes_n = (size_t)np;
    }
    e->os_key.valid = false;
    hipLaunchKernelGGL(k_os_ir_mix, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, e->stream, mix, e->d_os_planes, np);
    OsGeo G;
    G.hop = OS_N;
    G.ovl = 0;
    G.n_in = np;
    G.tau0 = 0;
    G.base = 0;
    G.wet_end = np;
    G.seg0 = 0;
    for (int i = 0; i < 2; i++) {
        float4* buf = e->d_os_T + seg_el * i;
        hipLaunchKernelGGL(k_os_cols, dim3(OS_TPS), dim3(OS_THREADS), 0, e->stream, e->d_os_planes + (size_t)np * 2 * i,
                           e->d_os_planes + (size_t)np * (2 * i + 1), (const float*)nullptr, 0, G, buf, (float4*)nullptr, e->d_tw);
        hipLaunchKernelGGL(k_os_rows_fwd, dim3(OS_ITEMS), dim3(G2_THREADS), 0, e->stream, buf);
    }
    hipLaunchKernelGGL(k_os_ir_combine, dim3(OS_ITEMS), dim3(512), 0, e->stream, e->d_os_T, e->d_os_T + seg_el, e->d_os_SP, e->d_os_SP0,
                       0.5f / (float)OS_N);
    HIP_TRY(hipGetLastError());
    e->os_key = key;
    e->n_os[1]++;
    return MC_OK;
}

// The whole batch (T blocks from e->t_front, already staged in `st`) through the overlap-save passes.  Leaves the engine as
// the partitioned passes would: the last blocks' delay-line slots, slot gains and histories (k_fwd over the batch's tail), the
// Q1/Q2 prefix ring, the wet ring where later calls reach, and the last block's segment (its second half opens the next call).
int run_os(mc_engine* e, const Staged& st, mc_engine::BatchCtx& stored, const float* d_in1, const float* d_in2, float* d_outL, float* d_outR,
           int T, int ovl_blocks, int slot0, bool slice, int first, int count, int halo, const DropAhead* q8_da, int q8_shift) {
    // whole batch: wet frames of blocks [0, T).  Block-sliced: of the slice and the reach-back blocks before it (predelay),
    // the segments' history read from the batch's own buffers in front of the window
    const int w0 = slice ? first - halo : 0, wn = slice ? count + halo : T;
    const int64_t ovl = (int64_t)ovl_blocks * MC_B, hop = (int64_t)OS_N - ovl;
    const int hop_blocks = (int)(hop / MC_B);
    const int nseg = (wn + hop_blocks - 1) / hop_blocks;
    int rc = ensure_os(e, st, nseg);
    if (rc) return rc;
    const BlockParams* d_ptab = st.d_ptab;
    // (on HIP's special stream handles - MC_STREAM_DEFAULT = hipStreamLegacy, hipStreamPerThread - everything runs in line:
    // hipStreamWaitEvent on such a handle faults in this runtime)
    const hipStream_t main = e->stream;
    const hipStream_t side = (e->os_side && reinterpret_cast<uintptr_t>(main) > 2) ? e->os_stream : main;
    if (side != main) {  // the side stream starts where the engine's stream stands
        HIP_TRY(hipEventRecord(e->os_ev[0], main));
        HIP_TRY(hipStreamWaitEvent(side, e->os_ev[0], 0));
    }
    const int hist_from = (int)std::max<int64_t>(0, (int64_t)T - (int64_t)((e->cfg.n_ref + MC_MAX_PREDELAY) / MC_B + 4));
    DropAhead da;
    std::memset(&da, 0, sizeof(da));
    if (!slice) {
        // state for later calls: delay line, slot gains, input / gain histories of the last blocks (what any later window,
        // Q8 pass or re-render of a predelay epoch can reach), and the last block's segment as the partitioned passes leave it
        // (its partition sums from the delay line, one inverse transform: its second half opens the next call)
        const int reach = std::max(round_up(e->Pcap, 16) + 64, (int)((e->cfg.n_ref + MC_MAX_PREDELAY) / MC_B) + 8);
        const int from = std::max(0, T - reach) & ~(FWD_TILE - 1);
        if (q8_shift >= 0) {
            // Q8 regime at the shipped shape: the forward transforms of ALL blocks sum the cut terms of output block t + shift while
            // X_t is in registers (k_fwd<true>, DropAhead) - delay-line slots still only for the tail; the first `shift` output
            // blocks, whose source lies before the batch, through k_drop_fft from the slots the previous call left
            hipLaunchKernelGGL(k_fwd<true>, dim3((T + FWD_TILE - 1) / FWD_TILE), dim3(XF_THREADS), 0, side, d_in1, d_in2, 1, (int64_t)T * MC_B, T,
                               e->d_fdl, e->ring, slot0, d_ptab, 0, (float4*)nullptr, e->d_slotgain, e->d_tw, e->d_fdl16, e->d_xhist, e->xr, e->d_gring,
                               e->rc, (int64_t)e->t_front, 0, T, T, hist_from, 0, *q8_da, from);
            const TailDrop tdq = make_taildrop(e, st.ctx.vir, st.ctx.predelay);
            const int nd = std::min(T, q8_shift);
            hipLaunchKernelGGL(k_drop_fft, dim3((nd + DF_WAVES - 1) / DF_WAVES), dim3(64 * DF_WAVES), 0, side, tdq, e->d_dropbuf, (int64_t)st.ctx.t0, 0, nd,
                               (int64_t)st.ctx.predelay, (int64_t)e->cfg.n_ref, e->pm, (int64_t)e->epoch_b0);
            (q8_shift < T ? e->n_drop_ahead : e->n_drop_fft)++;
            stored.drop_done = true;
        } else
            hipLaunchKernelGGL(k_fwd<false>, dim3((T - from + FWD_TILE - 1) / FWD_TILE), dim3(XF_THREADS), 0, side, d_in1, d_in2, 1, (int64_t)T * MC_B, T,
                               e->d_fdl, e->ring, slot0, d_ptab, 0, (float4*)nullptr, e->d_slotgain, e->d_tw, e->d_fdl16, e->d_xhist, e->xr, e->d_gring,
                               e->rc, (int64_t)e->t_front, 0, 0, from, hist_from, from, da, 0);
        MacOut mo;
        const uint64_t b = e->t_front + (uint64_t)T - 1;
        e->stream = side;  // (the MAC and inverse-transform launchers use the engine's stream)
        rc = launch_mac_batch(e, st.act, st.nact, false, 1, (int)(b & (uint64_t)(e->ring - 1)), &mo);
        if (!rc) launch_inv(e, mo, b, side);
        e->stream = main;
        if (rc) return rc;
    } else if (stored.need_b0 < T) {
        // block-sliced: the tail of the batch that the next call's windows reach back to (delay line, histories, block sums), as k_fwd leaves it
        const int from = stored.need_b0 & ~(FWD_TILE - 1);
        hipLaunchKernelGGL(k_fwd<false>, dim3((T - from + FWD_TILE - 1) / FWD_TILE), dim3(XF_THREADS), 0, side, d_in1, d_in2, 1, (int64_t)T * MC_B, T,
                           e->d_fdl, e->ring, slot0, d_ptab, 0, st.d_sums, e->d_slotgain, e->d_tw, e->d_fdl16, e->d_xhist, e->xr, e->d_gring,
                           e->rc, (int64_t)e->t_front, 0, 0, stored.need_b0, hist_from, from, da, 0);
    }
    const int head = slice ? 0 : (int)std::min<uint64_t>((uint64_t)T, (st.ctx.predelay + MC_B - 1) / MC_B);
    OutArgs oa;
    std::memset(&oa, 0, sizeof(oa));
    oa.in1 = d_in1;
    oa.in2 = d_in2;
    oa.outL = d_outL;
    oa.outR = d_outR;
    oa.ptab = d_ptab;
    oa.pstride = 0;
    oa.cring = e->d_cring;
    oa.rc = e->rc;
    oa.tabs0 = (int64_t)st.ctx.t0;
    oa.predelay = (int64_t)st.ctx.predelay;
    oa.n_ref = (int64_t)e->cfg.n_ref;
    oa.b0 = make_retired(e).b0;
    oa.compat = (int)e->cfg.compat;
    oa.pm = e->pm;
    oa.blk0 = 0;
    oa.drop = (!slice && q8_shift >= 0) ? e->d_dropbuf : nullptr;
    if (slice) {  // the slice [first, first + count), nothing else; a sliced engine keeps no wet history
        oa.out_from = first;
        oa.out_end = first + count;
        oa.out_blk0 = first;
        oa.wet_head = 0;
        oa.wet_from = 1 << 30;
    } else {
        oa.out_from = head;
        oa.out_end = T;
        oa.out_blk0 = 0;
        oa.wet_head = head;
        oa.wet_from = std::max(0, T - (MC_MAX_PREDELAY / MC_B + 8));
    }
    OsGeo G;
    G.hop = hop;
    G.ovl = ovl;
    G.n_in = (int64_t)T * MC_B;
    G.tau0 = (int64_t)e->t_front * MC_B;
    G.base = (int64_t)w0 * MC_B;
    G.wet_end = (int64_t)(w0 + wn) * MC_B;
    G.seg0 = 0;
    hipLaunchKernelGGL(k_os_cols, dim3(nseg * OS_TPS), dim3(OS_THREADS), 0, main, d_in1, d_in2, (const float*)e->d_xhist, e->xr, G, e->d_os_T,
                       e->cfg.compat ? e->d_os_part : (float4*)nullptr, e->d_tw);
    if (side != main) {
        HIP_TRY(hipEventRecord(e->os_ev[1], main));
        HIP_TRY(hipStreamWaitEvent(side, e->os_ev[1], 0));
    }
    {  // Q1/Q2 prefix sums of the batch from the column pass's partial sums, ahead of the output pass
        CorrArgs ca;
        std::memset(&ca, 0, sizeof(ca));
        ca.sums = st.d_sums;
        ca.parts = e->d_os_part;
        ca.parts_hop = hop_blocks;
        ca.parts_ovl = ovl_blocks;
        ca.parts_t0 = w0;
        ca.parts_end = w0 + wn;
        ca.ptab = d_ptab;
        ca.pstride = 0;
        ca.T = T;
        ca.vs = st.ctx.vs;
        ca.inv_n = 1.0 / (double)e->cfg.n_ref;
        ca.compat = (int)e->cfg.compat;
        ca.cring = e->d_cring;
        ca.rc = e->rc;
        ca.tabs0 = (int64_t)st.ctx.t0;
        ca.ctot = e->d_ctot;
        corr_chunks(&ca, T, stored.need_a0, stored.need_a1, stored.need_b0);
        hipLaunchKernelGGL(k_corr_terms, dim3(ca.nchunks), dim3(CORR_CHUNK), 0, side, ca);
        if (ca.nchunks > 1) hipLaunchKernelGGL(k_corr_fix, dim3(ca.nchunks), dim3(CORR_CHUNK), 0, side, ca);
    }
    if (side != main) HIP_TRY(hipEventRecord(e->os_ev[2], side));
    if (e->ktiming && e->kev_n == kEvPool) {
        rc = drain_kernel_events(e);
        if (rc) return rc;
    }
    if (e->ktiming) {
        e->kev_blocks[e->kev_n] = (uint32_t)wn;
        HIP_TRY(hipEventRecord(e->kev[e->kev_n][0], main));
    }
    hipLaunchKernelGGL(k_os_rows, dim3(nseg * OS_ITEMS), dim3(G2B_THREADS), 0, main, e->d_os_T, (const float4*)e->d_os_SP, (const float4*)e->d_os_SP0, nseg);
    if (e->ktiming) {
        HIP_TRY(hipEventRecord(e->kev[e->kev_n][1], main));
        e->kev_n++;
        e->ks.resident = 1;
        e->ks.partitions = (uint32_t)ovl_blocks;
        e->ks.fast_levels = 253u;
    }
    if (side != main) HIP_TRY(hipStreamWaitEvent(main, e->os_ev[2], 0));  // (everything the side stream did: the output pass needs the prefix ring, later calls the rest)
    hipLaunchKernelGGL(k_os_out, dim3(nseg * OS_TPS), dim3(OS_THREADS), 0, main, (const float4*)e->d_os_T, G, e->d_wet, e->wr, e->d_tw, oa);
    HIP_TRY(hipGetLastError());
    stored.out_from = head;
    stored.corr_done = true;
    e->n_os[0]++;
    return MC_OK;
}

// forward transform + MAC + inverse + overlap-add; lin != null -> sharded partial.
// first/count: the output blocks of the batch this engine will finish (block-sliced operation when count < T):
// everything that later calls depend on (delay line, gains, Q1/Q2 sums, histories) is still produced for all T
// blocks, the partition sums and inverse transforms only for the slice and the few blocks before it that the
// overlap-add and the predelay reach back to.
// d_outL / d_outR != null: the caller finishes the batch right away (run_back follows with the same buffers), so the
// inverse-transform launches may write the output themselves
int run_front(mc_engine* e, const float* d_in1, const float* d_in2, int T, float* lin, int first, int count,
              float* d_outL = nullptr, float* d_outR = nullptr) {
    if (T <= 0 || T > e->Tmax) return fail(MC_ERR_ARG, "nblocks %d outside [1, %d]", T, e->Tmax);
    if (e->pipe_count >= kPipe) return fail(MC_ERR_STATE, "%d batches already await mc_finish_batch_device", kPipe);
    if (T % e->pm) return fail(MC_ERR_ARG, "nblocks %d is not a multiple of the period (%d blocks)", T, e->pm);
    {
        int rc = leave_jack_path(e);
        if (rc) return rc;
    }
    const bool slice = !(first == 0 && count == T);
    if (first < 0 || count <= 0 || first + count > T) return fail(MC_ERR_ARG, "slice [%d, %d) outside the batch of %d", first, first + count, T);
    if (slice && lin) return fail(MC_ERR_ARG, "a partition shard cannot be block-sliced");
    if (slice && e->res_end > e->t_front * MC_B) return fail(MC_ERR_STATE, "block-sliced call while a retired predelay epoch is ringing out");
    if (!slice && e->sliced) return fail(MC_ERR_STATE, "whole-batch call on a block-sliced engine (mc_reset first)");
    // pipelined: this batch's parity owns one set of scratch buffers and one slot of the parameter tables; the
    // post stage of the batch two calls ago must be done with them
    const bool piped = e->pipelined && !lin;
    const int par = (int)(e->batch_seq & 1);
    if (e->pipelined && !piped) {
        int rc = drain_post(e);
        if (rc) return rc;
    }
    if (piped) {
        if (e->post_pending[par]) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_post[par], 0));
        e->d_Y = e->d_Ybuf[par];
        e->d_Yc = e->d_Ycbuf[par];
        e->d_part = e->d_partbuf[par];
        e->d_tail = e->d_tailbuf[par];
    }
    const hipStream_t inv_stream = piped ? e->post_stream : e->stream;
    Staged st;
    int halo = 0;  // blocks before the slice that the predelay and the overlap-add reach back to
    {
        mc_cc_value cc[2];
        int rc = sample_params(e, cc);
        if (rc) return rc;
        if (slice) {  // checked before any state advances
            halo = (int)((cc[0].predelay + MC_B - 1) / MC_B) + 1;
            halo = (int)std::min<uint64_t>((uint64_t)halo, e->t_front + (uint64_t)first);  // the stream starts at block 0
            if (count + halo > e->Tcap) return fail(MC_ERR_ARG, "slice of %d blocks + %d blocks of reach-back exceeds the capacity of %d", count, halo, e->Tcap);
            if (e->sliced && e->slice_first != first) return fail(MC_ERR_STATE, "the slice start moved from block %d to %d (mc_reset first)", e->slice_first, first);
        }
        rc = retire_epoch(e, cc[0].predelay);
        if (!rc) rc = stage_params(e, T, cc, &st);
        if (rc) return rc;
    }
    if (slice) {
        e->sliced = true;
        e->slice_first = first;
    }
    // Blocks of this batch that some window of this engine can reach - now (its slice and what lies within one
    // reference length + the largest predelay before it) or from the next call (the same distance before the next
    // slice start).  Only those are transformed; a whole-batch call needs them all.
    int need_a0 = 0, need_a1 = T, need_b0 = T;
    if (slice) {
        const int64_t reach = (int64_t)(e->cfg.n_ref / MC_B) + MC_MAX_PREDELAY / MC_B + 2;
        need_a0 = (int)std::max<int64_t>(0, (int64_t)first - reach);
        need_a1 = first + count;
        need_b0 = (int)std::max<int64_t>(0, std::min<int64_t>(T, (int64_t)T + first - reach));
    }
    const uint64_t wblock = e->t_front + (uint64_t)first - (uint64_t)halo;  // first block of the window (absolute)
    st.ctx.first = first;
    st.ctx.count = count;
    st.ctx.need_a0 = need_a0;
    st.ctx.need_a1 = need_a1;
    st.ctx.need_b0 = need_b0;
    st.ctx.win0 = wblock * MC_B;
    // the finished output comes from this engine alone: overlap-add in the inverse-transform kernel, straight into the
    // wet ring (a partition shard's partial goes through k_ola and the segment ring instead)
    // ... unless the inverse transforms can emit the delayed partial themselves: no retired epoch ringing out)
    const bool lin_fused = lin && e->inv_to_wet && e->fuse_out && e->res_end <= e->t_front * MC_B;
    const bool to_wet = (!lin && e->inv_to_wet) || lin_fused;
    st.ctx.wet_ready = to_wet;
    e->pipe[(e->pipe_head + e->pipe_count) % kPipe] = st.ctx;
    e->spec_valid = e->dspec.valid = false;
    BlockParams* d_ptab = st.d_ptab;
    float4* d_sums = st.d_sums;
    const int pstride = st.ctx.pstride;

    const int slot0 = (int)(e->t_front & (uint64_t)(e->ring - 1));

    {  // long settled batches: overlap-save segments instead of the three passes below (ossave.hip.h)
        int os_ovl = 0;
        const int halo_full = (int)((st.ctx.predelay + MC_B - 1) / MC_B) + 1;
        const bool os_shape = !lin && !piped && to_wet && (!slice || halo == halo_full || wblock == 0);
        DropAhead os_da;
        std::memset(&os_da, 0, sizeof(os_da));
        int os_shift = -1;
        bool q8_ok = true;
        if (os_shape && count >= e->os_min_blocks && make_taildrop(e, st.ctx.vir, st.ctx.predelay).on) {
            q8_ok = false;
            if (!slice && e->os_on && e->fuse_drop && e->drop_ahead && e->pm == 1 && e->epoch_b0 <= e->t_front && e->res_end <= e->t_front * MC_B) {
                const int rc_t = plan_drop_ahead(e, st.ctx.vir, st.ctx.predelay, &os_da, &os_shift);
                if (rc_t != MC_OK) return rc_t;
                q8_ok = os_shift >= 0;
            }
        }
        if (os_shape && os_applies(e, st, count, slice, d_in1, d_in2, d_outL, d_outR, q8_ok, &os_ovl)) {
            const int rc = run_os(e, st, e->pipe[(e->pipe_head + e->pipe_count) % kPipe], d_in1, d_in2, d_outL, d_outR, T, os_ovl, slot0, slice, first, count, halo,
                                  &os_da, os_shift);
            if (rc) return rc;
            e->pipe_count++;
            e->batch_seq++;
            e->t_front += (uint64_t)T;
            return MC_OK;
        }
    }

    // K1: the blocks this engine can reach (all T unless block-sliced).  The input history ring serves the Q8 pass of
    // LATER calls (a batch reads its own samples from its input buffers): they look back less than one reference
    // length + the largest predelay, so a long batch keeps only its tail.
    const int hist_from = (int)std::max<int64_t>(0, (int64_t)T - (int64_t)((e->cfg.n_ref + MC_MAX_PREDELAY) / MC_B + 4));
    DropAhead da;
    std::memset(&da, 0, sizeof(da));
    int da_shift = -1;
    {
        // one launch per run of needed blocks (a whole-batch call: one launch over everything)
        int lo[2] = {need_a0 & ~(FWD_TILE - 1), need_b0 & ~(FWD_TILE - 1)}, hi[2] = {need_a1, T};
        if (lo[1] <= hi[0]) hi[0] = T, lo[1] = T;  // the two runs meet
        // Q8 regime, a whole batch whose output blocks each lose ONE term of ONE source block: the forward transforms sum the cut terms
        // themselves (DropAhead; da_shift = -1 otherwise and k_drop_fft sums them all)
        if (!slice && !lin && to_wet && !e->pipelined && e->fuse_out && e->fuse_drop && e->drop_ahead && e->pm == 1 && first == 0 && count == T &&
            e->res_end <= e->t_front * MC_B && e->epoch_b0 <= e->t_front && hi[0] == T && lo[0] == 0) {
            const int rc_t = plan_drop_ahead(e, st.ctx.vir, st.ctx.predelay, &da, &da_shift);
            if (rc_t != MC_OK) return rc_t;
        }
        for (int r = 0; r < 2; r++)
            if (hi[r] > lo[r])
                hipLaunchKernelGGL(da_shift >= 0 ? k_fwd<true> : k_fwd<false>, dim3((hi[r] - lo[r] + FWD_TILE - 1) / FWD_TILE), dim3(XF_THREADS), 0, e->stream, d_in1, d_in2, 1,
                                   (int64_t)T * MC_B, T, e->d_fdl, e->ring, slot0, d_ptab, pstride, d_sums, e->d_slotgain, e->d_tw,
                                   e->d_fdl16, e->d_xhist, e->xr, e->d_gring, e->rc, (int64_t)e->t_front, need_a0, need_a1, need_b0,
                                   hist_from, lo[r], da, 0);
    }
    if (e->ktiming && e->kev_n == kEvPool) {
        int rc = drain_kernel_events(e);
        if (rc) return rc;
    }
    int lin_head = 0;  // blocks of a shard's partial that k_ola_head fills (fused form)
    {
        // a window that starts before this batch sees the previous batch's gains too
        bool per_slot = pstride != 0;
        if (slice)
            for (int a = 0; a < st.nact; a++)
                if (e->gain_change_block[st.act[a].v] + (uint64_t)st.act[a].p_end + (uint64_t)halo > e->t_front) per_slot = true;
        // K2 + K3.  The reach-back blocks run as their own short launch (for <= 33 blocks the streaming kernel), so
        // that the slice itself fills whole 256-block tiles of the resident kernel - unless the second-level
        // transform takes the batch, which has no tiles: then the window is one launch.
        const bool whole = halo > 0 && fft2_applies(e, st.act, st.nact, per_slot, halo + count);
        const int parts[2][2] = {{0, whole ? 0 : halo}, {whole ? 0 : halo, whole ? halo + count : count}};
        for (int h = 0; h < 2; h++) {
            const int off = parts[h][0], n = parts[h][1];
            if (n <= 0) continue;
            const uint64_t b = wblock + (uint64_t)off;
            const bool timed = e->ktiming && h == 1;  // the MAC of the blocks this engine finishes: the roofline kernel
            if (timed) {
                e->kev_blocks[e->kev_n] = (uint32_t)n;
                HIP_TRY(hipEventRecord(e->kev[e->kev_n][0], e->stream));
            }
            MacOut mo;
            // the Q1/Q2 prefix sums of the batch ride along with this launch and the inverse transforms' (fused form only;
            // they read nothing but the block sums k_fwd left and are needed first by k_post)
            CorrArgs ca;
            std::memset(&ca, 0, sizeof(ca));
            // (every riding workgroup looks at the totals of all chunks before it: beyond 160 chunks the two launches of
            // run_back are cheaper than that quadratic chain.  A block-sliced rank has few: only the runs it transforms)
            bool may_ride = h == 1 && !lin && !piped && to_wet && T > CORR_CHUNK && e->corr_ride;
            auto corr_args = [&]() {
                ca.sums = d_sums;
                ca.ptab = d_ptab;
                ca.pstride = pstride;
                ca.T = T;
                ca.vs = st.ctx.vs;
                ca.inv_n = 1.0 / (double)e->cfg.n_ref;
                ca.compat = (int)e->cfg.compat;
                ca.cring = e->d_cring;
                ca.rc = e->rc;
                ca.tabs0 = (int64_t)st.ctx.t0;
                ca.ctot = e->d_ctot;
                corr_chunks(&ca, T, need_a0, need_a1, need_b0);
            };
            if (may_ride) {
                corr_args();
                may_ride = ca.nchunks > 1 && ca.nchunks <= 160;
            }
            if (may_ride) {
                ca.chain = 1;
                ca.ticket = e->d_cticket;
                ca.ticket_base = e->cticket_base;
                ca.flags = e->d_cflag;
                ca.seq = e->cflag_seq + 1;
            }
            int rc = launch_mac_batch(e, st.act, st.nact, per_slot, n, (int)(b & (uint64_t)(e->ring - 1)), &mo, may_ride ? &ca : nullptr);
            if (rc) return rc;
            if (mo.corr_done) {  // the riding workgroups took their tickets
                e->cticket_base += (unsigned)ca.nchunks;
                e->cflag_seq++;
            }
            if (timed) {
                HIP_TRY(hipEventRecord(e->kev[e->kev_n][1], e->stream));
                e->kev_n++;
                e->ks.resident = mo.resident ? 1 : 0;
                e->ks.partitions = (uint32_t)mo.swept;
                e->ks.fast_levels = mo.lvl == -2 ? 254u : (mo.lvl < 0 ? 255u : (uint32_t)mo.lvl);  // 255 / 254 = second-level transform
            }
            if (piped) {  // the inverse transforms wait for this MAC on the post stream; the engine's stream moves on
                HIP_TRY(hipEventRecord(e->ev_mac[par][h], e->stream));
                HIP_TRY(hipStreamWaitEvent(inv_stream, e->ev_mac[par][h], 0));
            }
            {
                mc_engine::BatchCtx& stored = e->pipe[(e->pipe_head + e->pipe_count) % kPipe];
                // The inverse transforms finish the output themselves when nothing but this batch's own wet signal and
                // final prefix sums go into it: a whole batch on one engine, no Q8 pass, no retired epoch ringing out.
                OutArgs oa;
                std::memset(&oa, 0, sizeof(oa));
                // whole batch: the launch covers it all.  Block-sliced: the launch covers the slice and the reach-back blocks
                // before it, from which the slice's first predelay frames come - unless the stream starts inside the reach-back
                const int halo_full = (int)((st.ctx.predelay + MC_B - 1) / MC_B) + 1;
                const bool covers = slice ? (n == halo + count && off == 0 && halo == halo_full) : (off == 0 && n == T);
                // ... and in the Q8 regime when the cut terms come as a buffer (k_drop_fft, whole batches): they depend on the delay line and
                // the gains only, both complete before the partition sums
                TailDrop tdq = make_taildrop(e, st.ctx.vir, st.ctx.predelay);
                const bool drop_buf = tdq.on && tdq.fft && !slice && !lin && e->fuse_drop;
                bool fuse = lin_fused || (h == 1 && d_outL && d_outR && e->fuse_out && to_wet && !piped && covers &&
                                          e->res_end <= e->t_front * MC_B && (!tdq.on || drop_buf));
                if (fuse && tdq.on && !lin_fused) {
                    const int rc_t = prepare_drop_fft(e, st.ctx.vir, st.ctx.predelay);
                    if (rc_t != MC_OK) return rc_t;
                    tdq = make_taildrop(e, st.ctx.vir, st.ctx.predelay);  // (with the partition-major spectra)
                    const int nd = da_shift >= 0 ? std::min(T, da_shift) : T;  // (the forward transforms summed the blocks from da_shift on)
                    (da_shift >= 0 && da_shift < T ? e->n_drop_ahead : e->n_drop_fft)++;
                    hipLaunchKernelGGL(k_drop_fft, dim3((nd + DF_WAVES - 1) / DF_WAVES), dim3(64 * DF_WAVES), 0, inv_stream, tdq, e->d_dropbuf, (int64_t)st.ctx.t0, 0, nd,
                                       (int64_t)st.ctx.predelay, (int64_t)e->cfg.n_ref, e->pm, (int64_t)e->epoch_b0);
                    stored.drop_done = true;
                }
                if (fuse) {
                    if (!mo.corr_done && !lin) {  // the prefix sums as launches of their own, ahead of their reader
                        corr_args();
                        ca.chain = 0;
                        hipLaunchKernelGGL(k_corr_terms, dim3(ca.nchunks), dim3(CORR_CHUNK), 0, inv_stream, ca);
                        if (ca.nchunks > 1) hipLaunchKernelGGL(k_corr_fix, dim3(ca.nchunks), dim3(CORR_CHUNK), 0, inv_stream, ca);
                        mo.corr_done = true;
                    }
                    const int head = slice ? 0 : (int)std::min<uint64_t>((uint64_t)T, (st.ctx.predelay + MC_B - 1) / MC_B);
                    oa.in1 = d_in1;
                    oa.in2 = d_in2;
                    oa.outL = d_outL;
                    oa.outR = d_outR;
                    oa.lin = lin;
                    oa.drop = stored.drop_done ? e->d_dropbuf : nullptr;
                    oa.ptab = d_ptab;
                    oa.pstride = pstride;
                    oa.cring = e->d_cring;
                    oa.rc = e->rc;
                    oa.tabs0 = (int64_t)st.ctx.t0;
                    oa.predelay = (int64_t)st.ctx.predelay;
                    oa.n_ref = (int64_t)e->cfg.n_ref;
                    oa.b0 = make_retired(e).b0;
                    oa.compat = (int)e->cfg.compat;
                    oa.pm = e->pm;
                    if (slice) {  // the slice [first, first + count), nothing else; a sliced engine keeps no wet history
                        oa.out_from = first;
                        oa.out_end = first + count;
                        oa.out_blk0 = first;
                        oa.wet_head = 0;
                        oa.wet_from = 1 << 30;
                    } else {
                        oa.out_from = head;
                        oa.out_end = T;
                        oa.out_blk0 = 0;
                        oa.wet_head = head;
                        oa.wet_from = std::max(0, T - (MC_MAX_PREDELAY / MC_B + 8));  // what later calls and a predelay change can reach
                    }
                    if (!lin) stored.out_from = head;
                    lin_head = head;
                }
                launch_inv(e, mo, b, inv_stream, to_wet, fuse ? &oa : nullptr);
                if (mo.corr_done) stored.corr_done = true;
            }
            if (piped && h == 0 && count < e->stream_threshold) {
                // both parts would use the streaming kernel's partial buffer: the second waits for the first's reader
                HIP_TRY(hipEventRecord(e->ev_mac[par][0], inv_stream));
                HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_mac[par][0], 0));
            }
        }
    }
    if (lin && lin_fused) {
        if (lin_head > 0)  // the frames the predelay fills from the previous call
            hipLaunchKernelGGL(k_ola_head, dim3(lin_head), dim3(256), 0, e->stream, e->d_wet, e->wr, T, (int64_t)e->t_front,
                               (int64_t)st.ctx.predelay, lin);
    } else if (lin)
        hipLaunchKernelGGL(k_ola, dim3(T), dim3(256), 0, e->stream, e->d_seg, e->sr, T, e->d_wet, e->wr, (int64_t)e->t_front,
                           (int64_t)st.ctx.predelay, make_retired(e), lin);
    HIP_TRY(hipGetLastError());
    e->pipe_count++;
    e->batch_seq++;
    e->t_front += (uint64_t)T;
    return MC_OK;
}

// Q1/Q2 prefix sums, predelay, clamp, dry mix of the oldest batch in flight.
// d_outL == null: the caller does not need this engine's output (a non-root
// rank of a reduce-to-root): nothing is launched, the batch is just retired.
// lin_first / lin_count >= 0 (partition shards after a reduce-scatter): lin_sum holds only blocks [lin_first, lin_first +
// lin_count) of the summed partial, [2][lin_count * 256], and only those blocks are finished (into d_outL / d_outR, which
// start at block lin_first); the Q1/Q2 prefix sums still run over the whole batch - every shard keeps that history.
int run_back(mc_engine* e, const float* d_in1, const float* d_in2, const float* lin_sum, float* d_outL, float* d_outR, int T,
             bool publish = false, int lin_first = -1, int lin_count = -1) {  // publish: the buffers are mapped host memory; raise the completion flag
    if (!e->pipe_count) return fail(MC_ERR_STATE, "no batch awaits its second half");
    const mc_engine::BatchCtx ctx = e->pipe[e->pipe_head];
    if (ctx.T != T) return fail(MC_ERR_ARG, "finish of %d blocks but the pending batch has %d", T, ctx.T);
    const bool lin_slice = lin_first >= 0;
    if (lin_slice && (!lin_sum || lin_count <= 0 || lin_first + lin_count > T || lin_first % e->pm || lin_count % e->pm))
        return fail(MC_ERR_ARG, "slice [%d, %d) of the summed partial outside the batch of %d blocks (or not whole periods)", lin_first, lin_first + lin_count, T);
    e->pipe_head = (e->pipe_head + 1) % kPipe;
    e->pipe_count--;
    if (d_outL && d_outR) {
        const BlockParams* d_ptab = e->d_ptab + (size_t)ctx.slot * e->Tmax;
        const float4* d_sums = e->d_sums + (size_t)ctx.slot * e->Tmax;
        // Q1/Q2 prefix sums of this batch (only where the output is finished: a non-root shard skips them)
        if (!ctx.corr_done) {  // (on the headline path they rode along with the front half's launches)
            CorrArgs ca;
            std::memset(&ca, 0, sizeof(ca));
            ca.sums = d_sums;
            ca.ptab = d_ptab;
            ca.pstride = ctx.pstride;
            ca.T = T;
            ca.vs = ctx.vs;
            ca.inv_n = 1.0 / (double)e->cfg.n_ref;
            ca.compat = (int)e->cfg.compat;
            ca.cring = e->d_cring;
            ca.rc = e->rc;
            ca.tabs0 = (int64_t)ctx.t0;
            ca.ctot = e->d_ctot;
            corr_chunks(&ca, T, ctx.need_a0, ctx.need_a1, ctx.need_b0);
            hipLaunchKernelGGL(k_corr_terms, dim3(ca.nchunks), dim3(CORR_CHUNK), 0, e->stream, ca);
            if (ca.nchunks > 1) hipLaunchKernelGGL(k_corr_fix, dim3(ca.nchunks), dim3(CORR_CHUNK), 0, e->stream, ca);  // a single chunk adds its base itself
        }
        {
            const int rc_t = prepare_drop_fft(e, ctx.vir, ctx.predelay);
            if (rc_t != MC_OK) return rc_t;
        }
        const bool piped = e->pipelined && !lin_sum && !publish;
        hipStream_t ps = e->stream;
        if (piped) {  // k_post follows this batch's inverse transforms on the post stream, after the prefix sums
            HIP_TRY(hipEventRecord(e->ev_corr[ctx.slot], e->stream));
            HIP_TRY(hipStreamWaitEvent(e->post_stream, e->ev_corr[ctx.slot], 0));
            ps = e->post_stream;
        }
        const TailDrop td_ = make_taildrop(e, ctx.vir, ctx.predelay);
        // the front half finished blocks >= out_from itself: only the blocks the predelay fills from the previous batch remain
        const int post_first = lin_slice ? lin_first : ctx.first;
        const int post_count = lin_slice ? lin_count : (ctx.out_from >= 0 ? ctx.out_from : ctx.count);
        TailDrop td = td_;
        if (post_count > 0 && td.on && td.fft && ctx.drop_done) td.dropbuf = e->d_dropbuf;  // (whole batch, indexed from block 0: post_first is 0 then)
        else if (post_count > 0 && td.on && td.fft) {
            e->n_drop_fft++;
            hipLaunchKernelGGL(k_drop_fft, dim3((post_count + DF_WAVES - 1) / DF_WAVES), dim3(64 * DF_WAVES), 0, ps, td, e->d_dropbuf, (int64_t)ctx.t0, post_first,
                               post_count, (int64_t)ctx.predelay, (int64_t)e->cfg.n_ref, e->pm, (int64_t)e->epoch_b0);
            td.dropbuf = e->d_dropbuf;
        }
        if (post_count > 0 && td.on && !td.fft) e->n_drop_tiles++;
        if (post_count > 0)
        hipLaunchKernelGGL(td.on ? (td.fft ? k_post<3> : k_post<1>) : k_post<0>, dim3((post_count + 3) / 4), dim3(256), 0, ps, ctx.wet_ready ? (const float*)nullptr : e->d_seg, e->sr,
                           lin_sum, e->d_wet, e->wr, e->d_cring,
                           e->rc, d_ptab, ctx.pstride, d_in1, d_in2, d_outL, d_outR, T, (int64_t)ctx.t0, post_first, post_count,
                           ctx.wet_ready ? INT64_MAX : (int64_t)ctx.win0, (int64_t)ctx.predelay, (int64_t)e->cfg.n_ref, (int)e->cfg.compat,
                           td, e->pm, make_retired(e), publish ? e->hd_flag : (unsigned*)nullptr,
                           publish ? ++e->flag_seq : 0u, e->d_done_ctr, lin_slice ? (int64_t)lin_count * MC_B : (int64_t)T * MC_B,
                           lin_slice ? lin_first : 0);
        HIP_TRY(hipGetLastError());
        if (piped) {
            HIP_TRY(hipEventRecord(e->ev_post[ctx.slot], e->post_stream));
            e->post_pending[ctx.slot] = true;
        }
    }
    e->t_abs = ctx.t0 + (uint64_t)T;
    return MC_OK;
}


// One JACK period with host buffers (Convolution::onProcess, conv.cu:287-466):
// zero-copy I/O through mapped pinned memory, the streaming MAC over partitions
// >= 1 (independent of the new block) and the fused k_tail1.
// The cross-fade has converged and nothing is ramping: staging the next block's parameters now or at the next call
// gives the same table entry (what lets a period be launched one call ahead, see process_one)
bool params_steady(const mc_engine* e, const mc_cc_value (&cc)[2]) {
    for (int i = 0; i < 2; i++) {
        if (cc[i].vsteps != 0) return false;
        bool have = false;
        for (int v = 0; v < MC_MAXV; v++) {
            const mc_engine::VoiceSlot& s = e->voice[i][v];
            if (s.ir < 0) continue;
            if (s.ir == (int)cc[i].select) {
                if (s.coef != (double)cc[i].wet) return false;
                have = true;
            } else if (s.coef != 0.0) {
                return false;
            }
        }
        if (!have) return false;
    }
    return true;
}

// The output of the period is on the host once its last kernel has published the sequence number: spin on the
// mapped word (a JACK callback blocks here anyway; the reference blocks in cudaEventSynchronize, conv.cu:455);
// fall back to a stream sync if it does not arrive in time.  Returns 1 when a parked tail says it gave up on its own.
int wait_period(mc_engine* e, unsigned seq) {
    if (!e->spin_wait) {
        HIP_TRY(hipEventSynchronize(e->ev_tail));
        return MC_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(e->h_flag, __ATOMIC_ACQUIRE) != seq) {
        if (__atomic_load_n(e->h_exited, __ATOMIC_ACQUIRE) == seq) return 1;
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(300)) {
            unpark(e);  // (a kernel parked for the NEXT period would hold the stream until its own timeout)
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (__atomic_load_n(e->h_flag, __ATOMIC_ACQUIRE) != seq) return fail(MC_ERR_HIP, "period did not complete");
            break;
        }
        __builtin_ia32_pause();
    }
    return MC_OK;
}

// Tagged output (mc_engine::tio): the period's 512 output granules carry its sequence number; they are on the host when every
// tag matches.  Same return values as wait_period.
int wait_period_tagged(mc_engine* e, unsigned seq) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    const int ngran = 2 * e->pm * MC_B;
    int first_missing = ngran - 1;  // (stores mostly arrive in order: watch the last one, then check them all)
    for (;;) {
        if ((unsigned)(__atomic_load_n(e->h_gran + first_missing, __ATOMIC_ACQUIRE) >> 32) == seq) {
            int i = 0;
            while (i < ngran && (unsigned)(__atomic_load_n(e->h_gran + i, __ATOMIC_ACQUIRE) >> 32) == seq) i++;
            if (i == ngran) return MC_OK;
            first_missing = i;
            continue;
        }
        if (__atomic_load_n(e->h_exited, __ATOMIC_ACQUIRE) == seq) return 1;
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(300)) {
            unpark(e);
            HIP_TRY(hipStreamSynchronize(e->stream));
            for (int i = 0; i < ngran; i++)
                if ((unsigned)(__atomic_load_n(e->h_gran + i, __ATOMIC_ACQUIRE) >> 32) != seq) return fail(MC_ERR_HIP, "period did not complete");
            return MC_OK;
        }
        __builtin_ia32_pause();
    }
}

// One JACK period with host buffers (Convolution::onProcess, conv.cu:287-466): zero-copy I/O through mapped pinned
// memory, the streaming MAC over the partitions that do not depend on the new block, and the fused k_tail1.
//
// Who sums what.  An engine that owns the IR's first partitions (every engine but a partition shard with
// part_begin > 0): the tail takes partition 0 (the new block) and partition 1 (the previous block, from the delay
// line); the streaming sweep takes partitions >= 2 - they pair with blocks at least two periods old, so the sweep of
// period t + 1 depends on nothing period t produces.  A shard that starts later sweeps its whole range.
//
// Launching a period ahead.  Measured (MC_JACK_TRACE build, profiles/r2_jack_trace.md): of 16 us per back-to-back call
// the tail kernel works 6.5 us; the rest is the host's launch call and the dispatch latency that follow the arrival of
// the period.  So, in the steady state, call t launches - behind the tail of period t - ONE kernel (k_jack) whose
// workgroup 0 is the tail of period t + 1 and whose other workgroups are the sweep of period t + 2.  The tail requests
// everything it needs except the period itself and parks on a doorbell in mapped memory; call t + 1 copies the period
// in, rings, and waits for the completion word: no launch on the critical path.  The parameters a parked period was
// staged with are compared with the ones the next call samples; any difference (a controller moved, an IR was
// selected) tells it to give up - it has written nothing - and the period is launched the ordinary way.  A parked
// tail that hears nothing for 100 ms gives up on its own (a host that stopped calling must not leave a kernel behind).
int process_one(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR) {
    if (!in1 || !in2 || !outL || !outR) return fail(MC_ERR_ARG, "null buffer");
    if (e->pipe_count) return fail(MC_ERR_STATE, "a sharded batch is still pending");
    if (e->sliced) return fail(MC_ERR_STATE, "single-period call on a block-sliced engine (mc_reset first)");
#ifdef MC_JACK_TRACE
    const auto tr0 = std::chrono::steady_clock::now();
#endif
    const size_t cap = (size_t)e->Thost * MC_B;
    auto copy_period_in = [&]() {  // the period where a tail launched now reads it
        if (e->bar_io) {
            std::memcpy(e->d_bar + 16, in1, sizeof(float) * MC_B);
            std::memcpy(e->d_bar + 16 + 4 * MC_B, in2, sizeof(float) * MC_B);
            _mm_sfence();  // (write-combined stores: out of the buffers before a doorbell or a launch can refer to them)
        } else {
            std::memcpy(e->h_io + 0 * cap, in1, sizeof(float) * MC_B);
            std::memcpy(e->h_io + 1 * cap, in2, sizeof(float) * MC_B);
        }
    };
    mc_cc_value cc[2];
    {
        int rc = sample_params(e, cc);
        if (rc) return rc;
    }

    // ---- the voices and ranges of a staged block
    struct Plan {
        ActiveVoice sweep[MC_MAXV];
        int lo[MC_MAXV], hi[MC_MAXV], nsweep = 0, nsum = 1;
        VoiceSet vset;
    };
    auto make_plan = [&](const Staged& st, Plan& pl) {
        std::memset(&pl.vset, 0, sizeof(pl.vset));
        pl.nsweep = 0;
        for (int a = 0; a < st.nact; a++) {
            int pb, pe;
            partition_range(e, st.act[a].p_end, &pb, &pe);
            if (pe <= pb) continue;
            if (pb == 0) {
                pl.vset.vid[pl.vset.n] = st.act[a].v;
                pl.vset.H0[pl.vset.n] = st.act[a].ir0->d_H;
                pl.vset.H1[pl.vset.n] = st.act[a].ir1->d_H;
                pl.vset.n++;
            }
            const int first = pb == 0 ? 2 : pb;
            if (pe > first) {
                pl.sweep[pl.nsweep] = st.act[a];
                pl.lo[pl.nsweep] = first;
                pl.hi[pl.nsweep] = pe;
                pl.nsweep++;
            }
        }
        pl.nsum = std::max(1, pl.nsweep) * e->nchunk;
    };
    auto sweep_uniform = [&](const ActiveVoice& av, int hi, uint64_t blk) {
        // one gain for all slots only when the last change is older than every slot of the sweep
        return e->gain_change_block[av.v] + (uint64_t)hi <= blk && e->gain_change_block[av.v] < blk;
    };
    // standalone sweep of block `blk` into its partial buffer (timed: the kernel the latency-mode roofline is quoted on)
    auto launch_sweep = [&](const Plan& pl, uint64_t blk) -> int {
        float4* dst = e->d_part_jack[blk & 1];
        if (!pl.nsweep) {
            HIP_TRY(hipMemsetAsync(dst, 0, sizeof(float4) * (size_t)MC_NB * pl.nsum, e->stream));
            return MC_OK;
        }
        if (e->ktiming) {
            if (e->kev_n >= kStampSlots) {
                int rc = drain_kernel_events(e);
                if (rc) return rc;
            }
            e->kev_blocks[e->kev_n] = 1;
            e->kev_stamped[e->kev_n] = true;
            HIP_TRY(hipEventRecord(e->kev[e->kev_n][0], e->stream));
        }
        unsigned long long* stamps = e->ktiming ? e->d_stamps + (size_t)2 * MC_STAMP_WGS * e->kev_n : nullptr;
        const int bslot0 = (int)(blk & (uint64_t)(e->ring - 1));
        int swept = 0;
        for (int a = 0; a < pl.nsweep; a++) {
            ActiveVoice av = pl.sweep[a];
            av.uniform = sweep_uniform(av, pl.hi[a], blk);
            launch_mac_stream(e, av, pl.lo[a], pl.hi[a], 1, bslot0, pl.nsum, a * e->nchunk, dst, stamps);
            swept = std::max(swept, pl.hi[a] - (pl.lo[a] == 2 ? 0 : pl.lo[a]));
        }
        if (e->ktiming) {
            HIP_TRY(hipEventRecord(e->kev[e->kev_n][1], e->stream));
            e->kev_n++;
            e->ks.resident = 0;
            e->ks.partitions = (uint32_t)swept;
        }
        return MC_OK;
    };
    auto remember_sweep = [&](const Plan& pl, const Staged& st, uint64_t blk) {
        e->spec_valid = true;
        e->spec_block = blk;
        e->spec_nact = pl.nsweep;
        for (int a = 0; a < pl.nsweep; a++) {
            e->spec_vir[0][a] = st.ctx.vir[0][pl.sweep[a].v];
            e->spec_vir[1][a] = st.ctx.vir[1][pl.sweep[a].v];
        }
    };
    auto sweep_matches = [&](const Plan& pl, const Staged& st, uint64_t blk) {
        bool ok = e->spec_valid && e->spec_block == blk && e->spec_nact == pl.nsweep;
        for (int a = 0; a < pl.nsweep && ok; a++)
            ok = e->spec_vir[0][a] == st.ctx.vir[0][pl.sweep[a].v] && e->spec_vir[1][a] == st.ctx.vir[1][pl.sweep[a].v];
        return ok;
    };
    int prep_rc = MC_OK;  // first failure of prepare_drop_fft inside tail_args (checked behind every launch that used it)
    auto tail_args = [&](const Staged& st, const Plan& pl, uint64_t blk, unsigned seq, bool parked) {
        TailArgs A;
        std::memset(&A, 0, sizeof(A));
        A.in1 = e->bar_io ? e->d_bar + 16 : e->hd_io + 0 * cap;
        A.in2 = e->bar_io ? e->d_bar + 16 + 4 * MC_B : e->hd_io + 1 * cap;
        A.vset = pl.vset;
        A.pstride_ir = e->Pstride;
        A.fdl = e->d_fdl;
        A.slotgain = e->d_slotgain;
        A.ring = e->ring;
        A.slot0 = (int)(blk & (uint64_t)(e->ring - 1));
        A.part = e->d_part_jack[blk & 1];
        A.nsum = pl.nsum;
        A.ptab = st.d_ptab;
        A.seg = e->d_seg;
        A.sr = e->sr;
        A.seg0 = (int)(blk & (uint64_t)(e->sr - 1));
        A.wet = e->d_wet;
        A.wr = e->wr;
        A.cring = e->d_cring;
        A.rc = e->rc;
        A.vs = st.ctx.vs;
        A.inv_n = 1.0 / (double)e->cfg.n_ref;
        A.compat = (int)e->cfg.compat;
        A.tabs0 = (int64_t)blk;
        A.predelay = (int64_t)st.ctx.predelay;
        A.n_ref = (int64_t)e->cfg.n_ref;
        A.outL = e->hd_io + 2 * cap;
        A.outR = e->hd_io + 3 * cap;
        A.g_tw = e->d_tw;
        {  // (Q8 regime: the last partitions partition-major and the buffer of the cut terms)
            const int r = prepare_drop_fft(e, st.ctx.vir, st.ctx.predelay);
            if (r != MC_OK && prep_rc == MC_OK) prep_rc = r;
        }
        A.td = make_taildrop(e, st.ctx.vir, st.ctx.predelay);
        A.fdl16 = e->d_fdl16;
        A.done_flag = e->hd_flag;
        A.seq = seq;
        A.ret = make_retired(e);
        A.bell = parked ? (e->bar_io ? reinterpret_cast<unsigned long long*>(e->d_bar) : e->hd_bell) : nullptr;
        A.exited = e->hd_exited;
        A.park_ticks = e->park_ticks;
        {
            const mc_engine::DropSpec& ds = e->dspec;
            if (A.td.on && ds.valid && ds.block == blk && ds.predelay == st.ctx.predelay && ds.epoch_b0 == e->epoch_b0 &&
                std::memcmp(ds.vir, st.ctx.vir, sizeof(ds.vir)) == 0) {
                A.drop = ds.buf;  // (summed by the launch before this one)
                e->n_drop_carried++;
            } else
                A.drop = launch_drop_period(e, A.td, blk, st.ctx.predelay);  // (queued ahead of the tail this argument block is for)
        }
        A.formcount = e->d_tailform;
        A.form = e->tail_form;
        A.in_gran = e->tio ? reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(e->d_bar) + 16384) : nullptr;
        A.out_gran = e->tio ? e->hd_gran : nullptr;
        return A;
    };

    // ---- this period: a parked tail staged with the same parameters, or the ordinary launch
    unsigned my_seq = 0;
    bool relaunch_ok = false;  // (a parked period that gave up on its own can be launched again as it was staged)
    Staged st_now;
    Plan pl_now;
    auto same_cc = [](const mc_cc_value& a, const mc_cc_value& b) {  // (field by field: the struct has padding)
        return a.select == b.select && a.predelay == b.predelay && a.speed == b.speed && a.vsteps == b.vsteps && a.dry == b.dry &&
               a.wet == b.wet && a.panDry == b.panDry && a.panWet == b.panWet && a.level == b.level;
    };
    const bool hit = e->pre.valid && e->pre.pm == 1 && e->pre.block == e->t_front && same_cc(cc[0], e->pre.cc[0]) &&
                     same_cc(cc[1], e->pre.cc[1]) && cc[0].predelay == e->cur_delay;
    if (hit) {
        my_seq = e->pre.seq;
        if (e->tio) {
            // tagged input: the period as granules {sample, sequence number} straight into device memory - every lane of the
            // parked tail is polling its own two; no doorbell, no second round trip for the data it would announce
            volatile unsigned long long* g = reinterpret_cast<volatile unsigned long long*>(reinterpret_cast<char*>(e->d_bar) + 16384);
            const unsigned long long tag = (unsigned long long)my_seq << 32;
            for (int i = 0; i < MC_B; i++) {
                uint32_t a, b;
                std::memcpy(&a, in1 + i, 4);
                std::memcpy(&b, in2 + i, 4);
                g[i] = tag | a;
                g[MC_B + i] = tag | b;
            }
            _mm_sfence();
        } else {
            copy_period_in();
            ring_bell(e, my_seq, 0);
        }
        e->pre.valid = false;
        e->n_park_hit++;
        st_now = e->pre.st;
        pl_now = Plan();
        make_plan(st_now, pl_now);
        relaunch_ok = true;
        // the sweep the parked kernel carries is the one the NEXT period will use
        e->spec_valid = false;
        if (e->pre.carries_sweep) {
            e->spec_valid = true;
            e->spec_block = e->t_front + 1;
            e->spec_nact = 1;
            e->spec_vir[0][0] = e->pre.carried_vir[0];
            e->spec_vir[1][0] = e->pre.carried_vir[1];
        }
    } else {
        unpark(e);
        copy_period_in();
        {
            int rc = drain_post(e);
            if (!rc) rc = retire_epoch(e, cc[0].predelay);
            if (!rc) rc = stage_params(e, 1, cc, &st_now);
            if (rc) return rc;
        }
        make_plan(st_now, pl_now);
        if (!sweep_matches(pl_now, st_now, e->t_front)) {
            int rc = launch_sweep(pl_now, e->t_front);
            if (rc) return rc;
        }
        e->spec_valid = false;
        my_seq = ++e->flag_seq;
        hipLaunchKernelGGL(k_tail1, dim3(1), dim3(TAIL1_THREADS), 0, e->stream, tail_args(st_now, pl_now, e->t_front, my_seq, false));
        if (prep_rc) return prep_rc;
        HIP_TRY(hipGetLastError());
    }
    if (!e->spin_wait) HIP_TRY(hipEventRecord(e->ev_tail, e->stream));
    e->batch_seq++;
    e->t_front += 1;
    e->t_abs = e->t_front;

    // ---- the next period: parked one call ahead when nothing is moving, else only its sweep (as a speculation: every
    // slot carries its own gains, so it stays exact under any parameter change except a change of the sounding IR set
    // or an IR reload, which the next call checks)
    bool parked_next = false;
    if (e->park && e->stream == e->own_stream && e->speculate && e->spin_wait && !e->pipelined && !e->ktiming && !e->half &&
        params_steady(e, cc) && cc[0].predelay == e->cur_delay) {
        Staged st_next;
        int rc = stage_params(e, 1, cc, &st_next);  // (steady: advancing the cross-fade by a block changes nothing)
        if (rc) return rc;
        Plan pl_next;
        make_plan(st_next, pl_next);
        if (st_next.ctx.pstride == 0 && pl_next.nsweep <= 1 && std::memcmp(&st_next.first, &st_now.first, sizeof(BlockParams)) == 0) {
            // the sweep of the period about to be parked, unless the kernel ahead of it already carries it
            if (!sweep_matches(pl_next, st_next, e->t_front)) {
                rc = launch_sweep(pl_next, e->t_front);
                if (rc) return rc;
                remember_sweep(pl_next, st_next, e->t_front);
            }
            const unsigned seq = ++e->flag_seq;
            TailArgs A = tail_args(st_next, pl_next, e->t_front, seq, true);
            if (prep_rc) return prep_rc;
            SweepArgs S;
            std::memset(&S, 0, sizeof(S));
            const uint64_t blk2 = e->t_front + 1;  // the sweep this kernel carries
            bool uni = true;
            if (pl_next.nsweep == 1) {
                const ActiveVoice& av = pl_next.sweep[0];
                const int span = pl_next.hi[0] - pl_next.lo[0];
                uni = sweep_uniform(av, pl_next.hi[0], blk2);
                S.H0 = av.ir0->d_H;
                S.H1 = av.ir1->d_H;
                S.pstride_ir = e->Pstride;
                S.p_begin = pl_next.lo[0];
                S.p_end = pl_next.hi[0];
                S.chunk = round_up(std::max(1, (span + e->nchunk - 1) / e->nchunk), 64);
                S.fdl = e->d_fdl;
                S.slotgain = e->d_slotgain + (size_t)av.v * e->ring;
                S.ring = e->ring;
                S.slot0 = (int)(blk2 & (uint64_t)(e->ring - 1));
                S.part = e->d_part_jack[blk2 & 1];
                S.nsum = pl_next.nsum;
                S.ch_off = 0;
                S.ugain = av.ugain;
                S.inv = make_float2(1.f, 1.f);
                S.nchunk = e->nchunk;
            }
            // the cut terms of the period after the parked one ride along (its sweep does: same launch, one more workgroup)
            const bool carry = e->carry_drop && A.td.on && A.td.fft && e->pm == 1;
            S.drop_next = carry ? e->d_drop[blk2 & 1] : nullptr;
            const dim3 grid(1 + (pl_next.nsweep == 1 ? MC_NB * e->nchunk : 0) + (carry ? 1 : 0));
            e->dspec.valid = false;
            if (carry) {
                e->dspec.valid = true;
                e->dspec.block = blk2;
                e->dspec.predelay = st_next.ctx.predelay;
                e->dspec.epoch_b0 = e->epoch_b0;
                std::memcpy(e->dspec.vir, st_next.ctx.vir, sizeof(e->dspec.vir));
                e->dspec.buf = S.drop_next;
            }
            if (uni)
                hipLaunchKernelGGL(k_jack<true>, grid, dim3(TAIL1_THREADS), 0, e->stream, A, S);
            else
                hipLaunchKernelGGL(k_jack<false>, grid, dim3(TAIL1_THREADS), 0, e->stream, A, S);
            HIP_TRY(hipGetLastError());
            e->pre.valid = true;
            e->pre.pm = 1;
            e->pre.block = e->t_front;
            e->pre.seq = seq;
            e->pre.cc[0] = cc[0];
            e->pre.cc[1] = cc[1];
            e->pre.st = st_next;
            e->pre.carries_sweep = pl_next.nsweep == 1;
            if (pl_next.nsweep == 1) {
                e->pre.carried_vir[0] = st_next.ctx.vir[0][pl_next.sweep[0].v];
                e->pre.carried_vir[1] = st_next.ctx.vir[1][pl_next.sweep[0].v];
            }
            parked_next = true;
        }
    }
    if (!parked_next && e->speculate) {
        // (not steady, or more than one voice sounding: the next period's sweep alone, in the shadow of this period)
        Staged& st = st_now;
        Plan& pl = pl_now;
        if (pl.nsweep && !(e->spec_valid && e->spec_block == e->t_front)) {
            int rc = launch_sweep(pl, e->t_front);
            if (rc) return rc;
            remember_sweep(pl, st, e->t_front);
        }
    }
#ifdef MC_JACK_TRACE
    const auto tr1 = std::chrono::steady_clock::now();
#endif
    // the output of THIS block is on the host once its tail has published its sequence number.  Spin on the mapped word
    // (a JACK callback blocks here anyway; the reference blocks in cudaEventSynchronize, conv.cu:455).
    for (;;) {
        int rc = e->tio ? wait_period_tagged(e, my_seq) : wait_period(e, my_seq);
        if (rc == MC_OK) break;
        if (rc != 1) return rc;
        // the parked tail gave up on its own (the host was away for more than park_ms): the same period, launched the
        // ordinary way behind whatever is queued
        if (!relaunch_ok) return fail(MC_ERR_HIP, "period did not complete");
        relaunch_ok = false;
        e->n_park_timeout++;
        unpark(e);  // (the kernel parked for the period after this one must not run before it)
        copy_period_in();  // (a tail launched now reads the plain copy of the period)
        // That kernel (launched above, behind a tail that had already left the stream) ran at once: the sweep it carries - the
        // one of the period after next - went into the partial-sum buffer of THIS period (same parity) and read a delay line
        // without this period's block.  unpark() has forgotten it; this period's own partial sums are summed again, on the
        // same stream behind the stray sweep.
        {
            int rc2 = launch_sweep(pl_now, e->t_front - 1);
            if (rc2) return rc2;
        }
        my_seq = ++e->flag_seq;
        hipLaunchKernelGGL(k_tail1, dim3(1), dim3(TAIL1_THREADS), 0, e->stream, tail_args(st_now, pl_now, e->t_front - 1, my_seq, false));
        if (prep_rc) return prep_rc;
        HIP_TRY(hipGetLastError());
    }
#ifdef MC_JACK_TRACE
    const auto tr2 = std::chrono::steady_clock::now();
#endif
    if (e->tio) {
        for (int i = 0; i < MC_B; i++) {
            const uint32_t a = (uint32_t)e->h_gran[i], b = (uint32_t)e->h_gran[MC_B + i];
            std::memcpy(outL + i, &a, 4);
            std::memcpy(outR + i, &b, 4);
        }
    } else {
        std::memcpy(outL, e->h_io + 2 * cap, sizeof(float) * MC_B);
        std::memcpy(outR, e->h_io + 3 * cap, sizeof(float) * MC_B);
    }
#ifdef MC_JACK_TRACE
    {
        const auto tr3 = std::chrono::steady_clock::now();
        const unsigned long long* w = reinterpret_cast<const unsigned long long*>(e->h_flag);
        if (e->tr_n >= 100) {  // (skip the warm-up)
            e->tr_launch += std::chrono::duration<double, std::micro>(tr1 - tr0).count();
            e->tr_flag += std::chrono::duration<double, std::micro>(tr2 - tr0).count();
            e->tr_total += std::chrono::duration<double, std::micro>(tr3 - tr0).count();
            e->tr_kernel += (double)(w[2] - w[1]) * 0.01;
            e->tr_out += (double)(w[3] - w[1]) * 0.01;
            e->tr_clk += (double)w[4];
            for (int i = 0; i < 3; i++) e->tr_c[i] += (double)w[5 + i];
            if (e->tr_prev_end) e->tr_gap += (double)(w[1] - e->tr_prev_end) * 0.01;
        }
        e->tr_prev_end = w[2];
        e->tr_n++;
    }
#endif
    return MC_OK;
}

bool is_pinned_host(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory is "invalid value" to the runtime: not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// Host-buffer batch through the engine's own pinned staging buffer (pageable caller memory, as JACK's buffers are in
// the reference, conv.cu:321-328, 431-437): chunks of Thost blocks, each copied in, processed and copied out in turn.
int process_host_staged(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR, int T) {
    const size_t cap = (size_t)e->Thost * MC_B;
    for (int o = 0; o < T; o += e->Thost) {
        const int n = std::min(e->Thost, T - o);
        const size_t bytes = (size_t)n * MC_B * sizeof(float), off = (size_t)o * MC_B;
        std::memcpy(e->h_io + 0 * cap, in1 + off, bytes);
        std::memcpy(e->h_io + 1 * cap, in2 + off, bytes);
        HIP_TRY(hipMemcpyAsync(e->d_io[0], e->h_io + 0 * cap, bytes, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->d_io[1], e->h_io + 1 * cap, bytes, hipMemcpyHostToDevice, e->stream));
        int rc = run_front(e, e->d_io[0], e->d_io[1], n, nullptr, 0, n, e->d_io[2], e->d_io[3]);
        if (rc) return rc;
        rc = run_back(e, e->d_io[0], e->d_io[1], nullptr, e->d_io[2], e->d_io[3], n);
        if (!rc) rc = fence_post(e);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(e->h_io + 2 * cap, e->d_io[2], bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemcpyAsync(e->h_io + 3 * cap, e->d_io[3], bytes, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        std::memcpy(outL + off, e->h_io + 2 * cap, bytes);
        std::memcpy(outR + off, e->h_io + 3 * cap, bytes);
    }
    return MC_OK;
}

// Host-buffer batch with pinned caller memory: the DMA engines read and write the caller's buffers directly.  Chunks
// of Tdev blocks; chunk k's copy-in (H2D stream), chunk k - 1's kernels (engine stream) and chunk k - 2's copy-out
// (D2H stream) run together, over three staging sets.  Returns when the last output byte is in outL / outR.
int process_host_pinned(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR, int T) {
    if (!e->Tdev) {
        // whole chunks of the second-level transform where it applies, and LONG ones: the copies set the pace, and the
        // link carries 45-48 GB/s each way at once in copies of >= 33 MB but only 26 GB/s in 6.6 MB ones
        // (scripts/pcie_probe.py; 53-57 GB/s one way at a time)
        int tdev = (int)std::min<uint64_t>(mc_preferred_batch(e, std::min(e->Tmax, 32768)), (uint64_t)e->Tmax);
        tdev = std::max(tdev / e->pm * e->pm, e->pm);
        for (int b = 0; b < 3; b++)
            for (int i = 0; i < 4; i++) HIP_TRY(hipMalloc(&e->d_pio[b][i], sizeof(float) * (size_t)tdev * MC_B));
        HIP_TRY(hipStreamCreateWithFlags(&e->h2d_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&e->d2h_stream, hipStreamNonBlocking));
        for (int b = 0; b < 3; b++) {
            HIP_TRY(hipEventCreateWithFlags(&e->ev_h2d[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&e->ev_comp[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&e->ev_d2h[b], hipEventDisableTiming));
        }
        e->Tdev = tdev;
    }
    int k = 0;
    for (int o = 0; o < T; o += e->Tdev, k++) {
        const int n = std::min(e->Tdev, T - o), b = k % 3;
        const size_t bytes = (size_t)n * MC_B * sizeof(float), off = (size_t)o * MC_B;
        float* const* d = e->d_pio[b];
        // the staging set is free once chunk k - 3 has been computed (inputs) and copied out (outputs)
        if (k >= 3) HIP_TRY(hipStreamWaitEvent(e->h2d_stream, e->ev_comp[b], 0));
        HIP_TRY(hipMemcpyAsync(d[0], in1 + off, bytes, hipMemcpyHostToDevice, e->h2d_stream));
        HIP_TRY(hipMemcpyAsync(d[1], in2 + off, bytes, hipMemcpyHostToDevice, e->h2d_stream));
        HIP_TRY(hipEventRecord(e->ev_h2d[b], e->h2d_stream));
        HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_h2d[b], 0));
        if (e->host_out_direct) {
            // the kernels that finish the output store it straight into the caller's pinned buffers (posted writes over the
            // link): no copy-out, no second copy direction to take turns with the copy-in
            float *hl = nullptr, *hr = nullptr;
            HIP_TRY(hipHostGetDevicePointer((void**)&hl, outL + off, 0));
            HIP_TRY(hipHostGetDevicePointer((void**)&hr, outR + off, 0));
            // (the output goes over the link as posted writes: the partitioned passes store it in whole 1 KB rows, the overlap-save
            // form's output pass in 64-byte pieces - 119 000 against 109 000 x real time: the link prefers the former)
            e->os_hold = true;
            int rc = run_front(e, d[0], d[1], n, nullptr, 0, n, hl, hr);
            e->os_hold = false;
            if (rc) return rc;
            rc = run_back(e, d[0], d[1], nullptr, hl, hr, n);
            if (!rc) rc = fence_post(e);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(e->ev_comp[b], e->stream));
            continue;
        }
        if (k >= 3) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_d2h[b], 0));
        int rc = run_front(e, d[0], d[1], n, nullptr, 0, n, d[2], d[3]);
        if (rc) return rc;
        rc = run_back(e, d[0], d[1], nullptr, d[2], d[3], n);
        if (!rc) rc = fence_post(e);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e->ev_comp[b], e->stream));
        HIP_TRY(hipStreamWaitEvent(e->d2h_stream, e->ev_comp[b], 0));
        HIP_TRY(hipMemcpyAsync(outL + off, d[2], bytes, hipMemcpyDeviceToHost, e->d2h_stream));
        HIP_TRY(hipMemcpyAsync(outR + off, d[3], bytes, hipMemcpyDeviceToHost, e->d2h_stream));
        HIP_TRY(hipEventRecord(e->ev_d2h[b], e->d2h_stream));
    }
    HIP_TRY(hipStreamSynchronize(e->d2h_stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return MC_OK;
}

int process_host(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR, int T) {
    if (!in1 || !in2 || !outL || !outR) return fail(MC_ERR_ARG, "null buffer");
    if (T <= 0) return fail(MC_ERR_ARG, "nblocks %d < 1", T);
    if (T % e->pm) return fail(MC_ERR_ARG, "nblocks %d is not a multiple of the period (%d blocks)", T, e->pm);
    if (is_pinned_host(in1) && is_pinned_host(in2) && is_pinned_host(outL) && is_pinned_host(outR))
        return process_host_pinned(e, in1, in2, outL, outR, T);
    return process_host_staged(e, in1, in2, outL, outR, T);
}

// One JACK period of 512 / 1024 frames (pm = 2 / 4 blocks): the batch pipeline with zero-copy I/O - k_fwd reads the
// period from mapped host memory, k_post writes the output there and raises the completion flag.
int process_period(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR) {
    if (!in1 || !in2 || !outL || !outR) return fail(MC_ERR_ARG, "null buffer");
    if (e->pipe_count) return fail(MC_ERR_STATE, "a sharded batch is still pending");
    {
        int rc = drain_post(e);
        if (rc) return rc;
    }
    const int T = e->pm;
    const size_t bytes = (size_t)T * MC_B * sizeof(float), cap = (size_t)e->Thost * MC_B;
    std::memcpy(e->h_io + 0 * cap, in1, bytes);
    std::memcpy(e->h_io + 1 * cap, in2, bytes);
    const bool was_piped = e->pipelined;
    e->pipelined = false;  // a period is finished inside this call
    int rc = run_front(e, e->hd_io + 0 * cap, e->hd_io + 1 * cap, T, nullptr, 0, T);
    if (!rc) rc = run_back(e, e->hd_io + 0 * cap, e->hd_io + 1 * cap, nullptr, e->hd_io + 2 * cap, e->hd_io + 3 * cap, T, true);
    e->pipelined = was_piped;
    if (rc) return rc;
    if (!e->spin_wait) HIP_TRY(hipEventRecord(e->ev_tail, e->stream));
    rc = wait_period(e, e->flag_seq);
    if (rc) return rc;
    std::memcpy(outL, e->h_io + 2 * cap, bytes);
    std::memcpy(outR, e->h_io + 3 * cap, bytes);
    return MC_OK;
}

// One JACK period of 512 / 1024 frames on an unsharded engine: the streaming MAC over partitions >= pm (summed
// speculatively in the shadow of the previous period, as in process_one) and the fused k_tailp.  In the steady state the
// NEXT period's k_tailp is launched behind its sweep one call ahead and parks on the doorbell (as process_one's tail does):
// the call that brings the period writes it through the BAR, rings, and waits - no launch on its critical path.
int process_period_fused(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR) {
    if (!in1 || !in2 || !outL || !outR) return fail(MC_ERR_ARG, "null buffer");
    if (e->pipe_count) return fail(MC_ERR_STATE, "a sharded batch is still pending");
    if (e->sliced) return fail(MC_ERR_STATE, "single-period call on a block-sliced engine (mc_reset first)");
    const int pm = e->pm;
    const bool tio_p = e->tio && e->tio_long;  // tagged I/O pays for 256-frame periods only (see mc_engine::tio_long)
    const size_t bytes = (size_t)pm * MC_B * sizeof(float), cap = (size_t)e->Thost * MC_B;
    const float *pin1 = e->bar_io ? e->d_bar + 16 : e->hd_io + 0 * cap, *pin2 = e->bar_io ? e->d_bar + 16 + 4 * MC_B : e->hd_io + 1 * cap;
    auto copy_period_in = [&]() {  // the period where a tail launched now reads it
        if (e->bar_io) {  // straight into device memory through the BAR (see mc_engine::d_bar)
            std::memcpy(e->d_bar + 16, in1, bytes);
            std::memcpy(e->d_bar + 16 + 4 * MC_B, in2, bytes);
            _mm_sfence();
        } else {
            std::memcpy(e->h_io + 0 * cap, in1, bytes);
            std::memcpy(e->h_io + 1 * cap, in2, bytes);
        }
    };
    mc_cc_value cc[2];
    {
        int rc = sample_params(e, cc);
        if (rc) return rc;
    }

    // ---- what a staged period sweeps and what its tail multiplies itself
    struct PPlan {
        ActiveVoice sweep[MC_MAXV];
        int hi[MC_MAXV], nsweep = 0, nsum = 1;
        VoiceSet vset;
    };
    auto make_pplan = [&](const Staged& st, PPlan& pl) {
        std::memset(&pl.vset, 0, sizeof(pl.vset));
        pl.nsweep = 0;
        for (int a = 0; a < st.nact; a++) {
            const int pe = st.act[a].p_end;
            if (pe <= 0) continue;
            pl.vset.vid[pl.vset.n] = st.act[a].v;
            pl.vset.H0[pl.vset.n] = st.act[a].ir0->d_H;
            pl.vset.H1[pl.vset.n] = st.act[a].ir1->d_H;
            pl.vset.n++;
            if (pe > pm) {
                pl.sweep[pl.nsweep] = st.act[a];
                pl.hi[pl.nsweep] = pe;
                pl.nsweep++;
            }
        }
        pl.nsum = std::max(1, pl.nsweep) * e->nchunk;
    };
    // partitions >= pm of the pm blocks starting at `blk` pair only with blocks before blk
    auto launch_mac = [&](const PPlan& pl, uint64_t blk) -> int {
        const bool timed = e->ktiming;
        if (timed) {
            if (e->kev_n >= kStampSlots) {
                int rc = drain_kernel_events(e);
                if (rc) return rc;
            }
            e->kev_blocks[e->kev_n] = (uint32_t)pm;
            e->kev_stamped[e->kev_n] = true;
            HIP_TRY(hipEventRecord(e->kev[e->kev_n][0], e->stream));
        }
        unsigned long long* stamps = timed ? e->d_stamps + (size_t)2 * MC_STAMP_WGS * e->kev_n : nullptr;
        const int bslot0 = (int)(blk & (uint64_t)(e->ring - 1));
        int swept = 0;
        for (int a = 0; a < pl.nsweep; a++) {
            ActiveVoice av = pl.sweep[a];
            av.uniform = e->gain_change_block[av.v] + (uint64_t)pl.hi[a] <= blk && e->gain_change_block[av.v] < blk;
            launch_mac_stream(e, av, pm, pl.hi[a], pm, bslot0, pl.nsum, a * e->nchunk, nullptr, stamps);
            swept = std::max(swept, pl.hi[a]);
        }
        if (timed) {
            HIP_TRY(hipEventRecord(e->kev[e->kev_n][1], e->stream));
            e->kev_n++;
            e->ks.resident = 0;
            e->ks.partitions = (uint32_t)swept;
        }
        return MC_OK;
    };
    auto sweep_or_zero = [&](const PPlan& pl, uint64_t blk) -> int {
        if (pl.nsweep) return launch_mac(pl, blk);
        HIP_TRY(hipMemsetAsync(e->d_part, 0, sizeof(float4) * (size_t)pm * MC_NB * pl.nsum, e->stream));
        return MC_OK;
    };
    auto remember_sweep = [&](const PPlan& pl, const Staged& st, uint64_t blk) {
        e->spec_valid = true;
        e->spec_block = blk;
        e->spec_nact = pl.nsweep;
        for (int a = 0; a < pl.nsweep; a++) {
            e->spec_vir[0][a] = st.ctx.vir[0][pl.sweep[a].v];
            e->spec_vir[1][a] = st.ctx.vir[1][pl.sweep[a].v];
        }
    };
    auto sweep_matches = [&](const PPlan& pl, const Staged& st, uint64_t blk) {
        bool ok = e->spec_valid && e->spec_block == blk && e->spec_nact == pl.nsweep;
        for (int a = 0; a < pl.nsweep && ok; a++)
            ok = e->spec_vir[0][a] == st.ctx.vir[0][pl.sweep[a].v] && e->spec_vir[1][a] == st.ctx.vir[1][pl.sweep[a].v];
        return ok;
    };
    int prep_rc = MC_OK;  // first failure of prepare_drop_fft inside launch_tail (checked behind every call)
    auto launch_tail = [&](const Staged& st, const PPlan& pl, uint64_t blk, unsigned seq, bool parked) {
        const int slot0 = (int)(blk & (uint64_t)(e->ring - 1));
        const unsigned long long* bell = parked ? (e->bar_io ? reinterpret_cast<unsigned long long*>(e->d_bar) : e->hd_bell) : nullptr;
        {
            const int r = prepare_drop_fft(e, st.ctx.vir, st.ctx.predelay);
            if (r != MC_OK && prep_rc == MC_OK) prep_rc = r;
        }
        const TailDrop tdp = make_taildrop(e, st.ctx.vir, st.ctx.predelay);
        // Q8 regime: a parked tail sums its own cut terms while it waits for the period; one launched on arrival gets them from a launch ahead of it
        const bool self_drop = parked && tdp.on && tdp.fft && e->carry_drop;
        const float* drop = self_drop ? e->d_drop[(blk / (uint64_t)e->pm) & 1] : launch_drop_period(e, tdp, blk, st.ctx.predelay);
        if (self_drop) e->n_drop_carried++;
#define MC_LAUNCH_TAILP(PM)                                                                                                  \
    if (self_drop) MC_LAUNCH_TAILP_(PM, true); else MC_LAUNCH_TAILP_(PM, false)
#define MC_LAUNCH_TAILP_(PM, SD)                                                                                                  \
    hipLaunchKernelGGL((k_tailp<PM, SD>), dim3(1), dim3(256), 0, e->stream, pin1, pin2, pl.vset, e->Pstride, e->d_fdl, e->d_slotgain, \
                       e->ring, slot0, e->d_part, pl.nsum, st.d_ptab, e->d_seg, e->sr, e->d_wet, e->wr, e->d_cring, e->rc,    \
                       st.ctx.vs, 1.0 / (double)e->cfg.n_ref, (int)e->cfg.compat, (int64_t)blk, (int64_t)st.ctx.predelay,     \
                       (int64_t)e->cfg.n_ref, e->hd_io + 2 * cap, e->hd_io + 3 * cap, e->d_tw,                                \
                       tdp, e->d_fdl16, e->hd_flag, seq, make_retired(e), bell,     \
                       e->hd_exited, e->park_ticks, drop,                            \
                       tio_p ? reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(e->d_bar) + 16384) : nullptr, \
                       tio_p ? e->hd_gran : nullptr)
        if (pm == 2) {
            MC_LAUNCH_TAILP(2);
        } else {
            MC_LAUNCH_TAILP(4);
        }
#undef MC_LAUNCH_TAILP
#undef MC_LAUNCH_TAILP_
    };
    auto same_cc = [](const mc_cc_value& a, const mc_cc_value& b) {  // (field by field: the struct has padding)
        return a.select == b.select && a.predelay == b.predelay && a.speed == b.speed && a.vsteps == b.vsteps && a.dry == b.dry &&
               a.wet == b.wet && a.panDry == b.panDry && a.panWet == b.panWet && a.level == b.level;
    };

    // ---- this period: a parked tail staged with the same parameters, or the ordinary launches
    Staged st;
    PPlan pl;
    unsigned my_seq = 0;
    bool relaunch_ok = false;
    const bool hit = e->pre.valid && e->pre.pm == pm && e->pre.block == e->t_front && same_cc(cc[0], e->pre.cc[0]) &&
                     same_cc(cc[1], e->pre.cc[1]) && cc[0].predelay == e->cur_delay;
    if (hit) {
        my_seq = e->pre.seq;
        if (tio_p) {  // tagged input (see process_one): granules [2][pm * 256] {sample, sequence number}, no doorbell
            volatile unsigned long long* g = reinterpret_cast<volatile unsigned long long*>(reinterpret_cast<char*>(e->d_bar) + 16384);
            const unsigned long long tag = (unsigned long long)my_seq << 32;
            const int n = pm * MC_B;
            for (int i = 0; i < n; i++) {
                uint32_t a, b;
                std::memcpy(&a, in1 + i, 4);
                std::memcpy(&b, in2 + i, 4);
                g[i] = tag | a;
                g[n + i] = tag | b;
            }
            _mm_sfence();
        } else {
            copy_period_in();
            ring_bell(e, my_seq, 0);
        }
        e->pre.valid = false;
        e->n_park_hit++;
        st = e->pre.st;
        make_pplan(st, pl);
        relaunch_ok = true;
        e->spec_valid = false;  // (its sweep has been consumed)
    } else {
        int rc = drain_post(e);
        if (!rc) rc = leave_jack_path(e);
        copy_period_in();
        if (!rc) rc = retire_epoch(e, cc[0].predelay);
        if (!rc) rc = stage_params(e, pm, cc, &st);
        if (rc) return rc;
        if (st.ctx.pstride != 0) return fail(MC_ERR_STATE, "the blocks of one period must share their parameters");
        make_pplan(st, pl);
        if (!sweep_matches(pl, st, e->t_front)) {
            rc = sweep_or_zero(pl, e->t_front);
            if (rc) return rc;
        }
        e->spec_valid = false;
        my_seq = ++e->flag_seq;
        launch_tail(st, pl, e->t_front, my_seq, false);
        if (prep_rc) return prep_rc;
        HIP_TRY(hipGetLastError());
    }
    if (!e->spin_wait) HIP_TRY(hipEventRecord(e->ev_tail, e->stream));
    e->batch_seq++;
    e->t_front += (uint64_t)pm;
    e->t_abs = e->t_front;

    // ---- the next period: its sweep in the shadow of this one (a speculation the next call checks), and - when nothing
    // is moving - its tail behind the sweep, parked
    if (e->speculate && pl.nsweep) {
        int rc = launch_mac(pl, e->t_front);
        if (rc) return rc;
        remember_sweep(pl, st, e->t_front);
    }
    if (e->park && e->stream == e->own_stream && e->speculate && e->spin_wait && !e->pipelined && !e->ktiming && !e->half &&
        params_steady(e, cc) && cc[0].predelay == e->cur_delay) {
        Staged st_next;
        int rc = stage_params(e, pm, cc, &st_next);  // (steady: advancing the cross-fade by a call changes nothing)
        if (rc) return rc;
        PPlan pl_next;
        make_pplan(st_next, pl_next);
        if (st_next.ctx.pstride == 0 && std::memcmp(&st_next.first, &st.first, sizeof(BlockParams)) == 0 &&
            (pl_next.nsweep == 0 || sweep_matches(pl_next, st_next, e->t_front))) {
            if (!pl_next.nsweep) {
                rc = sweep_or_zero(pl_next, e->t_front);
                if (rc) return rc;
            }
            const unsigned seq = ++e->flag_seq;
            launch_tail(st_next, pl_next, e->t_front, seq, true);
            if (prep_rc) return prep_rc;
            HIP_TRY(hipGetLastError());
            e->pre.valid = true;
            e->pre.pm = pm;
            e->pre.block = e->t_front;
            e->pre.seq = seq;
            e->pre.cc[0] = cc[0];
            e->pre.cc[1] = cc[1];
            e->pre.st = st_next;
            e->pre.carries_sweep = false;
        }
    }
    for (;;) {
        int rc = tio_p ? wait_period_tagged(e, my_seq) : wait_period(e, my_seq);
        if (rc == MC_OK) break;
        if (rc != 1) return rc;
        copy_period_in();  // (a tail launched now reads the plain copy of the period)
        // the parked tail gave up on its own (the host was away for more than park_ms): the same period the ordinary way.
        // What was queued behind it worked on a delay line without this period: the next period's parked tail is told to
        // give up, its sweep is forgotten, and this period's own partial sums are summed again (the sweep behind overwrote them)
        if (!relaunch_ok) return fail(MC_ERR_HIP, "period did not complete");
        relaunch_ok = false;
        e->n_park_timeout++;
        unpark(e);
        e->spec_valid = false;
        rc = sweep_or_zero(pl, e->t_front - (uint64_t)pm);
        if (rc) return rc;
        my_seq = ++e->flag_seq;
        launch_tail(st, pl, e->t_front - (uint64_t)pm, my_seq, false);
        if (prep_rc) return prep_rc;
        HIP_TRY(hipGetLastError());
    }
    if (tio_p) {
        const int n = pm * MC_B;
        for (int i = 0; i < n; i++) {
            const uint32_t a = (uint32_t)e->h_gran[i], b = (uint32_t)e->h_gran[n + i];
            std::memcpy(outL + i, &a, 4);
            std::memcpy(outR + i, &b, 4);
        }
    } else {
        std::memcpy(outL, e->h_io + 2 * cap, bytes);
        std::memcpy(outR, e->h_io + 3 * cap, bytes);
    }
    return MC_OK;
}

}  // namespace

#include "singlefft_host.hip.h"


extern "C" {

uint32_t mc_abi_version(void) { return MC_ABI_VERSION; }
const char* mc_last_error(void) { return g_err; }

void mc_default_config(mc_config* cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(mc_config);
    cfg->device = -1;
    cfg->n_ref = 512 * 256;  // CONV_DEFAULT_FFTSIZE, conv.h:10-12
    cfg->max_batch = 256;
    cfg->compat = 1;
}

void mc_default_params(mc_cc_value* v) {
    if (!v) return;
    v->select = 0;  // conv.h:40-48
    v->predelay = 0;
    v->speed = 100;
    v->vsteps = 0;
    v->dry = 0.5f;
    v->wet = 0.5f;
    v->panDry = 0.0f;
    v->panWet = 0.0f;
    v->level = 1.0f;
}

int mc_create(const mc_config* cfg, mc_engine** out) {
    if (!cfg || !out) return fail(MC_ERR_ARG, "null argument");
    if (cfg->struct_size != sizeof(mc_config)) return fail(MC_ERR_ARG, "mc_config size mismatch (%u vs %zu)", cfg->struct_size, sizeof(mc_config));
    if (cfg->n_ref < 4096 || (cfg->n_ref & (cfg->n_ref - 1))) return fail(MC_ERR_ARG, "n_ref must be a power of two >= 4096");
    if (cfg->n_ref > (1ull << 26)) return fail(MC_ERR_ARG, "n_ref too large");
    if (cfg->max_batch < 1 || cfg->max_batch > 1048576) return fail(MC_ERR_ARG, "max_batch must be in [1, 1048576]");
    if ((cfg->part_begin % 16) || (cfg->part_end % 16)) return fail(MC_ERR_ARG, "partition shard bounds must be multiples of 16");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(MC_ERR_HIP, "no HIP device");
    int dev = cfg->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(MC_ERR_ARG, "device %d out of range (%d devices)", dev, ndev);
    HIP_TRY(hipSetDevice(dev));

    mc_engine* e = new (std::nothrow) mc_engine();
    if (!e) return fail(MC_ERR_NOMEM, "out of host memory");
    e->cfg = *cfg;
    e->device = dev;
    e->ph.update([](mc_cc_value (&cc)[2]) {
        mc_default_params(&cc[0]);
        mc_default_params(&cc[1]);
    });
    std::memset(&e->ks, 0, sizeof(e->ks));
    e->Tmax = (int)cfg->max_batch;
    e->Tcap = std::max(e->Tmax, 256);
    e->Pcap = cfg->max_partitions ? (int)cfg->max_partitions : (int)((cfg->n_ref - 1024 + MC_B - 1) / MC_B);
    e->Pstride = (int)next_pow2((uint64_t)round_up(e->Pcap, 16));
    e->ring = (int)next_pow2((uint64_t)e->Pstride + (uint64_t)e->Tcap + 64);  // + reach-back of a block slice (<= 33)
    e->sr = (int)next_pow2((uint64_t)e->Tcap + 4);  // power of two: ring indices are masks in the kernels; a slice + reach-back <= Tmax
    e->wr = (int)next_pow2((uint64_t)MC_MAX_PREDELAY + (uint64_t)e->Tmax * MC_B + 2 * MC_B);
    e->pipelined = cfg->pipeline != 0 && cfg->precision == 0;
    // (pipelined: the forward stage of batch k + 1 writes histories while the post stage of batch k still reads them)
    e->rc = (int)next_pow2(cfg->n_ref / MC_B + (uint64_t)e->Tmax * (e->pipelined ? 2 : 1) + 64);
    e->stream_threshold = cfg->stream_threshold ? (int)cfg->stream_threshold : 48;  // measured crossover (scripts/sweep_T.sh)
    if (const char* nc = LAB_ENV("MCCONV_NCHUNK")) e->nchunk = std::max(1, std::min(64, std::atoi(nc)));
    if (const char* nt = LAB_ENV("MCCONV_STREAM_NT")) e->stream_nt = std::atoi(nt) == 512 ? 512 : 256;
    {
        const uint32_t period = cfg->period ? cfg->period : MC_BLOCK;
        if (period != 256 && period != 512 && period != 1024) {
            delete e;
            return fail(MC_ERR_ARG, "period must be 256, 512 or 1024 frames");
        }
        e->pm = (int)(period / MC_BLOCK);
        if (e->Tmax % e->pm) {
            delete e;
            return fail(MC_ERR_ARG, "max_batch must be a multiple of period / 256");
        }
    }
    e->half = cfg->precision == 1;
    if (e->half) e->stream_threshold = e->Tmax + 1;  // the fp16 MAC is the streaming sweep
    // (at least 8 blocks: the fast-FIR form sends the last 2^levels blocks of a batch through the streaming kernel)
    e->Tstream = std::min(e->half ? e->Tmax : e->Tcap, std::max(8, e->stream_threshold - 1));

#define ENG_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) {                                                                           \
            int _rc = fail(MC_ERR_HIP, "%s -> %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            mc_destroy(e);                                                                                \
            return _rc;                                                                                   \
        }                                                                                                 \
    } while (0)

    if (const char* v = std::getenv("MCCONV_TD_FFT")) e->td_fft = std::atoi(v) != 0;
    ENG_TRY(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    {
        const char* fm = std::getenv("MCCONV_FORM");  // lets an unmodified host pick the form: "single" / "partitioned"
        if (fm && !std::strcmp(fm, "single")) e->cfg.form = 1;
        else if (fm && !std::strcmp(fm, "partitioned")) e->cfg.form = 0;
    }
    if (e->cfg.form > 1) {
        mc_destroy(e);
        return fail(MC_ERR_ARG, "mc_config.form must be 0 (partitioned) or 1 (single transform)");
    }
    if (e->cfg.form == 1) {  // the reference's own shape: none of the partitioned engine's state
        ENG_TRY(hipMalloc(&e->d_tw, sizeof(float2) * FFT_N));
        {
            std::vector<float2> tw;
            host_twiddles(tw);
            ENG_TRY(hipMemcpy(e->d_tw, tw.data(), sizeof(float2) * FFT_N, hipMemcpyHostToDevice));
        }
        e->Thost = 4;  // the mapped buffer holds one period
        ENG_TRY(hipHostMalloc(&e->h_io, sizeof(float) * 4 * (size_t)e->Thost * MC_B, hipHostMallocMapped));
        ENG_TRY(hipHostGetDevicePointer((void**)&e->hd_io, e->h_io, 0));
        ENG_TRY(hipHostMalloc(&e->h_flag, 256, hipHostMallocMapped));  // completion word of mc_process
        ENG_TRY(hipHostGetDevicePointer((void**)&e->hd_flag, e->h_flag, 0));
        std::memset(e->h_flag, 0, 256);
        if (std::getenv("MCCONV_NO_SPIN")) e->spin_wait = false;
        {  // the period through the BAR where the CPU can write device memory (see mc_engine::d_bar)
            int large_bar = 0;
            const char* bi = std::getenv("MCCONV_BAR_IO");
            if ((!bi || std::atoi(bi) != 0) && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && large_bar &&
                hipExtMallocWithFlags((void**)&e->d_bar, 32768, hipDeviceMallocFinegrained) == hipSuccess) {
                ENG_TRY(hipMemset(e->d_bar, 0, 32768));
                e->bar_io = true;
            } else {
                (void)hipGetLastError();
                e->d_bar = nullptr;
            }
        }
        int rc = sf_create(e);
        if (rc) {
            mc_destroy(e);
            return rc;
        }
        *out = e;
        return MC_OK;
    }
    ENG_TRY(hipMalloc(&e->d_fdl, sizeof(float4) * (size_t)MC_NB * e->ring));
    if (e->half) ENG_TRY(hipMalloc(&e->d_fdl16, sizeof(uint2) * (size_t)MC_NB * e->ring));
    ENG_TRY(hipMalloc(&e->d_slotgain, sizeof(float4) * (size_t)MC_MAXV * e->ring));
    // >= 8 planes of 256 blocks; the fast-FIR form writes three half-rate sequences (1.5 x the blocks, + a tile each)
    ENG_TRY(hipMalloc(&e->d_Y, sizeof(float4) * (size_t)MC_NB * y_capacity(e)));
    if (!e->half) ENG_TRY(hipMalloc(&e->d_Yc, sizeof(float4) * (size_t)MC_NB * e->Tcap));
    ENG_TRY(hipMalloc(&e->d_tail, sizeof(float4) * (size_t)8 * MC_NB * e->nchunk * MC_MAXV));
    e->d_Ybuf[0] = e->d_Y;
    e->d_Ycbuf[0] = e->d_Yc;
    e->d_tailbuf[0] = e->d_tail;
    if (e->pipelined) {
        ENG_TRY(hipStreamCreateWithFlags(&e->post_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            ENG_TRY(hipEventCreateWithFlags(&e->ev_mac[i][0], hipEventDisableTiming));
            ENG_TRY(hipEventCreateWithFlags(&e->ev_mac[i][1], hipEventDisableTiming));
            ENG_TRY(hipEventCreateWithFlags(&e->ev_corr[i], hipEventDisableTiming));
            ENG_TRY(hipEventCreateWithFlags(&e->ev_post[i], hipEventDisableTiming));
        }
        ENG_TRY(hipMalloc(&e->d_Ybuf[1], sizeof(float4) * (size_t)MC_NB * y_capacity(e)));
        ENG_TRY(hipMalloc(&e->d_Ycbuf[1], sizeof(float4) * (size_t)MC_NB * e->Tcap));
        ENG_TRY(hipMalloc(&e->d_tailbuf[1], sizeof(float4) * (size_t)8 * MC_NB * e->nchunk * MC_MAXV));
    }
    ENG_TRY(hipMalloc(&e->d_part, sizeof(float4) * (size_t)e->Tstream * MC_NB * e->nchunk * MC_MAXV));
    e->d_partbuf[0] = e->d_part;
    if (cfg->pipeline != 0 && cfg->precision == 0)
        ENG_TRY(hipMalloc(&e->d_partbuf[1], sizeof(float4) * (size_t)e->Tstream * MC_NB * e->nchunk * MC_MAXV));
    ENG_TRY(hipMalloc(&e->d_sums, sizeof(float4) * (size_t)e->Tmax * kPipe));
    ENG_TRY(hipMalloc(&e->d_seg, sizeof(float) * (size_t)e->sr * 2 * FFT_N));
    ENG_TRY(hipMalloc(&e->d_wet, sizeof(float) * 2 * (size_t)e->wr));
    ENG_TRY(hipMalloc(&e->d_cring, sizeof(double) * 4 * (size_t)e->rc));
    ENG_TRY(hipMalloc(&e->d_ctot, sizeof(double) * 4 * (size_t)((e->Tmax + 255) / 256 + 1)));
    ENG_TRY(hipMalloc(&e->d_cflag, sizeof(unsigned) * (size_t)((e->Tmax + 255) / 256 + 2)));
    ENG_TRY(hipMemset(e->d_cflag, 0, sizeof(unsigned) * (size_t)((e->Tmax + 255) / 256 + 2)));
    e->d_cticket = e->d_cflag + (e->Tmax + 255) / 256 + 1;
    e->rr = (int)next_pow2(cfg->n_ref);
    ENG_TRY(hipMalloc(&e->d_res_mac, sizeof(float) * 2 * (size_t)e->rr));
    ENG_TRY(hipMalloc(&e->d_res_fix, sizeof(float) * 2 * (size_t)e->rr));
    e->xr = (int)next_pow2(cfg->n_ref + (uint64_t)e->Tmax * MC_B * (e->pipelined ? 2 : 1) + MC_MAX_PREDELAY + 1024);
    ENG_TRY(hipMalloc(&e->d_xhist, sizeof(float) * 2 * (size_t)e->xr));
    ENG_TRY(hipMalloc(&e->d_gring, sizeof(float4) * (size_t)MC_MAXV * e->rc));
    ENG_TRY(hipMalloc(&e->d_ptab, sizeof(BlockParams) * (size_t)e->Tmax * kPipe));
    ENG_TRY(hipMalloc(&e->d_tw, sizeof(float2) * FFT_N));
    e->Thost = std::min(e->Tmax, 16384);  // host-buffer calls with pageable buffers stage through pinned memory in chunks of this
    for (int i = 0; i < 4; i++) ENG_TRY(hipMalloc(&e->d_io[i], sizeof(float) * (size_t)e->Thost * MC_B));
    ENG_TRY(hipHostMalloc(&e->h_io, sizeof(float) * 4 * (size_t)e->Thost * MC_B, hipHostMallocMapped));
    ENG_TRY(hipHostGetDevicePointer((void**)&e->hd_io, e->h_io, 0));
    ENG_TRY(hipMalloc(&e->d_tailform, 2 * sizeof(unsigned)));
    ENG_TRY(hipMemset(e->d_tailform, 0, 2 * sizeof(unsigned)));
    if (const char* f = LAB_ENV("MCCONV_TAIL_FORM")) e->tail_form = !std::strcmp(f, "td") ? 1 : !std::strcmp(f, "fd") ? 2 : 0;
    ENG_TRY(hipHostMalloc(&e->h_flag, 256, hipHostMallocMapped));  // completion word, doorbell, "gave up" word: a line each
    ENG_TRY(hipHostGetDevicePointer((void**)&e->hd_flag, e->h_flag, 0));
    std::memset(e->h_flag, 0, 256);
    e->h_bell = reinterpret_cast<unsigned long long*>(e->h_flag + 16);
    e->hd_bell = reinterpret_cast<unsigned long long*>(e->hd_flag + 16);
    e->h_exited = e->h_flag + 32;
    e->hd_exited = e->hd_flag + 32;
    if (std::getenv("MCCONV_NO_PARK")) e->park = false;
    if (const char* ho = LAB_ENV("MCCONV_HOST_OUT_DIRECT")) e->host_out_direct = std::atoi(ho) != 0;
    {
        int large_bar = 0;
        const char* bi = std::getenv("MCCONV_BAR_IO");
        if ((!bi || std::atoi(bi) != 0) && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && large_bar &&
            hipExtMallocWithFlags((void**)&e->d_bar, 32768, hipDeviceMallocFinegrained) == hipSuccess) {
            ENG_TRY(hipMemset(e->d_bar, 0, 32768));
            e->bar_io = true;
        } else {
            (void)hipGetLastError();
            e->d_bar = nullptr;
        }
    }
    if (const char* pm = std::getenv("MCCONV_PARK_MS")) e->park_ticks = (unsigned long long)std::max(1, std::atoi(pm)) * 100000ull;
    ENG_TRY(hipMalloc(&e->d_done_ctr, sizeof(unsigned)));
    ENG_TRY(hipMemset(e->d_done_ctr, 0, sizeof(unsigned)));
    if (std::getenv("MCCONV_NO_SPIN")) e->spin_wait = false;
    if (e->bar_io && e->spin_wait && !(LAB_ENV("MCCONV_TAGGED_IO") && std::atoi(LAB_ENV("MCCONV_TAGGED_IO")) == 0)) {
        ENG_TRY(hipHostMalloc(&e->h_gran, sizeof(unsigned long long) * 2 * 4 * MC_B, hipHostMallocMapped));  // (room for a 1024-frame period)
        ENG_TRY(hipHostGetDevicePointer((void**)&e->hd_gran, e->h_gran, 0));
        std::memset(e->h_gran, 0, sizeof(unsigned long long) * 2 * 4 * MC_B);
        e->tio = true;
        e->tio_long = LAB_ENV("MCCONV_TAGGED_IO") && std::atoi(LAB_ENV("MCCONV_TAGGED_IO")) == 2;
    }
    for (int i = 0; i < kStageBufs; i++) {
        ENG_TRY(hipHostMalloc(&e->h_ptab[i], sizeof(BlockParams) * (size_t)e->Tmax, hipHostMallocDefault));
        ENG_TRY(hipEventCreateWithFlags(&e->ptab_ev[i], hipEventDisableTiming));
    }
    ENG_TRY(hipEventCreateWithFlags(&e->ev_tail, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) ENG_TRY(hipMalloc(&e->d_part_jack[i], sizeof(float4) * (size_t)MC_NB * e->nchunk * MC_MAXV));
    for (int i = 0; i < 2; i++) ENG_TRY(hipMalloc(&e->d_drop[i], sizeof(float) * 2 * 4 * MC_B));
    if (LAB_ENV("MCCONV_NO_SPECULATE")) e->speculate = false;
    if (const char* f2 = std::getenv("MCCONV_FFT2")) e->fft2 = std::atoi(f2) != 0;
    if (const char* g2 = std::getenv("MCCONV_FFT2_FUSED")) e->fft2_fused = std::atoi(g2) != 0;
    {
        int cus = 0;
        if (const char* gw = LAB_ENV("MCCONV_G2_WIDE")) e->g2_wide = std::atoi(gw) != 0;
        // k_g2_mac: one workgroup per (bin, chunk) item - the dispatcher keeps two resident per CU and hands a CU its next
        // one the moment a slot frees (measured against 512 persistent workgroups striding over 1280 items: 98 vs 108 us).
        // The one-workgroup-per-CU form is persistent (one per CU, look-ahead into its next item).
        e->g2_grid = 1 << 30;
        if (e->g2_wide) e->g2_grid = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus >= 8) ? (cus & ~7) : 256;
    }
    // (any grid >= 1 is correct: a workgroup strides over the items; multiples of 8 keep a bin's chunks on one XCD)
    if (const char* gg = LAB_ENV("MCCONV_G2_GRID")) e->g2_grid = std::max(1, std::atoi(gg));
    if (const char* gd = LAB_ENV("MCCONV_G2_DUO")) e->g2_duo = std::atoi(gd) != 0;
    if (const char* gd = LAB_ENV("MCCONV_G2_DUO_MINCH")) e->g2_duo_minch = std::max(1, std::atoi(gd));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) == hipSuccess && cus >= 8) e->g2_duo_grid = std::min(cus & ~7, 256);
    }
    if (const char* gd = LAB_ENV("MCCONV_G2_DUO_GRID")) e->g2_duo_grid = std::max(8, std::atoi(gd));
    if (LAB_ENV("MCCONV_DEBUG_ADDR")) e->debug_addr = true;
    if (const char* cr = LAB_ENV("MCCONV_CORR_RIDE")) e->corr_ride = std::atoi(cr) != 0;
    if (const char* tm = LAB_ENV("MCCONV_FFT2_WORK")) e->fft2_work = std::max<int64_t>(1, std::atoll(tm));
    if (const char* gm = LAB_ENV("MCCONV_G2_PMIN")) e->g2_pmin = std::max(16, std::atoi(gm));
    if (const char* gp = LAB_ENV("MCCONV_G2_PMAX")) e->g2_pmax = std::max(256, std::min(G2_N / 2 + 2048, std::atoi(gp)));
    if (const char* fo = LAB_ENV("MCCONV_FUSE_OUT")) e->fuse_out = std::atoi(fo) != 0;
    if (const char* fo = LAB_ENV("MCCONV_FUSE_DROP")) e->fuse_drop = std::atoi(fo) != 0;
    if (const char* fo = LAB_ENV("MCCONV_DROP_AHEAD")) e->drop_ahead = std::atoi(fo) != 0;
    if (const char* fo = LAB_ENV("MCCONV_CARRY_DROP")) e->carry_drop = std::atoi(fo) != 0;
    if (const char* iw = LAB_ENV("MCCONV_INV_WET")) e->inv_to_wet = std::atoi(iw) != 0;
    if (const char* os = std::getenv("MCCONV_OS")) e->os_on = std::atoi(os) != 0;
    if (const char* os = LAB_ENV("MCCONV_OS_SIDE")) e->os_side = std::atoi(os) != 0;
    if (const char* os = LAB_ENV("MCCONV_OS_MIN")) e->os_min_blocks = std::max(1, std::atoi(os));
    if (const char* fl = std::getenv("MCCONV_FFA_LEVELS")) e->ffa_levels = std::max(0, std::min(3, std::atoi(fl)));
    {
        std::vector<float2> tw;
        host_twiddles(tw);
        ENG_TRY(hipMemcpy(e->d_tw, tw.data(), sizeof(float2) * FFT_N, hipMemcpyHostToDevice));
    }
    {
        int rc = zero_state(e);
        if (rc) {
            mc_destroy(e);
            return rc;
        }
    }
#undef ENG_TRY
    *out = e;
    return MC_OK;
}

void mc_destroy(mc_engine* e) {
    if (!e) return;
#ifdef MC_JACK_TRACE
    if (e->tr_n > 100) {
        const double n = (double)(e->tr_n - 100);
        fprintf(stderr, "mcconv JACK trace over %ld periods (us): entry -> launches issued %.2f, -> flag seen %.2f, -> return %.2f; k_tail1 start -> output issued %.2f (%.0f shader clocks; first barrier %.0f, sums done %.0f, second barrier %.0f), -> end %.2f, "
                        "end of the previous k_tail1 -> start of this one %.2f\n", e->tr_n - 100, e->tr_launch / n, e->tr_flag / n, e->tr_total / n, e->tr_out / n, e->tr_clk / n, e->tr_c[0] / n, e->tr_c[1] / n, e->tr_c[2] / n, e->tr_kernel / n, e->tr_gap / n);
    }
#endif
    (void)hipSetDevice(e->device);
    unpark(e);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    sf_free(e);
    for (int i = 0; i < kMaxIrs + kMixIrs; i++)
        if (e->irs[i].d_H) (void)hipFree(e->irs[i].d_H);
    for (int i = 0; i < kMaxIrs + kMixIrs; i++)
        for (int l = 0; l < 3; l++)
            if (e->irs[i].d_Hp[l]) (void)hipFree(e->irs[i].d_Hp[l]);
    for (int i = 0; i < kMaxIrs + kMixIrs; i++)
        if (e->irs[i].d_H2) (void)hipFree(e->irs[i].d_H2);
    for (int i = 0; i < kMaxIrs + kMixIrs; i++)
        if (e->irs[i].d_G2) (void)hipFree(e->irs[i].d_G2);
    for (int i = 0; i < kMaxIrs + kMixIrs; i++)
        if (e->irs[i].d_Htail) (void)hipFree(e->irs[i].d_Htail);
    for (int i = 0; i < kMaxIrs + kMixIrs; i++)
        if (e->irs[i].d_h) (void)hipFree(e->irs[i].d_h);
    for (int i = 0; i < kMaxIrs + kMixIrs; i++)
        if (e->irs[i].d_H16) (void)hipFree(e->irs[i].d_H16);
    (void)hipFree(e->d_fdl);
    (void)hipFree(e->d_slotgain);
    (void)hipFree(e->d_fdl16);
    (void)hipFree(e->d_stash);
    if (e->post_stream) (void)hipStreamSynchronize(e->post_stream);
    for (int i = 0; i < 2; i++) {
        (void)hipFree(e->d_Ybuf[i]);
        (void)hipFree(e->d_Ycbuf[i]);
        (void)hipFree(e->d_tailbuf[i]);
        if (e->post_stream) {
            (void)hipEventDestroy(e->ev_mac[i][0]);
            (void)hipEventDestroy(e->ev_mac[i][1]);
            (void)hipEventDestroy(e->ev_corr[i]);
            (void)hipEventDestroy(e->ev_post[i]);
        }
    }
    if (e->post_stream) (void)hipStreamDestroy(e->post_stream);
    (void)hipFree(e->d_partbuf[0]);
    (void)hipFree(e->d_partbuf[1]);
    (void)hipFree(e->d_sums);
    (void)hipFree(e->d_seg);
    (void)hipFree(e->d_wet);
    (void)hipFree(e->d_cring);
    (void)hipFree(e->d_ctot);
    (void)hipFree(e->d_cflag);
    (void)hipFree(e->d_stamps);
    (void)hipFree(e->d_done_ctr);
    (void)hipFree(e->d_res_mac);
    (void)hipFree(e->d_res_fix);
    (void)hipFree(e->d_xhist);
    (void)hipFree(e->d_os_T);
    (void)hipFree(e->d_os_part);
    (void)hipFree(e->d_os_SP);
    (void)hipFree(e->d_os_SP0);
    (void)hipFree(e->d_os_planes);
    if (e->os_stream) {
        (void)hipStreamSynchronize(e->os_stream);
        for (int i = 0; i < 3; i++) (void)hipEventDestroy(e->os_ev[i]);
        (void)hipStreamDestroy(e->os_stream);
    }
    (void)hipFree(e->d_gring);
    (void)hipFree(e->d_ptab);
    (void)hipFree(e->d_tw);
    for (int i = 0; i < 4; i++) (void)hipFree(e->d_io[i]);
    for (int b = 0; b < 3; b++) {
        for (int i = 0; i < 4; i++) (void)hipFree(e->d_pio[b][i]);
        if (e->ev_h2d[b]) (void)hipEventDestroy(e->ev_h2d[b]);
        if (e->ev_comp[b]) (void)hipEventDestroy(e->ev_comp[b]);
        if (e->ev_d2h[b]) (void)hipEventDestroy(e->ev_d2h[b]);
    }
    if (e->h2d_stream) (void)hipStreamDestroy(e->h2d_stream);
    if (e->d2h_stream) (void)hipStreamDestroy(e->d2h_stream);
    if (e->d_bar) (void)hipFree(e->d_bar);
    if (e->h_gran) (void)hipHostFree(e->h_gran);
    if (e->h_io) (void)hipHostFree(e->h_io);
    if (e->d_tailform) (void)hipFree(e->d_tailform);
    if (e->h_flag) (void)hipHostFree(e->h_flag);
    for (int i = 0; i < kStageBufs; i++) {
        if (e->h_ptab[i]) {
            (void)hipHostFree(e->h_ptab[i]);
            (void)hipEventDestroy(e->ptab_ev[i]);
        }
    }
    if (e->ev_tail) (void)hipEventDestroy(e->ev_tail);
    (void)hipFree(e->d_part_jack[0]);
    (void)hipFree(e->d_part_jack[1]);
    (void)hipFree(e->d_dropbuf);
    (void)hipFree(e->d_drop[0]);
    (void)hipFree(e->d_drop[1]);
    if (e->kev_created)
        for (int i = 0; i < kEvPool; i++) {
            (void)hipEventDestroy(e->kev[i][0]);
            (void)hipEventDestroy(e->kev[i][1]);
        }
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int mc_reset(mc_engine* e) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    return e->sf ? sf_zero(e) : zero_state(e);
}

int mc_set_period(mc_engine* e, uint32_t nframes) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    if (nframes != 256 && nframes != 512 && nframes != 1024) return fail(MC_ERR_ARG, "period must be 256, 512 or 1024 frames");
    const int pm = (int)(nframes / MC_BLOCK);
    if (e->Tmax % pm) return fail(MC_ERR_ARG, "max_batch %d is not a multiple of %d blocks", e->Tmax, pm);
    if (pm == e->pm) return MC_OK;
    HIP_TRY(hipSetDevice(e->device));
    e->pm = pm;
    return e->sf ? sf_zero(e) : zero_state(e);  // the per-call semantics change: start from the cold state
}

int mc_load_ir(mc_engine* e, uint64_t idx, const float* lr, uint64_t frames, uint64_t nframes) {
    // Convolution::prepare, conv.cu:207-253
    if (!e || !lr) return fail(MC_ERR_ARG, "null argument");
    if (idx >= (uint64_t)kMaxIrs) return fail(MC_ERR_ARG, "IR index %llu >= %d", (unsigned long long)idx, kMaxIrs);
    if (nframes >= e->cfg.n_ref) return fail(MC_ERR_ARG, "nframes >= n_ref");
    if (frames == 0) return fail(MC_ERR_ARG, "empty IR");
    HIP_TRY(hipSetDevice(e->device));
    if (e->sf) return sf_load_ir(e, idx, lr, frames, nframes);
    const uint64_t n = std::min<uint64_t>(frames, e->cfg.n_ref - nframes);  // conv.cu:239
    const int P = (int)((n + MC_B - 1) / MC_B);
    if (P > e->Pcap) return fail(MC_ERR_ARG, "IR needs %d partitions, engine capacity is %d", P, e->Pcap);
    IrEntry& ir = e->irs[idx];
    {
        int rc = drain_post(e);
        if (!rc) rc = leave_jack_path(e);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!ir.d_H) HIP_TRY(hipMalloc(&ir.d_H, sizeof(float4) * (size_t)MC_NB * e->Pstride));
    float* d_lr = nullptr;
    HIP_TRY(hipMalloc(&d_lr, sizeof(float) * 2 * n));
    hipError_t er = hipMemcpy(d_lr, lr, sizeof(float) * 2 * n, hipMemcpyHostToDevice);
    if (er == hipSuccess) er = hipMemsetAsync(ir.d_H, 0, sizeof(float4) * (size_t)MC_NB * e->Pstride, e->stream);
    if (er == hipSuccess) {
        hipLaunchKernelGGL(k_fwd<false>, dim3((P + FWD_TILE - 1) / FWD_TILE), dim3(XF_THREADS), 0, e->stream, d_lr, d_lr + 1, 2, (int64_t)n, P,
                           ir.d_H, e->Pstride, 0, (const BlockParams*)nullptr, 0, (float4*)nullptr, (float4*)nullptr, e->d_tw,
                           (uint2*)nullptr, (float*)nullptr, 0, (float4*)nullptr, 0, (int64_t)0, 0, P, P, 0, 0, DropAhead(), 0);
        er = hipGetLastError();
    }
    if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
    if (er != hipSuccess) {
        (void)hipFree(d_lr);
        return fail(MC_ERR_HIP, "IR preparation failed: %s", hipGetErrorString(er));
    }
    if (ir.d_h) (void)hipFree(ir.d_h);
    if (ir.d_Htail) {
        (void)hipFree(ir.d_Htail);
        ir.d_Htail = nullptr;
    }
    ir.d_h = reinterpret_cast<float2*>(d_lr);  // the truncated taps stay on the device for the Q8 pass
    invalidate_derived(ir);
    e->ir_gen++;
    for (int l = 0; l < 3; l++)  // (a reloaded IR gives its fast-FIR components back; the stream is idle here)
        if (ir.d_Hp[l]) {
            (void)hipFree(ir.d_Hp[l]);
            ir.d_Hp[l] = nullptr;
        }
    if (e->half) {
        // scaled fp16 copy: one power-of-two scale per IR puts the largest bin near 2^13 (half max 65504);
        // a 60 dB decay then still sits ~2^3 above the smallest normal half
        const size_t nel = (size_t)MC_NB * e->Pstride;
        std::vector<float> host(nel * 4);
        HIP_TRY(hipMemcpy(host.data(), ir.d_H, sizeof(float4) * nel, hipMemcpyDeviceToHost));
        float mx = 0.f;
        for (float v : host) mx = std::max(mx, std::fabs(v));
        int ex = 0;
        if (mx > 0.f) std::frexp(mx, &ex);
        ir.scale16 = std::ldexp(1.0f, 13 - ex);
        if (!ir.d_H16) HIP_TRY(hipMalloc(&ir.d_H16, sizeof(uint2) * nel));
        hipLaunchKernelGGL(k_to_half, dim3(1024), dim3(256), 0, e->stream, ir.d_H, ir.d_H16, nel, ir.scale16);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    double s[4] = {0, 0, 0, 0};
    for (uint64_t m = 0; m < n; m++) {
        const double sg = (m & 1) ? -1.0 : 1.0;
        s[0] += lr[2 * m];
        s[1] += lr[2 * m + 1];
        s[2] += sg * lr[2 * m];
        s[3] += sg * lr[2 * m + 1];
    }
    std::memcpy(ir.sums, s, sizeof(s));
    ir.taps = n;
    ir.P = P;
    if ((int)idx + 1 > e->nirs) e->nirs = (int)idx + 1;
    e->spec_valid = e->dspec.valid = false;
    e->uniform_valid[0] = e->uniform_valid[1] = false;
    return MC_OK;
}

int mc_num_irs(const mc_engine* e) { return e ? e->nirs : 0; }

int mc_ir_info(const mc_engine* e, uint64_t idx, double out[6]) {
    if (!e || !out || idx >= (uint64_t)kMaxIrs || !(e->irs[idx].d_H || e->irs[idx].d_S)) return fail(MC_ERR_ARG, "IR %llu not loaded", (unsigned long long)idx);
    for (int i = 0; i < 4; i++) out[i] = e->irs[idx].sums[i];
    out[4] = (double)e->irs[idx].taps;
    out[5] = (double)e->irs[idx].P;
    return MC_OK;
}

int mc_set_params(mc_engine* e, int half, const mc_cc_value* v) {
    if (!e || !v || half < 0 || half > 1) return fail(MC_ERR_ARG, "bad argument");
    e->ph.update([&](mc_cc_value (&cc)[2]) { cc[half] = *v; });
    return MC_OK;
}

int mc_get_params(const mc_engine* e, int half, mc_cc_value* v) {
    if (!e || !v || half < 0 || half > 1) return fail(MC_ERR_ARG, "bad argument");
    mc_cc_value cc[2];
    e->ph.sample(cc);
    *v = cc[half];
    return MC_OK;
}

int mc_handle_cc(mc_engine* e, int half, const uint8_t ccmap[8], uint8_t m2, int val) {
    // handleCC, conv.cu:255-276
    if (!e || !ccmap || half < 0 || half > 1) return fail(MC_ERR_ARG, "bad argument");
    const uint64_t nb = (uint64_t)e->nirs;
    e->ph.update([&](mc_cc_value (&cc)[2]) {
        mc_cc_value& v = cc[half];
        if (ccmap[0] == m2) {
            v.select = (uint64_t)val * nb / 0x80;
            v.vsteps = v.speed;
        }
        if (ccmap[1] == m2) v.predelay = (uint64_t)val * MC_MAX_PREDELAY / 0x80;
        if (ccmap[2] == m2) v.dry = val / 128.0f;
        if (ccmap[3] == m2) v.wet = val / 128.0f;
        if (ccmap[5] == m2) v.panDry = val / 64.0f - 1;
        if (ccmap[6] == m2) v.panWet = val / 64.0f - 1;
        if (ccmap[7] == m2) v.level = val / 128.0f;
        if (ccmap[4] == m2) {
            v.speed = ((uint64_t)val * MC_MAX_SPEED) / 0x80;
            if (v.vsteps > v.speed) v.vsteps = v.speed;
        }
    });
    return MC_OK;
}

int mc_process(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR, uint64_t nframes) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    if (nframes != (uint64_t)MC_BLOCK * e->pm)
        return fail(MC_ERR_ARG, "nframes must be %d (got %llu)", MC_BLOCK * e->pm, (unsigned long long)nframes);
    HIP_TRY(hipSetDevice(e->device));
    // the reference brackets its GPU work with events (conv.cu:299-302, 454-462); the call below
    // returns only when the output is on the host, so a host clock around it measures a superset
    const auto t0 = std::chrono::steady_clock::now();
    // 256-frame periods take the fused single-block path; 512 / 1024 run as one small zero-copy batch
    // (partition shards keep the batch kernels: their low partitions are not the IR's first ones)
    const bool whole_ir = e->cfg.part_begin == 0 && e->cfg.part_end == 0;
    int rc = e->sf ? sf_process(e, in1, in2, outL, outR) : e->pm == 1 ? process_one(e, in1, in2, outL, outR)
                        : (whole_ir ? process_period_fused(e, in1, in2, outL, outR) : process_period(e, in1, in2, outL, outR));
    if (rc) return rc;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (++e->nruns > 0) e->runtime_ms += ms;  // first 10 calls discarded, conv.h:80
    return MC_OK;
}

void* mc_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(MC_ERR_NOMEM, "mc_host_alloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void mc_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int mc_process_batch(mc_engine* e, const float* in1, const float* in2, float* outL, float* outR, uint64_t nblocks) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    if (e->sf) return sf_batch_host(e, in1, in2, outL, outR, (int)std::min<uint64_t>(nblocks, 1u << 30));
    return process_host(e, in1, in2, outL, outR, (int)std::min<uint64_t>(nblocks, 1u << 30));
}

int mc_process_batch_device(mc_engine* e, const float* d_in1, const float* d_in2, float* d_outL, float* d_outR, uint64_t nblocks) {
    if (!e || !d_in1 || !d_in2 || !d_outL || !d_outR) return fail(MC_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const int T = (int)std::min<uint64_t>(nblocks, 1u << 30);
    if (e->sf) return sf_batch_device(e, d_in1, d_in2, d_outL, d_outR, T);
    int rc = run_front(e, d_in1, d_in2, T, nullptr, 0, T, d_outL, d_outR);
    if (rc) return rc;
    return run_back(e, d_in1, d_in2, nullptr, d_outL, d_outR, T);
}

int mc_process_batch_slice_device(mc_engine* e, const float* d_in1, const float* d_in2, float* d_outL, float* d_outR,
                                  uint64_t nblocks, uint64_t first, uint64_t count) {
    if (!e || !d_in1 || !d_in2 || !d_outL || !d_outR) return fail(MC_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const int T = (int)std::min<uint64_t>(nblocks, 1u << 30);
    if (e->sf) return fail(MC_ERR_ARG, "the single-transform form has no block slices");
    if (first > (uint64_t)T || count > (uint64_t)T) return fail(MC_ERR_ARG, "slice outside the batch");
    int rc = run_front(e, d_in1, d_in2, T, nullptr, (int)first, (int)count, d_outL, d_outR);
    if (rc) return rc;
    return run_back(e, d_in1, d_in2, nullptr, d_outL, d_outR, T);
}

int mc_partial_batch_device(mc_engine* e, const float* d_in1, const float* d_in2, float* d_partial, uint64_t nblocks) {
    if (!e || !d_in1 || !d_in2 || !d_partial) return fail(MC_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const int T = (int)std::min<uint64_t>(nblocks, 1u << 30);
    if (e->sf) return fail(MC_ERR_ARG, "the single-transform form has no partition shards");
    return run_front(e, d_in1, d_in2, T, d_partial, 0, T);
}

int mc_finish_batch_device(mc_engine* e, const float* d_in1, const float* d_in2, const float* d_wet_sum, float* d_outL,
                           float* d_outR, uint64_t nblocks) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    if (e->sf) return fail(MC_ERR_ARG, "the single-transform form has no partition shards");
    const bool want_out = d_outL || d_outR || d_wet_sum;
    if (want_out && (!d_in1 || !d_in2 || !d_wet_sum || !d_outL || !d_outR)) return fail(MC_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    const int T = (int)std::min<uint64_t>(nblocks, 1u << 30);
    if (T <= 0 || T > e->Tmax) return fail(MC_ERR_ARG, "nblocks outside range");
    return run_back(e, d_in1, d_in2, d_wet_sum, want_out ? d_outL : nullptr, d_outR, T);
}

int mc_finish_batch_slice_device(mc_engine* e, const float* d_in1, const float* d_in2, const float* d_wet_sum_slice, float* d_outL,
                                 float* d_outR, uint64_t nblocks, uint64_t first, uint64_t count) {
    if (!e || !d_in1 || !d_in2 || !d_wet_sum_slice || !d_outL || !d_outR) return fail(MC_ERR_ARG, "null argument");
    if (e->sf) return fail(MC_ERR_ARG, "the single-transform form has no partition shards");
    HIP_TRY(hipSetDevice(e->device));
    const int T = (int)std::min<uint64_t>(nblocks, 1u << 30);
    if (T <= 0 || T > e->Tmax) return fail(MC_ERR_ARG, "nblocks outside range");
    if (first >= (uint64_t)T || count == 0 || count > (uint64_t)T - first) return fail(MC_ERR_ARG, "slice outside the batch");
    return run_back(e, d_in1, d_in2, d_wet_sum_slice, d_outL, d_outR, T, false, (int)first, (int)count);
}

int mc_sync(mc_engine* e) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    int rc = drain_post(e);
    if (!rc) rc = leave_jack_path(e);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return MC_OK;
}

int mc_fence(mc_engine* e) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    return fence_post(e);
}

int mc_fence_older(mc_engine* e) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->pipelined || e->batch_seq == 0) return MC_OK;
    const int older = (int)(e->batch_seq & 1);  // the most recent batch had parity (batch_seq - 1) & 1
    if (e->post_pending[older]) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_post[older], 0));
    return MC_OK;
}

int mc_set_stream(mc_engine* e, void* s) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    {
        int rc = drain_post(e);
        if (!rc) rc = leave_jack_path(e);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stream = s ? (hipStream_t)s : e->own_stream;
    return MC_OK;
}

void* mc_get_stream(mc_engine* e) { return e ? (void*)e->stream : nullptr; }

double mc_avg_runtime_ms(const mc_engine* e) { return (e && e->nruns > 0) ? e->runtime_ms / e->nruns : 0.0; }  // conv.h:61

int mc_enable_kernel_timing(mc_engine* e, int on) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    if (on && !e->kev_created) {
        for (int i = 0; i < kEvPool; i++) {
            HIP_TRY(hipEventCreate(&e->kev[i][0]));
            HIP_TRY(hipEventCreate(&e->kev[i][1]));
            e->kev_stamped[i] = false;
        }
        HIP_TRY(hipMalloc(&e->d_stamps, sizeof(unsigned long long) * 2 * MC_STAMP_WGS * kStampSlots));
        HIP_TRY(reset_stamps(e));
        e->kev_created = true;
    }
    if (!on) {
        int rc = drain_kernel_events(e);
        if (rc) return rc;
    }
    e->ktiming = on != 0;
    return MC_OK;
}

int mc_get_kernel_stats(mc_engine* e, mc_kernel_stats* out, int reset) {
    if (!e || !out) return fail(MC_ERR_ARG, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    int rc = drain_kernel_events(e);
    if (rc) return rc;
    *out = e->ks;
    if (reset) {
        uint32_t res = e->ks.resident, parts = e->ks.partitions, lv = e->ks.fast_levels;
        std::memset(&e->ks, 0, sizeof(e->ks));
        e->ks.resident = res;
        e->ks.partitions = parts;
        e->ks.fast_levels = lv;
    }
    return MC_OK;
}

uint64_t mc_algorithmic_bytes_per_block(const mc_engine* e) {
    // 4 IR paths + 2 delay-line inputs, P partitions, 256 bins x 8 B
    if (!e) return 0;
    if (e->sf) return 24ull * e->cfg.n_ref;  // four half spectra (4 N/2 x 8 B) + the input window (2 N x 4 B), per call
    mc_cc_value cc[2];
    e->ph.sample(cc);
    const IrEntry& a = e->irs[cc[0].select % kMaxIrs];
    const IrEntry& b = e->irs[cc[1].select % kMaxIrs];
    int P = std::max(a.P, b.P);
    if (e->cfg.part_end) P = std::max(0, std::min<int>(P, (int)e->cfg.part_end) - (int)e->cfg.part_begin);
    return (uint64_t)(4 + 2) * (uint64_t)P * MC_NB * (e->half ? 4ull : 8ull);
}

uint64_t mc_blocks_processed(const mc_engine* e) { return e ? e->t_abs : 0; }

uint64_t mc_preferred_batch(const mc_engine* e, uint64_t at_most) {
    if (!e) return 0;
    at_most = std::min<uint64_t>(at_most, (uint64_t)e->Tmax);
    if (e->sf) return at_most - at_most % (uint64_t)e->pm;
    int pmax = 0;
    for (int i = 0; i < kMaxIrs; i++)
        if (e->irs[i].d_H) pmax = std::max(pmax, round_up(e->irs[i].P, 16));
    const bool shard = e->cfg.part_begin || e->cfg.part_end;
    if (shard) {  // taps of the shard's own run of partitions
        int pb, pe;
        partition_range(e, pmax, &pb, &pe);
        pmax = pe - pb;
    }
    uint64_t chunk = 0;
    // overlap-save segments of 16384 - P16 blocks (ossave.hip.h): whole segments waste nothing
    if (e->os_on && !e->half && !e->pipelined && !shard && pmax > 0 && (int64_t)pmax * MC_B <= OS_N / 2) {
        const uint64_t hopb = (uint64_t)(OS_N / MC_B - pmax);
        if (at_most >= hopb && at_most >= (uint64_t)e->os_min_blocks) return at_most / hopb * hopb;
    }
    if (e->fft2 && !e->half && (shard ? pmax >= 16 : pmax >= e->g2_pmin)) {
        if (e->fft2_fused && pmax <= e->g2_pmax) chunk = (uint64_t)(G2_N - pmax + 1);
        else if (pmax <= F2_N / 2) chunk = (uint64_t)(F2_N - pmax + 1);
    }
    if (!chunk || at_most < chunk) return at_most >= 8 ? (at_most & ~(uint64_t)7) : at_most;
    return ((at_most + 1) / chunk * chunk - 1) & ~(uint64_t)7;
}

int mc_debug_read(mc_engine* e, int which, uint64_t idx, void* dst, uint64_t off, uint64_t bytes, uint64_t dims[4]) {
    if (!e) return fail(MC_ERR_ARG, "null engine");
    HIP_TRY(hipSetDevice(e->device));
    if (e->sf) {  // single-transform form: 0 = an IR's spectra [H_L | H_R] (float2, n_ref / 2 bins each, bin d + (n_ref / 512) c at [d][c]), 4 = the accumulators [2][512][n_ref / 512]
        if (dims) dims[0] = dims[1] = dims[3] = e->cfg.n_ref, dims[2] = (uint64_t)e->Tmax;
        if (!dst || !bytes) return MC_OK;
        const char* src = nullptr;
        uint64_t cap = 0;
        if (which == 0 && idx < (uint64_t)kMaxIrs && e->irs[idx].d_S) src = (const char*)e->irs[idx].d_S, cap = sizeof(float2) * e->cfg.n_ref;
        else if (which == 4) src = (const char*)e->sf->d_acc, cap = sizeof(float) * 2 * e->cfg.n_ref;
        else return fail(MC_ERR_ARG, "nothing to read for item %d in the single-transform form", which);
        if (off > cap || bytes > cap - off) return fail(MC_ERR_ARG, "read outside the buffer");
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipMemcpy(dst, src + off, bytes, hipMemcpyDeviceToHost));
        return MC_OK;
    }
    if (dims) {
        dims[0] = (uint64_t)e->Pstride;
        dims[1] = (uint64_t)e->ring;
        dims[2] = (uint64_t)e->Tmax;
        dims[3] = (uint64_t)e->wr;
    }
    if (!dst || !bytes) return MC_OK;
    if (which == 7 || which == 8) {  // generation of the parameter pair the last process call sampled (7) / published last (8): no stream access
        mc_cc_value cc[2];
        const uint64_t g = which == 7 ? e->last_gen : e->ph.sample(cc);
        if (off + bytes > sizeof(uint64_t)) return fail(MC_ERR_ARG, "read beyond the generation word");
        std::memcpy(dst, reinterpret_cast<const char*>(&g) + off, bytes);
        return MC_OK;
    }
    if (which == 10) {  // batches by the form their partition sums took {fused, split second-level transform, resident MAC}: no stream access
        if (off + bytes > sizeof(e->n_mac_form)) return fail(MC_ERR_ARG, "read beyond the counters");
        std::memcpy(dst, reinterpret_cast<const char*>(e->n_mac_form) + off, bytes);
        return MC_OK;
    }
    if (which == 15) {  // 1 = built with -DMCCONV_LAB (the measurement switches and the alternative kernels exist): no stream access
#ifdef MCCONV_LAB
        const uint64_t lab = 1;
#else
        const uint64_t lab = 0;
#endif
        if (off + bytes > sizeof(lab)) return fail(MC_ERR_ARG, "read beyond the word");
        std::memcpy(dst, reinterpret_cast<const char*>(&lab) + off, bytes);
        return MC_OK;
    }
    if (which == 11) {  // overlap-save form: {batches that took it, spectra builds}: no stream access
        if (off + bytes > sizeof(e->n_os)) return fail(MC_ERR_ARG, "read beyond the counters");
        std::memcpy(dst, reinterpret_cast<const char*>(e->n_os) + off, bytes);
        return MC_OK;
    }
    if (which == 9) {  // Q8 regime, batches: {cut terms summed by k_drop_fft for the whole batch, by the forward transforms (k_fwd<true>), in the time domain}: no stream access
        const uint64_t c[4] = {e->n_drop_fft, e->n_drop_ahead, e->n_drop_tiles, e->n_drop_carried};
        if (off + bytes > sizeof(c)) return fail(MC_ERR_ARG, "read beyond the counters");
        std::memcpy(dst, reinterpret_cast<const char*>(c) + off, bytes);
        return MC_OK;
    }
    if (which == 16) {  // 256-frame tails by the form their partition 0 took {frequency domain, time domain}: counted on the device, read behind the stream
        unsigned c[2] = {0, 0};
        if (off + bytes > sizeof(c)) return fail(MC_ERR_ARG, "read beyond the counters");
        if (e->d_tailform) {
            int rc = leave_jack_path(e);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(e->stream));
            HIP_TRY(hipMemcpy(c, e->d_tailform, sizeof(c), hipMemcpyDeviceToHost));
        }
        std::memcpy(dst, reinterpret_cast<const char*>(c) + off, bytes);
        return MC_OK;
    }
    if (which == 6) {  // host-side counters of the JACK path's parked periods {used, gave up on their own, told to give up}: no stream access
        const uint64_t c[3] = {e->n_park_hit, e->n_park_timeout, e->n_park_cancel};
        if (off + bytes > sizeof(c)) return fail(MC_ERR_ARG, "read beyond the counters");
        std::memcpy(dst, reinterpret_cast<const char*>(c) + off, bytes);
        return MC_OK;
    }
    const char* src = nullptr;
    uint64_t cap = 0;
    switch (which) {
        case 0:
            if (idx >= (uint64_t)kMaxIrs || !e->irs[idx].d_H) return fail(MC_ERR_ARG, "IR not loaded");
            src = (const char*)e->irs[idx].d_H;
            cap = sizeof(float4) * (uint64_t)MC_NB * e->Pstride;
            break;
        case 1: src = (const char*)e->d_fdl; cap = sizeof(float4) * (uint64_t)MC_NB * e->ring; break;
        case 2: src = (const char*)e->d_Y; cap = sizeof(float4) * (uint64_t)MC_NB * y_capacity(e); break;
        case 3: src = (const char*)e->d_seg; cap = sizeof(float) * (uint64_t)e->sr * 2 * FFT_N; break;
        case 4: src = (const char*)e->d_wet; cap = sizeof(float) * 2 * (uint64_t)e->wr; break;
        case 5: src = (const char*)e->d_cring; cap = sizeof(double) * 4 * (uint64_t)e->rc; break;
        case 12: src = (const char*)e->d_os_T; cap = sizeof(float4) * (uint64_t)OS_ITEMS * OS_N2 * e->os_T_segs; break;
        case 13: src = (const char*)e->d_os_SP; cap = e->d_os_SP ? sizeof(float4) * (uint64_t)OS_ITEMS * OS_N2 * 2 : 0; break;
        case 14: src = (const char*)e->d_os_SP0; cap = e->d_os_SP0 ? sizeof(float4) * 2 * (uint64_t)OS_N2 : 0; break;
        default: return fail(MC_ERR_ARG, "unknown buffer %d", which);
    }
    if (off + bytes > cap) return fail(MC_ERR_ARG, "read beyond buffer (%llu + %llu > %llu)", (unsigned long long)off, (unsigned long long)bytes, (unsigned long long)cap);
    {
        int rc = leave_jack_path(e);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(dst, src + off, bytes, hipMemcpyDeviceToHost));
    return MC_OK;
}

}  // extern "C"
