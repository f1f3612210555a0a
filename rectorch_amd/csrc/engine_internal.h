// engine_internal.h -- what the translation units of the engine share (engine.hip: life cycle, forward, the training step;
// engine_dp.hip: the data-parallel plan and its transports; engine_api.hip: options, stand-alone operators, instrumentation).
// Not part of the C ABI (include/rectorch_hip.h is).
#pragma once
#include "../../include/rectorch_hip.h"
#include "rtx_gemm.h"
#include "rtx_kernels.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <vector>

const char* rtx_last_error_str();



struct Layer {
    int in = 0, out = 0, inp = 0, outp = 0;
    bool tanh_act = false;
    void* Wsh = nullptr;      // the compute copy every reader of this step uses
    void* Wsh_alt = nullptr;  // the fused optimizer writes the NEXT step's copy here (they swap after the step), so the
                              // weight-gradient kernels may run beside the data-gradient chain that still reads Wsh
    void* WshT = nullptr;     // hidden layers, bf16: the transposed compute copy [inp][outp] the backward chain reads (small_layers.hip)
    void* A = nullptr;
    float* O32 = nullptr;
    void* D = nullptr;
};

struct TimingSite {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0;
    int launches = 0;
};

struct TempCsr {  // dense batch converted to CSR on the device
    int64_t* indptr = nullptr;
    int32_t* counts = nullptr;
    int32_t* indices = nullptr;
    float* values = nullptr;
    int64_t cap = 0;
};

// data-parallel step scheduled by the engine (rtx_engine_dp_attach): the exchange buffer every gradient is produced into, in
// comm dtype.  Layout (elements; every tensor starts at a multiple of 64): W[NL-1], b[NL-1], W[NL-2], b[NL-2], ..., W[1], b[1],
// b[0], W[0] -- bucket A (the decoder matrix, exchanged on the side stream beside the data-gradient chain) first, bucket B
// (everything else, behind the chain) after it, the sharded encoder matrix last so that the replicated tensors of a bucket
// form ONE contiguous all-reduce range.  A sharded matrix's region is rows padded to P(out) = roundup(out + 1, 128) (zeros):
// world equal row blocks.
struct DpState {
    bool on = false;
    rtx_dp_cfg cfg = {};
    rtx_dp_ops ops = {};                      // bucket B (and everything on ONE stream): the caller's stream
    rtx_dp_ops ops_side = {};                 // bucket A on the side stream: a communicator of its own when the plan brings one (ABI 7)
    bool two_comms = false;                   // ops_side is a different communicator / table than ops
    void* xg = nullptr;                       // exchange buffer, comm dtype
    size_t xbytes = 0, xesz = 4;
    size_t xoff[2 * 2 * RTX_MAX_LAYERS] = {};   // element offset of tensor t
    bool shard[2 * RTX_MAX_LAYERS] = {};        // per layer: weight matrix reduce-scattered / updated by rows / all-gathered
    bool broken = false;                      // a collective failed inside a group: no further step until the plan is attached again
    // what the LAST step's exchange moved, per rank (buffer bytes handed to the collectives; rtx_engine_get_option "dp_*")
    int64_t st_all_reduce = 0, st_reduce_scatter = 0, st_all_gather = 0;
    int st_collectives = 0;
    void* emu_scratch = nullptr;              // emulate: where the stand-in copies go
    size_t emu_bytes = 0;
    // emulate: the collectives of one group become ONE copy launch (as RCCL fuses a group into one kernel)
    struct EmuPiece { void* buf; size_t bytes; int back; };
    EmuPiece emu_q[8];
    int emu_n = 0, emu_grouped = 0;
    hipStream_t emu_stream = nullptr;
};

struct rtx_engine {
    rtx_cfg cfg;
    int NL = 0, I = 0, Z = 0, Ip = 0, Zp = 0;
    int Iin = 0;   // input columns = I + cfg.cond_dim
    int Bp_alloc = 0;
    bool bf16 = false, vae = false;
    size_t esz = 4;
    std::vector<Layer> L;
    std::vector<void*> allocs;
    float* Y = nullptr;
    float* Cacc = nullptr;
    size_t cacc_elems = 0;
    float *mu32 = nullptr, *lv32 = nullptr, *eps32 = nullptr;
    float2* lse_part = nullptr;
    int lse_strips = 0;
    float *tsum = nullptr, *lse = nullptr, *row_loss = nullptr, *sumsq = nullptr, *scratch_loss = nullptr;
    // bound tensors
    std::vector<float*> params, grads, m, v;
    std::vector<uint16_t*> grads16;   // optional bf16 gradient images (rtx_engine_bind_grads16; RTX_STEP_GRADS_BF16)
    bool bound = false, can_train = false, shadows_valid = false;
    TempCsr tmp_in, tmp_tg;
    // chunk stream of the batch's stored entries for the sparse first layer (spmm_in.hip)
    uint32_t* in_ent = nullptr;
    int32_t *in_desc = nullptr, *in_wsplit = nullptr;
    int64_t in_cap_chunks = 0;
    // batch image A[0] written by scatter (k_gather_scatter): per row slot, the columns the last launch wrote
    int32_t *img_written = nullptr, *img_nwritten = nullptr;
    int img_cap = 0;
    bool img_exact = false;           // A[0] is zero except the listed columns (any other writer of A[0] clears this flag)
    // The OTHER batch image (round 5): the NEXT step's gather (rtx_engine_set_next_batch) runs on the side stream, which idles
    // under this step's last weight-gradient + Adam launch, into a second image with lists and target sums of its own; the step
    // that then gets the announced batch swaps the two sets and starts with the first-layer product (the gather -- 7-9 us of pure
    // latency at the head of every step -- leaves the critical path).  swap_img_sets() exchanges these with L[0].A, tsum, img_*.
    void* A0_alt = nullptr;
    float* tsum_alt = nullptr;
    int32_t *img_written_alt = nullptr, *img_nwritten_alt = nullptr;
    int img_cap_alt = 0;
    bool img_exact_alt = false;
    struct { bool valid = false; rtx_batch b = {}; rtx_step s = {}; } next;        // announced for the step after the next call
    struct { bool valid = false; rtx_batch b = {}; uint64_t seed = 0, offset = 0; const uint8_t* mask = nullptr; } pre;   // gathered
    bool gather_done = false;         // run_forward: A[0] / tsum already hold this batch (a prefetch hit)
    int opt_prefetch = 1;             // 0: announced batches are ignored (A/B knob)
    int st_prefetch_hits = 0, st_prefetch_issued = 0;
    int st_join_folds = 0;             // deferred joins resolved INSIDE a first-layer product (get_option "join_folds")
    int opt_gather_scatter = 1;       // 0: k_gather rewrites the whole image every batch (rounds 1-3)
    // the weight-gradient + Adam kernels of the fused step run on a second stream beside the data-gradient chain
    hipStream_t side = nullptr;
    hipStream_t side_for = nullptr;   // the caller's stream the side stream was probed against (make_side_stream)
    std::map<hipStream_t, std::pair<hipStream_t, int>> side_cache;   // caller's stream -> (probed side stream, concurrent): a caller that
                                      //   alternates streams pays the ~1.5 ms probe once per stream, not on every change
    int side_concurrent = 0;          // 1: the probe saw the two streams run at the same time
    hipEvent_t ev_d[2 * RTX_MAX_LAYERS + 1] = {};   // ev_d[l]: D[l] is complete on the caller's stream
    hipEvent_t ev_done = nullptr;      // everything the step put on the side stream is complete
    // the two cross-stream dependencies of the fused step as stream memory operations (hipStreamWriteValue32 on the producing
    // stream, hipStreamWaitValue32 on the consuming one; option "hop_values", default on since round 4) instead of an event
    // record / wait: -3 us per step in three alternating pairs (283.7 / 280.1 / 276.0 -> 280.1 / 276.8 / 273.2,
    // profiles/r4_hop_values.txt); where the device cannot wait on a value the events remain
    int opt_hop_values = 1;
    int opt_f32_tail_split = 0;        // (measured: 963 vs 951 us/step, profiles/r5_fp32_tail_split.txt -- off) float32 parity mode: the last partial wave of a big weight-gradient product split over K (RtxGemm::tail_*)
    int n_cus = 256;                   // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    int opt_f32_adam_overlap = 0;      // (measured: 960.5 vs 960.7 us/step, no gain -- off; profiles/r5_fp32_tail_split.txt) float32 train step: the decoder matrix's Adam pass on the side stream under the remaining products
    int opt_f32_dw_split = 1;          // float32 parity mode: small weight-gradient products split over the batch (0: one workgroup per tile)
    int opt_splitk_bwd = 0;            // measurement: split factor of the K = n_items data-gradient product alone (0 = automatic)
    int opt_splitk_fwd = 0;            // measurement: split factor of the dense first-layer product alone (0 = automatic)
    uint32_t* hopk_mem = nullptr;      // the same two words in plain device memory, for the kernel form of the hop (k_hop_set / k_hop_wait)
    uint32_t hopk_seq = 0;
    // a step flagged RTX_STEP_DEFER_JOIN ends with a k_hop_set on the side stream instead of a wait on the caller's: the NEXT use of
    // the engine on a stream resolves it (resolve_join) -- folded into the first-layer product of the next training step when that
    // step starts from a prefetched batch image, as a one-wave k_hop_wait otherwise
    bool join_pending = false, join_fold = false;
    uint32_t join_seq = 0;
    int opt_timing_calibrate = 0;      // every timed bracket is followed by an empty one (site "<name>#empty"): what the events themselves cost
    // unused LDS of the decoder matrix's side-stream launch (RtxDw::lds_pad; knob "dw_side_pad").  12288 = one workgroup per CU beside the chain:
    // the chain's kernels then find registers at once (chain 95 -> 74 us on the timeline) and the step gains 2.3-4.3 us on fast and slow boxes
    // alike (profiles/r6_ab_small_waves_side_pad.txt) -- but the throttled launch itself stretches from 90 to 126 us and runs into the
    // encoder matrix's launch (90 -> 101 us): the step's dominant kernel would be REPORTED at 0.32 of the HBM roof instead of 0.40 for a 1 %
    // faster step.  Default off: the roofline of the dominant kernel is quoted for an unthrottled launch.
    int opt_dw_side_pad = 0;
    int opt_hop_fold = 1;              // the step's fork (caller's stream -> side stream) folded into the data-gradient product (loss_grads_impl)
    int opt_hop_kernels = 0;           // (measured: no gain, a one-wave kernel costs its stream 5-6 us like the packet it replaces) the two cross-stream dependencies of the step as one-wave kernels (stream_dependency)
    uint32_t* hop_mem = nullptr;       // [0]: caller's stream -> side stream, [1]: side stream -> caller's stream (signal memory)
    uint32_t hop_seq = 0;
    uint32_t hop_wrap = 0x7ffffff0u;   // the sequence restarts from zero here (option "hop_wrap": tests lower it)
    // measurement knobs (rtx_engine_set_option; defaults are the shipped configuration)
    int opt_fuse_adam = 1;      // bf16: Adam of every weight matrix inside its weight-gradient kernel (dw_adam.hip)
    int opt_dw_cfg = RTX_DW_64x128;
    int opt_dp_shard_min_elems = 1 << 20;   // sharded optimizer: weight matrices of at least this many elements are reduce-scattered /
                                //   updated by rows / all-gathered, smaller ones all-reduced and replicated (tests lower it so that
                                //   small golden networks exercise the sharded path with real data)
    uint32_t* loss_mailbox = nullptr;   // coherent host memory {loss bits, ticket, step}: rtx_engine_loss_mailbox / rtx_engine_wait_loss
    uint32_t loss_ticket = 0;           // ticket of the last loss reduction enqueued with the mailbox on (monotonic; never a step count)
    int opt_dp_one_comm = 0;    // 1: bucket A shares bucket B's communicator even when the plan brings a second one (ABI 5-6 schedule; A/B knob)
    int opt_dw_cfg_set = 0;     // 1: chosen through rtx_engine_set_option (the data-parallel step otherwise picks its own tile, see dw_cfg_of)
    int opt_lse_fuse = 1;       // log-sum-exp partials from the logits GEMM's epilogue (no separate pass over the logits)
    int opt_logits16 = 1;       // bf16 training step: the logits leave their product as IEEE half, written where d loss / d logits
                                //   (bf16, same size) goes, and the loss kernel turns them into it IN PLACE: 41 + 41 MB of float32
                                //   logits traffic per ml-20m step become 21 + 21 MB (the log-sum-exp still comes from the float32
                                //   accumulators; half keeps 11 significant bits -- the bf16 products' own error level)
    int opt_two_stream = 1;     // fused step: weight-gradient kernels on a side stream beside the data-gradient chain (-10 us)
    int opt_side_low_prio = 0;  // ... created with the lowest stream priority (1).  Round 3: OFF.  Neutral for the single-GPU step
                                //   (307.1 / 308.0 vs 308.2 / 308.5 us, A/B in one call), and with a live RCCL communicator in the process
                                //   -- any data-parallel job -- a lowest-priority queue beside RCCL's makes EVERY kernel of the step run 2-3x
                                //   slower (769 vs 343 us/step, profiles/r3_dp_priority_experiment.txt)
    int opt_in_on_main = 1;     // ... and the encoder matrix's kernel on the caller's stream behind the chain (see loss_grads_impl)
    int opt_sparse_in = 0;      // bf16, 1: the first encoder layer as a sparse VALU product over the stored entries (spmm_in.hip).
                                //   Default since round 4: the dense [batch, n_items] x [n_items, hidden] contraction on MFMA
                                //   (k_gather -> split-K rtx_gemm_nt -> k_post), the configuration BASELINE.json's north star names;
                                //   the sparse product is the measured alternative (2-6 us per ml-20m step faster at B = 500)
    int opt_small_fwd = 1;      // bf16: hidden layers / VAE head of the forward pass as one register-resident launch each (small_layers.hip)
    int opt_small_bwd = 1;      // ... and of the data-gradient chain (reads the transposed compute copies of the hidden layers)
    int opt_big_batch_tiles = 1;   // batches of >= 1024 rows: 512 x 128 data-gradient tiles (configs[3] on one GPU: 1486 -> 1343 us/step)
    int last_sparse_in = 0;     // what the last forward pass did with the first layer (rtx_engine_get_option "last_sparse_in")
    int opt_nt_regstage = 1;    // bf16: the K = n_items / N = n_items NT contractions on the register-staged kernel (gemm.hip):
                                //   33 + 30 us in the step against 41 + 40 us on the LDS-DMA kernel at B = 500 (1 workgroup / CU)
    // timing
    bool timing_all = false;
    std::map<std::string, int> timing_sites;   // site -> sampling period (every N-th launch of the site is bracketed by events)
    std::map<std::string, long> timing_seen;
    std::map<std::string, TimingSite> sites;
    std::vector<hipEvent_t> event_pool;
    DpState dp;
};

// ---- timing -------------------------------------------------------------------------------------
struct ScopedTimer {
    rtx_engine* e;
    hipStream_t s;
    TimingSite* site = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::string name;
    ScopedTimer(rtx_engine* eng, const char* nm, hipStream_t st) : e(eng), s(st), name(nm)
    {
        const char* name = nm;
        if (!e->timing_all && e->timing_sites.empty()) return;
        if (!e->timing_all) {
            auto it = e->timing_sites.find(name);
            if (it == e->timing_sites.end()) return;
            // an event record costs microseconds on the stream it is recorded on (two per timed launch): sample
            if (it->second > 1 && (e->timing_seen[name]++ % it->second) != 0) return;
        }
        site = &e->sites[name];
        auto get = [&]() {
            hipEvent_t ev;
            if (!e->event_pool.empty()) {
                ev = e->event_pool.back();
                e->event_pool.pop_back();
            } else if (hipEventCreate(&ev) != hipSuccess) {
                ev = nullptr;
            }
            return ev;
        };
        e0 = get();
        e1 = get();
        if (e0) (void)hipEventRecord(e0, s);
    }
    ~ScopedTimer()
    {
        if (!site) return;
        if (e1) (void)hipEventRecord(e1, s);
        if (e0 && e1) site->pending.push_back({e0, e1});
        if (e->opt_timing_calibrate && e1) {
            // an EMPTY bracket right behind the timed one, on the same stream at the same moment: the time between two event records
            // with nothing in between is what the bracket above contains besides its kernel (option "timing_calibrate";
            // reported as site "<name>#empty", the caller subtracts)
            hipEvent_t a = nullptr, b = nullptr;
            if (!e->event_pool.empty()) { a = e->event_pool.back(); e->event_pool.pop_back(); } else if (hipEventCreate(&a) != hipSuccess) a = nullptr;
            if (!e->event_pool.empty()) { b = e->event_pool.back(); e->event_pool.pop_back(); } else if (hipEventCreate(&b) != hipSuccess) b = nullptr;
            if (a && b) {
                (void)hipEventRecord(a, s);
                (void)hipEventRecord(b, s);
                e->sites[name + "#empty"].pending.push_back({a, b});
            }
        }
    }
};
#define RTX_CAT2(a, b) a##b
#define RTX_CAT(a, b) RTX_CAT2(a, b)
#define TIMED(name) ScopedTimer RTX_CAT(_timer_, __LINE__)(e, name, st)

// ---- helpers defined in engine.hip and used by the other translation units (hidden: not exported from librectorch_hip.so)
#define RTX_INTERNAL __attribute__((visibility("hidden")))
RTX_INTERNAL int dev_alloc(rtx_engine* e, void** p, size_t bytes, bool zero = true);
extern "C" RTX_INTERNAL int dp_layout_order(const rtx_engine* e, int* order);          // (defined inside engine.hip's extern "C" block)
extern "C" RTX_INTERNAL size_t dp_region_elems(const rtx_engine* e, const DpState& d, int t);
RTX_INTERNAL void dp_release(rtx_engine* e);          // engine_dp.hip
RTX_INTERNAL int ensure_shadows(rtx_engine* e, hipStream_t st);
RTX_INTERNAL int check_ready(rtx_engine* e, bool train);
RTX_INTERNAL size_t plan_cacc_elems(rtx_engine* e, int Np, int Kp);      // split-K scratch a product of this shape needs (set_option "splitk" re-sizes it)

