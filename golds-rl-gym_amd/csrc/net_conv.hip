// ConvSingleAgentPolicyNetwork on gfx950 (reference fed_gym/agents/paac/policy_v_network.py:5-80):
// forward, Gaussian action sampling (paac/paac.py:414-419), loss + backward, clip-by-global-norm and
// Adam (paac/actor_learner.py:31-68), and the device-resident PAAC rollout (paac/paac.py:302-372).
//
// The 84x84x3 input image is never materialised: conv1 (8x8/4) is evaluated SPARSELY from the
// compact observation -- an image has at most 80+10+1 non-zero pixels, so the exact same sum has
// ~12k instead of 2.46M multiply-adds -- by one workgroup per env that stages the 84x84 count
// grids in LDS, builds the pre-activation shared by the env's 10 agents once, and adds each
// agent's one-hot tap.  conv2/conv3/dense layers are fp32-MFMA implicit GEMMs (net_gemm.h).
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/goldsrl_net.h"
#include "common.h"
#include "net_gemm.h"
#include "net_reduce.inc"
#include "rng.h"

namespace grl {

enum : uint32_t { RS_ACTION = 16 };

// parameter offsets in the flat vector (tf.trainable_variables() order)
struct ConvOffsets {
    static constexpr long c1w = 0, c1b = c1w + 8 * 8 * 3 * 32, c2w = c1b + 32, c2b = c2w + 4 * 4 * 32 * 64, c3w = c2b + 64,
                          c3b = c3w + 3 * 3 * 64 * 64, d1w = c3b + 64, d1b = d1w + 3136 * 512, d2w = d1b + 512,
                          d2b = d2w + 512 * 256, p1w = d2b + 256, p1b = p1w + 256 * 512, heads = p1b + 512;
};
// everything behind pol1 depends on num_actions A (policy_v_network.py:40-43: mu / sigma are Dense(A) heads):
//   mu_w[512,A] mu_b[A] sigma_w[512,A] sigma_b[A] v1_w v1_b v2_w v2_b v3_w[256,1] v3_b
constexpr int GRL_MAX_ACTIONS = 4;
struct HeadOff {
    long muw, mub, sgw, sgb, v1w, v1b, v2w, v2b, v3w, v3b, total;
    int A, dz;      // dz: floats per row of the heads' dZ scratch = [dz_mu(A) | dz_sigma(A) | dz_v | policy-loss term | critic-loss term], padded
};
static HeadOff head_offsets(int A) {
    HeadOff o;
    o.A = A; o.dz = 2 * A + 4;
    o.muw = ConvOffsets::heads; o.mub = o.muw + 512L * A; o.sgw = o.mub + A; o.sgb = o.sgw + 512L * A; o.v1w = o.sgb + A;
    o.v1b = o.v1w + 256 * 512; o.v2w = o.v1b + 512; o.v2b = o.v2w + 512 * 256; o.v3w = o.v2b + 256; o.v3b = o.v3w + 256; o.total = o.v3b + 1;
    return o;
}

using GatherConv2 = ConvGather<9, 9, 2, 2, 0, 0, 4, 4, 32, 20, 20, false>;    // a1[n][20][20][32] -> (n*81, 512)
using GatherConv2Relu = ConvGather<9, 9, 2, 2, 0, 0, 4, 4, 32, 20, 20, false, true>;   // relu(sraw) -> the same operand
using GatherConv3 = ConvGather<7, 7, 1, 1, 0, 0, 3, 3, 64, 9, 9, false>;      // a2[n][9][9][64]  -> (n*49, 576)
// data gradients (transposed convolutions, zero-filled borders):
using GatherT3 = ConvGather<9, 9, 1, 1, -2, -2, 3, 3, 64, 7, 7, true>;        // dz3[n][7][7][64] -> rows (n,y,x) of a2
using GatherT3PM = ConvGatherPM<9, 9, -2, -2, 3, 3, 64, 7, 7>;               // same, pixel-major rows: border taps skipped
using GatherT2 = ConvGather<10, 10, 1, 1, -1, -1, 2, 2, 64, 9, 9, true>;      // dz2[n][9][9][64] -> rows (n,yh,xh) per parity

}  // namespace grl

// Per-stream workspace: every buffer one chunk's forward / backward pass writes.  Independent chunks are enqueued
// round-robin on a few "lanes" (HIP streams, each with its own copy of this struct) so that the memory-bound helper kernels
// of one chunk overlap the MFMA GEMMs of the others; grl_net derives from it, so kernels keep using net->d1 etc. and
// use_lane() swaps the whole pointer set between chunk enqueues (host-side, sequential).
constexpr int GRL_MAX_LANES = 8;
constexpr int kLossScaleAt = 8, kLossBoundAt = 10;      // slots of grl_net::stats (net_train.inc: loss scale)
constexpr int kRangeLatchAt = 12;                        // int: the range flag of a gradient pass, latched before the post-update background pass (train_apply)
struct NetLane {
    float *grads;              // this lane's gradient accumulator (lane 0's is THE gradient; lane 1's is added before the all-reduce)
    // forward activations (chunk)
    float *a1, *a2, *a3, *d1, *d2, *p1, *v1, *v2;
    // gradients of activations (chunk)
    float *ga1, *ga2, *ga3, *gd1, *gd2, *gp1, *gv1, *gv2;
    // shared-trunk evaluation of conv1/conv2 (net_shared.inc): per-ENV tensors
    float *sraw, *z2sh, *dz2, *gt;
    // shared conv3 gradients: per-env a2sh = relu(z2sh), DZ3; per (agent, slot <= 9 touched conv2 pixels): pixel id,
    // a2_a[u], a2_a[u] - a2sh[u], T3(dz3_a)[u], masked dz2_a[u]; tmpw3: correction GEMM output before the tap flip
    float *a2sh, *d2s, *gsl, *dza, *dz3sh, *tmpw3;
    unsigned long long *m2s;   // ReLU mask of the touched conv2 pixels: one word of 64 channel bits per compact slot row
    float *dl2, *tt2, *tmpw2;   // conv2-level corrections as class-major GEMM operands (net_shared.inc)
    int2 *t2desc;               // ... and the gather descriptors of their A rows (T2SlotGather)
    // shared conv3 forward: z3sh = conv3(a2sh) + b3 per env; the per-slot products (a2_a - a2sh)[u] . W3[tap]
    // (n*9 x 576) live in the a2 buffer, which has no other use in shared-trunk mode
    float *z3sh;
    // dense1 on the shared a3 (net_patch.inc): per-env a3sh and ysh = W^T a3sh + b; per agent the 5x5 patch values v3 and
    // differences d3 (1600 floats); group-sorted sample order for the patch GEMMs
    float *a3sh, *d3, *ysh;
    unsigned long long *m3;    // ReLU mask of the agent's 5x5 patch of a3: 25 words of 64 channel bits per sample
    float *gd1sh, *gsh3, *g3p, *dz3p;     // gradient side: per-env sum of gd1, its dense1 data gradient, per-agent patch gradients
    int *perm, *goffp, *blkcnt, *blkoff, *sbeg, *send, *sgrp, *binbase;
    unsigned short *pkey, *prank;      // sort key of a sample (group, column span, row span) and its rank inside its workgroup (net_patch.inc)
    unsigned *smask, *tmask, *zmask, *wmask;   // 25-bit patch support per sample / union per 128 sorted rows / per weight-gradient slice / slice union per sample
    int *slot_of;                      // sorted slot of a sample (inverse of perm)
    int2 *rowdesc;                     // gather descriptors of the compact slot rows (slot_rowdesc_kernel)
    // tap-class order of the slot rows (slot_sort, net_shared.inc): key / rank / live-tap mask per row, sort scratch, sorted -> compact
    // row, live-tap union per 256 sorted rows
    unsigned char *skey;
    unsigned short *srank, *stap;
    int *sblkcnt, *sblkoff, *sbinbase, *sperm;
    // the shared trunk's affected conv2 rows (trunk_index, net_shared.inc): 81-bit mask per env, row list + live count
    unsigned *tamask, *tbmask, *tcmask, *tnmask;      // (tnmask: blocks of sraw anybody reads) + 100-bit mask of touched 2x2 conv1 pixel blocks, 49-bit mask of affected conv3 outputs
    int *trowlist, *tblklist, *tc3list, *trows_n, *twgcnt, *twgoff;      // their lists, live counts [conv2 rows, blocks, conv3 rows], scan scratch
    unsigned *tumask;          // union of the chunk's conv3 masks (2 words)
    unsigned *tneed2;          // conv2 pixels inside the 3 x 3 windows of that union (3 words): what conv3 reads of a2sh
    float *tubias;             // dense1's per-env bias under that union (trunk_ubias_kernel)
    float *tug, *tuspix;       // gradient side: G = column sums of gd1sh (512), per-pixel closed-form sums (49 x 64)
    float *tslab, *tsums;      // tslab: per-workgroup sums of dz2 over unaffected rows; tsums: [dz2 total 64 | unaffected 64]
    unsigned *stmask, *szmask;      // ... and per row range of the slot weight gradient (slot_wgrad_launch)
    int *sbase, *rowagent, *sblk;      // compact slot rows (net_shared.inc): prefix of per-sample slot counts, row -> sample, scan scratch
    signed char *tilegroup, *org;
    signed char *ulist;
    float *slab;               // split-M partial sums
    float *slab1h;             // one-hot conv1 tap partials of agent_ds_kernel
    float *slabb;              // per-wave-tile column sums written by the EpiGradSum / EpiGradStride2 data-gradient GEMMs (bias gradients)
    float *slab1b;             // per-workgroup sums of agent_ds_kernel's dS corrections (conv1's bias gradient)
    float *cfold;              // kFoldParts x 32 partial sums of fold_class_sums_kernel
    // conv2 corrections as a GEMM (net_shared.inc): A rows, class sort, canonical-slot maps
    float *carow;
    int *cperm, *cblkcnt, *cblkoff, *cgoff, *cslot, *cinv;
    signed char *ctilegroup;
    double *slab64;
    float *ro_mu;              // chunk-sized scratch of the gradient step (cmu csigma cvs dzh cact cadv cy)
    float *ws_a3, *ws_d1, *ws_d2, *ws_p1, *ws_v1, *ws_v2, *ws_a3sh, *ws_d3;   // chunk workspace (the default binding)
    unsigned long long *ws_m3;
    // sign bits of d2 / v1 (4 + 8 words of 64 column bits per sample: EpiBiasActBits, net_gemm.h), bound like the activations
    unsigned long long *mb_d2, *mb_v1, *ws_mb;
    float *ws_sraw, *ws_a2sh, *ws_d2s;
    unsigned long long *ws_m2s;
    signed char *ws_ulist;
    void *ws_idx[24];          // the chunk workspace of the index lists GRL_IDX_LIST names (bound like the activations at keep level 3)
    bool train_ready;
};

struct grl_net : NetLane {
    grl_handle *h;
    grl_net_config cfg;
    std::string err;
    int chunk;                 // samples per pass
    float *params, *adam_m, *adam_v;
    float *paramsT;            // W^T of every GEMM layer at the same flat offsets ([N][K]: the forward's Bt operand)
    long adam_t;
    float *w3t, *w2t;          // rearranged conv weights for the data gradients
    float *w3f;                // w3f[(tap,co)][ci] = W3[tap][ci][co] (slot product GEMM of conv3's forward)
    float *wpvT;               // [pol1_w | v1_w] transposed side by side: (1024, 256), the B operand of the one GEMM that computes both from d2
    int shared_trunk;
    // GEMM arithmetic: 0 = three fp16 products (operands must stay inside the fp16 range), 1 = v_mfma_f32_16x16x4_f32 (no range
    // limit, 103 instead of 200 TFLOP/s).  A pass that raised the range flag switches the net to 1 (range_fallback below).
    double sfrac;              // likewise for the conv3 slot GEMMs: executed share of the 9 taps (per 256-row tile)
    double pfrac[3];           // executed share of the dense1 patch GEMMs' FLOPs in the last sorted chunk (profiling pass only; else 1)
    // the background's way through the trunk (trunk_background, net_shared.inc; per parameter version): the all-b1 image, [z2 | a2] of
    // its conv2 row, a2 at all 81 pixels, [z3 | a3] of its conv3 row, the one-row list of those passes
    float *tbgimg, *tbgz, *tbgimg3, *tbgz3, *tybg;      // tybg[49][512]: a background pixel's contribution to dense1 (trunk_ybg_kernel)
    int *tbglist;
    int trunk_skip;            // 1: conv2's forward / weight gradient / transposed convolution run over the rows the env's bins reach (GRL_TRUNK_SKIP=off: all rows)
    int patch_skip;            // 1: the dense1 patch GEMMs skip what the support masks say is zero (GRL_PATCH_SKIP=off: the plain 5x5 patch)
    int gemm_f32, range_fallback_on, range_fallbacks, range_bits_last, update_skipped_last;
    // the way back: gemm_f32 was set BY a range violation (not by GRL_NET_GEMM=f32 / grl_net_set_gemm_f32), updates in a row whose
    // largest |GEMM output| stayed below 65 504 / 4, how many of them take the net back (0: never), how often that happened
    int f32_by_fallback, f32_clean_passes, range_return_k, range_returns;
    float absmax_last;
    int loss_scale_on;         // per-pass power-of-two scale of the head gradients (net_train.inc); GRL_NET_LOSS_SCALE=off disables it
    int acc1;                  // gemm_rowk instances built with ACC1 use it (GRL_NET_ACC1=off: the two-accumulator form everywhere)
    int expand3_gather;        // conv3's per-agent corrections as a gather GEMM at the patch pixels (default; GRL_NET_EXPAND3=prod: slot products + expansion kernel)
    int expand2_gemm, ctiles;  // conv2's per-agent corrections as a class-sorted GEMM (default) or the LDS-resident kernel (GRL_NET_EXPAND2=lds)
    float *w2corr;             // [4 classes][576][128] kernel slices of that GEMM, rebuilt with the transposes
    grl::HeadOff ho;           // offsets of the head / value parameters for this net's num_actions
    int npad, ptiles, pslices, pslice_rows, pwgrad_xcd, pdgrad_xcd, pwgrad_pair, pdgrad_pair, swgrad_pair, swgrad_chunks, slots_xcd;
    int tn_wgs, tn_wgs_dense;  // workgroups a split-M weight-gradient launch aims at (slab count = tn_wgs / tiles)      // dense1 patch weight gradient: slice length and tile order (tuning knobs)
    size_t slab_floats, slab_used, slabb_floats, slabb_used;      // bump allocation of a chunk's partial-sum regions (net_train.inc)
    std::vector<grl::RJob> rq;      // the chunk's queued reductions (net_reduce.inc)
    std::vector<int> rq_dst;
    float *stats;              // device: [0..4] loss parts / norm / clip factor, [8..9] loss scale S and 1/S, [10] bits of the head-gradient bound
    // rollout storage (allocated by grl_net_rollout)
    int T, B;
    uint8_t *ro_lb, *ro_ab, *ro_pos, *ro_done;      // ro_done (T,E): episode_over after each step (R6 / tests; the grid return ignores it, Q5)
    float *ro_act, *ro_envact, *ro_val, *ro_rew, *ro_y, *ro_adv, *ro_boot, *ro_pmu, *ro_psg;
    float *mu, *sigma, *vs;    // (B,2) (B,2) (B) of the last predict
    uint8_t *tmp_lb, *tmp_ab, *tmp_pos;
    int tmp_envs;
    unsigned long act_counter;
    std::vector<void *> allocs;
    bool prof_on;
    std::vector<hipEvent_t> prof_ev;
    std::vector<unsigned char> prof_tag;     // GRL_PROF_TAG_* of every bracketed launch
    std::vector<double> prof_launch_flops;
    int prof_tag_cur;
    size_t prof_used;
    double prof_flops;
    int last_n;                // samples in the last chunk (for read_activation)
    void *comm;                // ncclComm_t (RCCL) for the per-rollout gradient all-reduce, or nullptr
    int comm_world, comm_rank;
    hipEvent_t ar_ev0, ar_ev1; // bracket the all-reduce on the handle's stream (grl_net_comm_info)
    int ar_pending;
    long ar_calls;
    // host clocks of the actor loop (grl_net_host_times)
    long ht_rollouts, ht_updates;
    double ht_rollout_ms, ht_train_enq_ms, ht_train_wait_ms;
    double ht_train_t0;      // steady-clock ms at the entry of the gradient step in flight (0: none)
    double ar_ms_total;
    float ar_ms_last;
    // Rollout-resident activations: the rollout's forward pass writes its activations of every (step, chunk) into one
    // T*B-sample buffer (what 288 GB of HBM is for) and the gradient step reads them back instead of recomputing the
    // forward pass.  Exact: parameters do not change between the two (paac.py:302-387).  keep_version tracks that;
    // GRL_NET_F_RECOMPUTE_FORWARD or too little free memory selects recomputation.
    int keep_level;            // 0: nothing resident, 1: conv3/dense activations, 2: + the per-env trunk tensors the gradient step reads
    float *keep;
    size_t keep_slots;
    long param_version, keep_version;
    float last_inv_total;      // 1 / samples of the last gradient pass (grl_net_apply_grads)
    // lanes
    NetLane lanes[GRL_MAX_LANES];
    hipStream_t lane_stream[GRL_MAX_LANES];      // [0] = the handle's stream
    hipEvent_t ev_fork, ev_join[GRL_MAX_LANES];
    // The index kernels of a forward pass that depend on the agents' positions alone (slot lists, class / slot / patch sorts: ~17 small
    // launches, each a dependent step of a few us) run on a side stream per lane beside the env-level trunk (net_shared.inc,
    // forward_conv12_shared): with one chunk per step -- 4 096 or 8 192 envs per GPU -- nothing else hides them (round 5).
    hipStream_t side_stream[GRL_MAX_LANES];
    hipEvent_t ev_side0[GRL_MAX_LANES], ev_side1[GRL_MAX_LANES], ev_side2[GRL_MAX_LANES];
    int idx_side, side_now, pitem_on_side;      // side_now: set by grl_net_rollout while it enqueues a one-chunk-per-step rollout
    int nlanes, cur_lane, last_lane;
};

namespace grl {

static int nfail(grl_net *n, int code, const std::string &msg) {
    if (n) n->err = msg;
    return code;
}
#define NET_HIP(n, call)                                                                                   \
    do {                                                                                                   \
        hipError_t _e = (call);                                                                            \
        if (_e != hipSuccess) return nfail(n, GRL_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T>
static int nalloc(grl_net *n, T **p, size_t count) {
    NET_HIP(n, hipMalloc((void **)p, count * sizeof(T)));
    n->allocs.push_back(*p);
    NET_HIP(n, hipMemsetAsync(*p, 0, count * sizeof(T), n->h->stream));
    // allocations happen on whichever lane is current (a non-blocking stream) while other streams may be the first to touch the
    // buffer (w2t / w3t are written on the main stream): finish the fill before anyone can see the pointer
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    return GRL_OK;
}

// ---- lanes: swap the per-stream workspace (and the stream kernels are enqueued on) between chunk enqueues
static void use_lane(grl_net *net, int k) {
    if (k == net->cur_lane) return;
    net->lanes[net->cur_lane] = static_cast<NetLane &>(*net);
    static_cast<NetLane &>(*net) = net->lanes[k];
    net->h->stream = net->lane_stream[k];
    net->cur_lane = k;
}
static int lanes_active(const grl_net *net) { return net->prof_on ? 1 : net->nlanes; }
// lane 1 sees everything enqueued on the main stream so far
static int lanes_fork(grl_net *net) {
    if (lanes_active(net) < 2) return GRL_OK;
    use_lane(net, 0);
    NET_HIP(net, hipEventRecord(net->ev_fork, net->lane_stream[0]));
    for (int k = 1; k < net->nlanes; ++k) NET_HIP(net, hipStreamWaitEvent(net->lane_stream[k], net->ev_fork, 0));
    return GRL_OK;
}
// the main stream sees everything enqueued on lane 1; lane 0 becomes current again
static int lanes_join(grl_net *net) {
    use_lane(net, 0);
    if (lanes_active(net) < 2) return GRL_OK;
    for (int k = 1; k < net->nlanes; ++k) {
        NET_HIP(net, hipEventRecord(net->ev_join[k], net->lane_stream[k]));
        NET_HIP(net, hipStreamWaitEvent(net->lane_stream[0], net->ev_join[k], 0));
    }
    return GRL_OK;
}

// live compact slot rows of the chunk at hand -- read back only while profiling (the profiled pass is serialised anyway), so that
// the slot GEMMs' FLOPs are counted on the rows they really process; otherwise the upper bound
static double slot_rows_for_flops(grl_net *net, int n) {
    if (!net->prof_on) return 9.0 * n;
    int live = 9 * n;
    if (hipStreamSynchronize(net->h->stream) == hipSuccess) (void)hipMemcpy(&live, net->sbase + n, sizeof(int), hipMemcpyDeviceToHost);
    return (double)live;
}

// which family a bracketed GEMM launch belongs to (grl_net_profile_read_tags); set with net->prof_tag_cur before a group of launches
enum {
    PT_OTHER = 0, PT_DENSE_FWD, PT_DENSE_DGRAD, PT_DENSE_WGRAD, PT_PATCH_FWD, PT_PATCH_DGRAD, PT_PATCH_WGRAD, PT_ENV_FWD, PT_ENV_DGRAD,
    PT_ENV_WGRAD, PT_SLOT_FWD, PT_SLOT_DGRAD, PT_SLOT_WGRAD, PT_CLASS_CORR, PT_PER_AGENT, PT_COUNT
};

// every GEMM launch goes through these two: the net's arithmetic form picks the instantiation
template <int BM, int BN, int WGM, int WGN, class AG, class Epi, int XCD = 1, bool FENCE = true, int NBUF = 1, bool ACC1 = false>
static void launch_rowk(grl_net *net, dim3 grid, hipStream_t st, AG ag, const float *Bt, int ldb, int N, Epi epi) {
    // ACC1, the one-accumulator form (net_gemm.h), buys a workgroup per CU without spilling on the eight-wave 128 x 128 tiles (92-98 -> 78
    // registers, six waves per SIMD) and the four-wave tiles of <= 128 x 64 (134-150 -> 94-116, four or five instead of three); the
    // eight-wave 256 x 64 instances spill 16-96 bytes at the 80 registers a third workgroup needs, the four-wave 64 x 64-per-wave tiles
    // 320.  Measured on every instance it fits (round 4, per 81 920-sample chunk): pol1 + v1 190 -> 173 us, conv2's class corrections
    // 113 -> 102, the conv3 patch gather 124 -> 120 (interior 390 -> 379); the other dense forwards / data gradients -2 ... -3 %; the
    // NBUF = 2 instances +3 ... +7 % (80 KB of LDS), EpiGradSum +29 % (44 bytes of scratch) -- so the ones that gain use it.  (It also
    // changes the rounding of a sum, so the two forms of a layer that a test compares bit for bit must agree on it.)
    constexpr bool kAcc1 = ACC1 && ((WGM * WGN == 8 && BM == 128 && BN == 128) || (WGM * WGN == 4 && BM * BN <= 128 * 64));
    if (net->gemm_f32)
        hipLaunchKernelGGL((gemm_rowk<BM, BN, WGM, WGN, AG, Epi, XCD, FENCE, true, NBUF, false>), grid, dim3(64 * WGM * WGN), 0, st, ag, Bt, ldb, N, epi);
    else if (kAcc1 && net->acc1)
        hipLaunchKernelGGL((gemm_rowk<BM, BN, WGM, WGN, AG, Epi, XCD, FENCE, false, NBUF, kAcc1>), grid, dim3(64 * WGM * WGN), 0, st, ag, Bt, ldb, N, epi);
    else
        hipLaunchKernelGGL((gemm_rowk<BM, BN, WGM, WGN, AG, Epi, XCD, FENCE, false, NBUF, false>), grid, dim3(64 * WGM * WGN), 0, st, ag, Bt, ldb, N, epi);
}
template <int BM, int BN, int WGM, int WGN, class AG, int XCD = 1>
static void launch_tn(grl_net *net, dim3 grid, hipStream_t st, AG ag, const float *dY, int J, int mc, float *slab) {
    if (net->gemm_f32) hipLaunchKernelGGL((gemm_tn<BM, BN, WGM, WGN, AG, XCD, true>), grid, dim3(64 * WGM * WGN), 0, st, ag, dY, J, mc, slab);
    else hipLaunchKernelGGL((gemm_tn<BM, BN, WGM, WGN, AG, XCD, false>), grid, dim3(64 * WGM * WGN), 0, st, ag, dY, J, mc, slab);
}

// Measurement aid (GRL_NET_EVICT=1, tools/mall_probe.sh): a 512 MiB write in front of a kernel, so that the kernel finds neither
// its producer's output nor its weights in L2 / the 256 MB Infinity Cache.  Comparing a kernel's duration with and without it says
// how much of its FETCH_SIZE the cache hierarchy behind L2 was serving.
__global__ void evict_kernel(float4 *__restrict__ buf, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static void maybe_evict(grl_net *n) {
    static const bool on = getenv("GRL_NET_EVICT") && atoi(getenv("GRL_NET_EVICT")) != 0;
    if (!on) return;
    static float4 *buf = nullptr;      // one per process: only ever written
    constexpr size_t bytes = (size_t)512 << 20;
    if (!buf && hipMalloc((void **)&buf, bytes) != hipSuccess) { buf = nullptr; return; }
    hipLaunchKernelGGL(evict_kernel, dim3(2048), dim3(256), 0, n->h->stream, buf, bytes / 16);
}

struct GemmTimer {
    grl_net *n;
    GemmTimer(grl_net *net, double flops) : n(net) {
        maybe_evict(n);
        if (n->prof_on && n->prof_used + 2 <= n->prof_ev.size()) {
            (void)hipEventRecord(n->prof_ev[n->prof_used], n->h->stream);
            n->prof_flops += flops;
            n->prof_tag[n->prof_used / 2] = (unsigned char)n->prof_tag_cur;
            n->prof_launch_flops[n->prof_used / 2] = flops;
        }
    }
    ~GemmTimer() {
        if (n->prof_on && n->prof_used + 2 <= n->prof_ev.size()) {
            (void)hipEventRecord(n->prof_ev[n->prof_used + 1], n->h->stream);
            n->prof_used += 2;
        }
    }
};

// dst[n][k] = src[k][n]
__global__ void transpose_kernel(const float *__restrict__ src, int K, int N, float *__restrict__ dst) {
    __shared__ float tile[32][33];
    int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y)
        if (k0 + r < K && n0 + (int)threadIdx.x < N) tile[r][threadIdx.x] = src[(long)(k0 + r) * N + n0 + threadIdx.x];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y)
        if (n0 + r < N && k0 + (int)threadIdx.x < K) dst[(long)(n0 + r) * K + k0 + threadIdx.x] = tile[threadIdx.x][r];
}

__global__ void conv2_corr_weights_kernel(const float *__restrict__ w2, float *__restrict__ bt);      // net_shared.inc
// keep paramsT in step with params (after set_params, Adam, broadcast)
static int trunk_background(grl_net *net);      // net_shared.inc
static void refresh_transposes(grl_net *net) {
    const struct { long off; int K, N; } L[] = {
        {ConvOffsets::c2w, 512, 64}, {ConvOffsets::c3w, 576, 64}, {ConvOffsets::d1w, 3136, 512}, {ConvOffsets::d2w, 512, 256},
        {ConvOffsets::p1w, 256, 512}, {net->ho.v1w, 256, 512}, {net->ho.v2w, 512, 256}};
    for (const auto &l : L) {
        // pol1 and v1 run as ONE N = 1024 GEMM on d2 (forward_chunk): their transposed kernels sit side by side in wpvT
        float *dst = l.off == ConvOffsets::p1w ? net->wpvT : l.off == net->ho.v1w ? net->wpvT + (size_t)512 * 256 : net->paramsT + l.off;
        hipLaunchKernelGGL(transpose_kernel, dim3((l.N + 31) / 32, (l.K + 31) / 32), dim3(32, 8), 0, net->h->stream, net->params + l.off,
                           l.K, l.N, dst);
    }
    hipLaunchKernelGGL(conv2_corr_weights_kernel, dim3((4 * 576 * 128 + 255) / 256), dim3(256), 0, net->h->stream, net->params + ConvOffsets::c2w,
                       net->w2corr);
    for (int tap = 0; tap < 9; ++tap)     // w3f[(tap, co)][ci]: each tap's 64x64 block transposed
        hipLaunchKernelGGL(transpose_kernel, dim3(2, 2), dim3(32, 8), 0, net->h->stream, net->params + ConvOffsets::c3w + tap * 4096, 64, 64,
                           net->w3f + tap * 4096);
    (void)trunk_background(net);
}

// ------------------------------------------------------------------------------------------ conv1 (sparse)
// One workgroup per env.  Axis convention of the reference image: states[n][h][w][c] with h = x-bin,
// w = y-bin (np.histogram2d's first output axis is x; state_processors.py:31-33).
__global__ __launch_bounds__(256) void conv1_sparse_kernel(const uint8_t *__restrict__ lbins, const uint8_t *__restrict__ abins,
                                                           const uint8_t *__restrict__ pos, const float *__restrict__ w1,
                                                           const float *__restrict__ b1, float *__restrict__ a1, int G,
                                                           float *__restrict__ sraw) {
    __shared__ unsigned int cnt[2][7056 / 4 + 4];   // 84x84 byte counters per channel, packed 4 per word
    __shared__ float S[400 * 32];                    // pre-activation shared by the env's 10 agents
    const int env = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 2 * (7056 / 4 + 4); i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    if (tid < 80) {
        int bx = lbins[((size_t)env * 80 + tid) * 2], by = lbins[((size_t)env * 80 + tid) * 2 + 1];
        if (bx != 255) { int idx = bx * G + by; atomicAdd(&cnt[0][idx >> 2], 1u << (8 * (idx & 3))); }
    } else if (tid < 90) {
        int a = tid - 80;
        int bx = abins[((size_t)env * 10 + a) * 2], by = abins[((size_t)env * 10 + a) * 2 + 1];
        if (bx != 255) { int idx = bx * G + by; atomicAdd(&cnt[1][idx >> 2], 1u << (8 * (idx & 3))); }
    }
    __syncthreads();
    for (int pix = tid; pix < 400; pix += 256) {
        const int oy = pix / 20, ox = pix - oy * 20;
        float acc[32];
#pragma unroll
        for (int co = 0; co < 32; ++co) acc[co] = b1[co];
        for (int ky = 0; ky < 8; ++ky) {
            const int rowbase = (4 * oy + ky) * G + 4 * ox;   // multiple of 4: G = 84
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                unsigned int w0 = cnt[c][rowbase >> 2], w1_ = cnt[c][(rowbase >> 2) + 1];
                if ((w0 | w1_) == 0) continue;
#pragma unroll
                for (int kx = 0; kx < 8; ++kx) {
                    unsigned int k = ((kx < 4 ? w0 : w1_) >> (8 * (kx & 3))) & 255u;
                    if (k == 0) continue;
                    // x_grid / len(state[0]) in float64, fed to the float32 placeholder (state_processors.py:33)
                    float val = (float)((double)k / (c == 0 ? 80.0 : 10.0));
                    const float *wp = w1 + ((ky * 8 + kx) * 3 + c) * 32;
#pragma unroll
                    for (int co = 0; co < 32; ++co) acc[co] += val * wp[co];
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 32; ++co) S[pix * 32 + co] = acc[co];
    }
    __syncthreads();
    if (sraw) {   // shared-trunk mode (net_shared.inc): one pre-activation image per ENV, no per-agent copies
        // only the pre-activation is stored: conv2's gathers apply the ReLU while loading (GatherConv2Relu)
        float4 *o0 = reinterpret_cast<float4 *>(sraw + (size_t)env * 12800);
        for (int i = tid; i < 3200; i += 256) o0[i] = reinterpret_cast<const float4 *>(S)[i];
        return;
    }
    for (int a = 0; a < 10; ++a) {
        const int ph = pos[((size_t)env * 10 + a) * 2], pw = pos[((size_t)env * 10 + a) * 2 + 1];
        float4 *out = reinterpret_cast<float4 *>(a1 + ((size_t)env * 10 + a) * 12800);
        for (int i = tid; i < 3200; i += 256) {
            const int pix = i >> 3, co = (i & 7) * 4;
            const int oy = pix / 20, ox = pix - oy * 20;
            float4 v = *reinterpret_cast<const float4 *>(&S[pix * 32 + co]);
            const int ky = ph - 4 * oy, kx = pw - 4 * ox;
            if ((unsigned)ky < 8u && (unsigned)kx < 8u) {   // the agent's one-hot pixel lies in this window
                const float *wp = w1 + ((ky * 8 + kx) * 3 + 2) * 32 + co;
                v.x += wp[0]; v.y += wp[1]; v.z += wp[2]; v.w += wp[3];
            }
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            out[i] = v;
        }
    }
}

// Shared-trunk form of the same layer (net_shared.inc): ONE pre-activation image per env, written straight from registers.
// Only the pixels whose 8x8 window holds a bin ("touched": ~40 of 400) do arithmetic, the others are b1.  The touched pixels are
// listed in LDS (400-bit mask from the bins' four covers, compacted in pixel order) and a pixel gets HALF A WAVE, one output
// channel per lane: a step of the window walk -- one non-zero (row, channel, column) of the count grids -- then costs a dozen
// instructions for 2 pixels x 32 channels.  (Until round 3 a lane owned 8 channels of one of 16 pixels of its wave: the wave walked
// the union of its pixels' non-zero taps with 4 of 64 lanes busy per step, ~110 of the kernel's 152 us per 8 192-env chunk; and
// the count / 80 division sat on that path too -- now tabulated.)  Same sums in the same order as conv1_sparse_kernel
// (bit-identical output).  nmask (list form of the trunk): the 2 x 2 pixel blocks anybody reads; the rest is not written.
// Round 4: one resident wave of workgroups strides over the envs: the 16 KB of weights, the two value tables and the zeroed count
// grids are set up once per workgroup, an env's <= 90 increments are taken back after its walk (instead of zeroing 14 KB per env),
// and the next env's bins are fetched while the current one is walked.
__global__ __launch_bounds__(256) void conv1_sparse_shared_kernel(const uint8_t *__restrict__ lbins, const uint8_t *__restrict__ abins,
                                                                  const float *__restrict__ w1, const float *__restrict__ b1, int G,
                                                                  float *__restrict__ sraw, const unsigned *__restrict__ nmask, int nenv) {
    __shared__ unsigned int cnt[2][7056 / 4 + 4];   // 84x84 byte counters per channel, packed 4 per word
    __shared__ unsigned need[4], pm[13];            // blocks anybody reads (nmask == nullptr: all); touched pixels
    __shared__ float vtab[2][96];                   // count / 80, count / 10 as the reference forms them (float64 division rounded to float32, state_processors.py:33)
    __shared__ unsigned short plist[400];
    __shared__ unsigned short evl[8][128];          // per half wave: the non-zero taps of its pixel, in order
    __shared__ int ntouched;
    // the two count channels' kernels [tap][c][co]: with the weights in global memory every step of the walk below was a dependent
    // cache round trip.  (Keeping only the locust channel's 8 KB here -- six workgroups per CU instead of four -- measured 101 against
    // 100 us: occupancy is not what binds this kernel.)
    __shared__ __attribute__((aligned(16))) float wl[64 * 2 * 32];
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 2 * 8; i += 256) {
        const int t = i >> 4, c = (i >> 3) & 1, j = i & 7;
        reinterpret_cast<float4 *>(wl)[i] = *reinterpret_cast<const float4 *>(w1 + (t * 3 + c) * 32 + j * 4);
    }
    if (tid >= 64 && tid < 64 + 81) vtab[0][tid - 64] = (float)((double)(tid - 64) / 80.0);
    if (tid >= 160 && tid < 160 + 11) vtab[1][tid - 160] = (float)((double)(tid - 160) / 10.0);
    for (int i = tid; i < 2 * (7056 / 4 + 4); i += 256) (&cnt[0][0])[i] = 0;
    // this lane's bin of the first env; inside the loop: of the next one
    int nbx = 255, nby = 0;
    unsigned nneed = 0xFFFFFFFFu;
    auto fetch = [&](int e) {
        if (e < nenv) {
            if (tid < 90) {
                const uint8_t *q = tid < 80 ? lbins + ((size_t)e * 80 + tid) * 2 : abins + ((size_t)e * 10 + (tid - 80)) * 2;
                nbx = q[0]; nby = q[1];
            }
            if (tid < 4 && nmask) nneed = nmask[(size_t)e * 4 + tid];
        }
    };
    fetch(blockIdx.x);
  for (int env = blockIdx.x; env < nenv; env += gridDim.x) {
    const int bx = nbx, by = nby;
    __syncthreads();      // the previous env's walk is over (evl, plist, pm, need); its increments are taken back
    if (tid < 4) need[tid] = nmask ? nneed : 0xFFFFFFFFu;
    if (tid >= 32 && tid < 45) pm[tid - 32] = 0;
    fetch(env + gridDim.x);
    __syncthreads();
    if (tid < 90) {
        if (bx != 255) {
            const int idx = bx * G + by;
            atomicAdd(&cnt[tid < 80 ? 0 : 1][idx >> 2], 1u << (8 * (idx & 3)));
#pragma unroll
            for (int c = 0; c < 4; ++c) {      // the four conv1 outputs whose window holds the bin
                const int oy = (bx >> 2) - (c >> 1), ox = (by >> 2) - (c & 1);
                if (oy >= 0 && oy < 20 && ox >= 0 && ox < 20) atomicOr(&pm[(oy * 20 + ox) >> 5], 1u << ((oy * 20 + ox) & 31));
            }
        }
    }
    __syncthreads();
    for (int p = tid; p < 400; p += 256)
        if ((pm[p >> 5] >> (p & 31)) & 1u) {
            int before = __popc(pm[p >> 5] & ((1u << (p & 31)) - 1u));
            for (int w = 0; w < (p >> 5); ++w) before += __popc(pm[w]);
            plist[before] = (unsigned short)p;
        }
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < 13; ++w) n += __popc(pm[w]);
        ntouched = n;
    }
    // the untouched pixels somebody reads: b1
    float4 *out = reinterpret_cast<float4 *>(sraw + (size_t)env * 12800);
    for (int item = tid; item < 3200; item += 256) {
        const int pix = item >> 3, oy = pix / 20, ox = pix - oy * 20, blk = (oy >> 1) * 10 + (ox >> 1);
        if (!((need[blk >> 5] >> (blk & 31)) & 1u) || ((pm[pix >> 5] >> (pix & 31)) & 1u)) continue;
        out[item] = *reinterpret_cast<const float4 *>(b1 + (item & 7) * 4);
    }
    __syncthreads();
    // A touched pixel gets half a wave.  Its 8 x 8 x 2 window is 32 count words: lane l reads word (ky = l >> 2, c = (l >> 1) & 1,
    // half = l & 1), the non-zero bytes of all lanes are listed in (ky, c, kx) order -- the order conv1_sparse_kernel adds them in --
    // through a shuffle scan of the per-lane counts, and the 32 lanes then walk the list together, one output channel each.
    const int co = tid & 31, hw = tid >> 5, nt = ntouched;
    unsigned short *ev = evl[hw];
    for (int t = hw; t < ((nt + 7) & ~7); t += 8) {      // whole waves run every round (the shuffles)
        const bool live = t < nt;
        const int pix = live ? plist[t] : 0, oy = pix / 20, ox = pix - oy * 20;
        const int ky = co >> 2, c = (co >> 1) & 1, half = co & 1;
        const unsigned int w = live ? cnt[c][(((4 * oy + ky) * G + 4 * ox) >> 2) + half] : 0u;      // rowbase is a multiple of 4: G = 84
        const int n = (w & 0xFFu ? 1 : 0) + (w & 0xFF00u ? 1 : 0) + (w & 0xFF0000u ? 1 : 0) + (w >> 24 ? 1 : 0);
        int incl = n;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const int o = __shfl_up(incl, d, 32);
            if (co >= d) incl += o;
        }
        const int total = __shfl(incl, 31, 32);
        int at = incl - n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned int k = (w >> (8 * j)) & 255u;
            if (k) ev[at++] = (unsigned short)((((ky * 8 + half * 4 + j) * 2 + c) << 7) | k);      // tap-and-channel row of wl, count
        }
        // (a half wave's LDS writes are visible to its own lanes after the wait the compiler places before the reads: same wave)
        float acc = b1[co];
        for (int e = 0; e < total; ++e) {
            const int code = ev[e];
            acc += vtab[(code >> 7) & 1][code & 127] * wl[(code >> 7) * 32 + co];
        }
        if (live) sraw[(size_t)env * 12800 + pix * 32 + co] = acc;
    }
    __syncthreads();      // every half wave has read its windows
    if (tid < 90 && bx != 255) {
        const int idx = bx * G + by;
        cnt[tid < 80 ? 0 : 1][idx >> 2] = 0u;      // several bins of a word all write the same zero
    }
  }
}

// ------------------------------------------------------------------------------------------ heads
// mu = tanh(p1 Wmu + b), sigma = sigmoid(p1 Wsg + b), vs = -scale*softplus(v2 Wv3 + b)
// (policy_v_network.py:40-59).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// kHeadRows rows per wave: each lane keeps its 8 + 4 weight rows in registers and has 3 * kHeadRows 16-byte loads in flight (one
// row per wave with 4-byte loads left the kernel at 2.4 TB/s: latency bound).
constexpr int kHeadRows = 4;
template <int A>
__global__ __launch_bounds__(256) void heads_forward_kernel(const float *__restrict__ p1, const float *__restrict__ v2,
                                                            const float *__restrict__ params, HeadOff o, int n, float scale,
                                                            float *__restrict__ mu, float *__restrict__ sigma, float *__restrict__ vs) {
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kHeadRows, lane = threadIdx.x & 63;
    if (row0 >= n) return;
    const float *muw = params + o.muw, *sgw = params + o.sgw, *v3w = params + o.v3w;
    // this lane's inputs: p1 columns 4 lane .. + 3 and 256 + 4 lane .. + 3, v2 columns 4 lane .. + 3
    float wm[8][A], ws[8][A];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = (j >> 2) * 256 + 4 * lane + (j & 3);
#pragma unroll
        for (int a = 0; a < A; ++a) { wm[j][a] = muw[k * A + a]; ws[j][a] = sgw[k * A + a]; }
    }
    // four scalar loads: v3_w sits at 1 026 A + a multiple of 4 floats in the flat parameter vector (tf.trainable_variables()
    // order), i.e. only 8-byte aligned for an odd A -- a 16-byte vector load there would be a misaligned access
    const float4 wv = make_float4(v3w[4 * lane], v3w[4 * lane + 1], v3w[4 * lane + 2], v3w[4 * lane + 3]);
    float4 x0[kHeadRows], x1[kHeadRows], xv[kHeadRows];
#pragma unroll
    for (int r = 0; r < kHeadRows; ++r) {
        const size_t row = (size_t)min(row0 + r, n - 1);      // clamped: the tail rows are loaded twice and not stored
        x0[r] = reinterpret_cast<const float4 *>(p1 + row * 512)[lane];
        x1[r] = reinterpret_cast<const float4 *>(p1 + row * 512 + 256)[lane];
        xv[r] = reinterpret_cast<const float4 *>(v2 + row * 256)[lane];
    }
    // lane 16 r + i keeps output i of row r (i < A: mu, < 2A: sigma, = 2A: value), so the transcendental functions of the four rows
    // run once per wave side by side instead of 2A + 1 of them in sequence on lane 0 of every row
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < kHeadRows; ++r) {
        const float x[8] = {x0[r].x, x0[r].y, x0[r].z, x0[r].w, x1[r].x, x1[r].y, x1[r].z, x1[r].w};
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float m = 0.f, sg = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { m += x[j] * wm[j][a]; sg += x[j] * ws[j][a]; }
            m = wave_sum(m);
            sg = wave_sum(sg);
            if (lane == 16 * r + a) mine = m;
            if (lane == 16 * r + A + a) mine = sg;
        }
        const float zv = wave_sum(xv[r].x * wv.x + xv[r].y * wv.y + xv[r].z * wv.z + xv[r].w * wv.w);
        if (lane == 16 * r + 2 * A) mine = zv;
    }
    const int r = lane >> 4, i = lane & 15, row = row0 + r;
    if (row >= n || i > 2 * A) return;
    if (i < A) {
        mu[(size_t)row * A + i] = tanhf(mine + params[o.mub + i]);
    } else if (i < 2 * A) {
        sigma[(size_t)row * A + (i - A)] = 1.0f / (1.0f + expf(-(mine + params[o.sgb + (i - A)])));
    } else {
        const float zv = mine + params[o.v3b];
        const float sp = zv > 20.f ? zv : log1pf(expf(zv));
        vs[row] = -scale * sp;
    }
}
#define GRL_HEADS_DISPATCH(A_, CALL)                                  \
    switch (A_) {                                                     \
        case 1: { constexpr int kA = 1; CALL; } break;                \
        case 2: { constexpr int kA = 2; CALL; } break;                \
        case 3: { constexpr int kA = 3; CALL; } break;                \
        default: { constexpr int kA = 4; CALL; } break;               \
    }

// a = mu + sigma * N(0,1) (paac.py:418), then SwarmRunner.transform_actions_for_env (emulator_runner.py:113-118)
// The same launch keeps the compact observation the action was chosen from -- states[t] = shared_states (paac.py:319): 200 bytes per
// env, copied word by word from the handle's observation (obs_src: lbins / abins / pos of the chunk's first env) into the rollout's
// step t -- and zeroes the done counter the env step that follows on this stream appends to (three copies and a memset until round 3).
struct ObsRecord {
    const uint32_t *lb, *ab, *pos;      // sources, already at the chunk's first env
    uint32_t *ro_lb, *ro_ab, *ro_pos;   // destinations, already at (step, first env)
    int nenv;
    int32_t *done_count;
};
__global__ void sample_actions_kernel(const float *__restrict__ mu, const float *__restrict__ sigma, int i0, int n, uint64_t seed,
                                      uint32_t env_off, uint32_t counter, float *__restrict__ raw, float *__restrict__ envact, ObsRecord R) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;      // samples [i0, i0 + n) of the batch
    if (R.done_count && i == 0) *R.done_count = 0;
    for (int w = i; w < R.nenv * 50; w += gridDim.x * blockDim.x) {      // 40 + 5 + 5 words per env
        if (w < R.nenv * 40) R.ro_lb[w] = R.lb[w];
        else if (w < R.nenv * 45) R.ro_ab[w - R.nenv * 40] = R.ab[w - R.nenv * 40];
        else R.ro_pos[w - R.nenv * 45] = R.pos[w - R.nenv * 45];
    }
    if (i >= n) return;
    i += i0;
    int env = i / 10, a = i - env * 10;
    double e0, e1;
    normal_pair(rng_block(seed, (uint32_t)env + env_off, counter, RS_ACTION, a), e0, e1);
    float2 m = reinterpret_cast<const float2 *>(mu)[i], s = reinterpret_cast<const float2 *>(sigma)[i];
    float2 r = make_float2((float)((double)m.x + (double)s.x * e0), (float)((double)m.y + (double)s.y * e1));
    reinterpret_cast<float2 *>(raw)[i] = r;
    float d = sqrtf(r.x * r.x + r.y * r.y);
    if (d >= 1.0f) { r.x /= d; r.y /= d; }
    reinterpret_cast<float2 *>(envact)[i] = r;
}

#include "net_shared.inc"
#include "net_patch.inc"

// ------------------------------------------------------------------------------------------ forward pass of one chunk
// floats kept per chunk slot: per-agent-trunk mode a3 + dense stack; shared mode a3sh (per env) + d3 + patch mask + dense stack,
// and at level 2 also what the gradient step reads of the per-env trunk: sraw, a2sh (per env), d2s, slot mask, ulist (per slot)
// What the gradient step reads of a chunk's index lists (slot_index, slot_sort, patch_sort, trunk_index): at keep level 3 the
// rollout's forward pass writes them into the chunk's resident slot and the gradient step finds them there, instead of running the 17
// index kernels of a chunk a second time (340 launches and ~3 ms of kernel time per 8 192-env update).  X(member, bytes).
#define GRL_IDX_LIST(X)                                                                                                             \
    X(perm, (size_t)net->ptiles * 256 * 4) X(sbeg, (size_t)net->pslices * 4) X(send, (size_t)net->pslices * 4)                      \
    X(sgrp, (size_t)net->pslices * 4) X(smask, c * 4) X(tmask, ((size_t)net->ptiles * 2 + 2) * 4) X(zmask, (size_t)net->pslices * 4) \
    X(tilegroup, (size_t)net->ptiles) X(org, c) X(sbase, (c + 1) * 4) X(rowagent, c * 9 * 4) X(stap, c * 9 * 2)                     \
    X(sperm, (c * 9 + 256) * 4) X(stmask, ((c * 9 + 255) / 256 + 1) * 4) X(tamask, ((c / 10) * 3 + 3) * 4)                          \
    X(tcmask, ((c / 10) * 2 + 2) * 4) X(trowlist, ((c / 10) * 81 + 256) * 4) X(tblklist, ((c / 10) * 100 + 256) * 4)               \
    X(tc3list, ((c / 10) * 49 + 256) * 4) X(trows_n, 16) X(tumask, 16)
static size_t idx_bytes_per_slot(const grl_net *net) {
    const size_t c = net->chunk;
    size_t b = 0;
#define X(m, bytes) b += ((size_t)(bytes) + 127) & ~(size_t)127;
    GRL_IDX_LIST(X)
#undef X
    return b;
}

static size_t keep_floats_per_slot(const grl_net *net, int level) {
    const size_t c = net->chunk, dense = 512 + 256 + 512 + 512 + 256 + 32;      // + the sign bits of d2 and v1 (12 words per sample, padded to 16)
    if (!net->shared_trunk) return c * (3136 + dense);
    size_t f = (c / 10) * 3136 + c * (1600 + 50 + dense);      // m3: 25 x 8 bytes per sample
    if (level >= 2) f += (c / 10) * (12800 + 5184) + c * (576 + 18) + ((c * 9 + 3) / 4 + 3) / 4 * 4;      // stays a multiple of 16 bytes
    if (level >= 3) f = ((f + 31) & ~(size_t)31) + idx_bytes_per_slot(net) / 4;      // every list on its own 128-byte line, the slot a multiple of 128 bytes
    return f;
}

// the activations the gradient step reads point into slot `slot` of the rollout-resident buffer, or at the chunk
// workspace for slot < 0
static void bind_activations(grl_net *net, long slot) {
    net->a3 = net->ws_a3; net->d1 = net->ws_d1; net->d2 = net->ws_d2; net->p1 = net->ws_p1; net->v1 = net->ws_v1; net->v2 = net->ws_v2;
    net->a3sh = net->ws_a3sh; net->d3 = net->ws_d3; net->m3 = net->ws_m3;
    net->mb_d2 = net->ws_mb; net->mb_v1 = net->ws_mb + (size_t)net->chunk * 4;
    net->sraw = net->ws_sraw; net->a2sh = net->ws_a2sh; net->d2s = net->ws_d2s; net->m2s = net->ws_m2s; net->ulist = net->ws_ulist;
    if (net->ws_idx[0]) {
        int k = 0;
#define X(m, bytes) net->m = static_cast<decltype(net->m)>(net->ws_idx[k++]);
        GRL_IDX_LIST(X)
#undef X
    }
    if (slot < 0 || !net->keep) return;
    const size_t c = net->chunk;
    float *b = net->keep + (size_t)slot * keep_floats_per_slot(net, net->keep_level);
    if (net->shared_trunk) {
        net->a3sh = b; b += (c / 10) * 3136;
        net->d3 = b; b += c * 1600;
        net->m3 = reinterpret_cast<unsigned long long *>(b); b += c * 50;
    } else {
        net->a3 = b; b += c * 3136;
    }
    net->d1 = b; b += c * 512;
    net->d2 = b; b += c * 256;
    net->p1 = b; b += c * 512;
    net->v1 = b; b += c * 512;
    net->v2 = b; b += c * 256;
    net->mb_d2 = reinterpret_cast<unsigned long long *>(b); net->mb_v1 = net->mb_d2 + c * 4; b += c * 32;      // 12 words per sample + 8 words of padding: the slot stays a multiple of 128 bytes
    if (net->shared_trunk && net->keep_level >= 2) {
        net->sraw = b; b += (c / 10) * 12800;
        net->a2sh = b; b += (c / 10) * 5184;
        net->d2s = b; b += c * 576;
        net->m2s = reinterpret_cast<unsigned long long *>(b); b += c * 18;      // 9 words per sample
        net->ulist = reinterpret_cast<signed char *>(b); b += ((c * 9 + 3) / 4 + 3) / 4 * 4;
    }
    if (net->shared_trunk && net->keep_level >= 3) {
        const size_t used = (size_t)(b - (net->keep + (size_t)slot * keep_floats_per_slot(net, net->keep_level)));
        char *q = reinterpret_cast<char *>(b + (((used + 31) & ~(size_t)31) - used));
#define X(m, bytes) net->m = reinterpret_cast<decltype(net->m)>(q); q += ((size_t)(bytes) + 127) & ~(size_t)127;
        GRL_IDX_LIST(X)
#undef X
    }
}

// reuse_tail: a3..v2 of this chunk are already resident (bind_activations); only the cheap per-env trunk, the
// per-agent a2 and the heads are re-evaluated
static int forward_chunk(grl_net *net, const uint8_t *lb, const uint8_t *ab, const uint8_t *pos, int nenv, float *mu,
                         float *sigma, float *vs, bool reuse_tail = false, bool skip_heads = false) {
    hipStream_t st = net->h->stream;
    const float *P = net->params, *PT = net->paramsT;
    const int n = nenv * 10;
    net->last_n = n;
    net->pitem_on_side = 0;
    if (net->shared_trunk) {
        // gradient step on resident activations: at level 2 the trunk tensors are resident too, only the group sort reruns
        int rc;
        if (reuse_tail && net->keep_level >= 3) {
            // the rollout's index lists are resident with its activations (bind_activations): nothing to rebuild
            if ((rc = slot_prof(net, n))) return rc;
            rc = patch_prof(net);
        } else if (reuse_tail && net->keep_level >= 2) {
            if ((rc = slot_index(net, pos, n))) return rc;
            if (net->trunk_skip && net->expand2_gemm && (rc = trunk_index(net, lb, ab, pos, nenv))) return rc;      // the gradient step's row lists
            rc = patch_sort(net, n);
        } else {
            rc = forward_conv12_shared(net, lb, ab, pos, nenv, !reuse_tail);
        }
        if (rc) return rc;
    } else {
    net->prof_tag_cur = PT_PER_AGENT;
    hipLaunchKernelGGL(conv1_sparse_kernel, dim3(nenv), dim3(256), 0, st, lb, ab, pos, P + ConvOffsets::c1w, P + ConvOffsets::c1b,
                       net->a1, net->h->cfg.grid_size, (float *)nullptr);
    {
        GatherConv2 g{net->a1, n * 81};
        EpiBiasAct e{net->a2, 64, P + ConvOffsets::c2b, ACT_RELU};
        GemmTimer t(net, 2.0 * n * 81 * 512 * 64);
        launch_rowk<256, 64, kW256M, kW256N, GatherConv2, EpiBiasAct>(net, dim3(1, (n * 81 + 255) / 256), st, g, PT + ConvOffsets::c2w, 512, 64, e);
    }
    }
    if (!reuse_tail) {
    if (net->shared_trunk) {
        if (int rc = forward_conv3_dense1_shared(net, nenv, pos)) return rc;
    } else {
        GatherConv3 g{net->a2, n * 49};
        EpiBiasAct e{net->a3, 64, P + ConvOffsets::c3b, ACT_RELU};
        GemmTimer t(net, 2.0 * n * 49 * 576 * 64);
        launch_rowk<256, 64, kW256M, kW256N, GatherConv3, EpiBiasAct>(net, dim3(1, (n * 49 + 255) / 256), st, g, PT + ConvOffsets::c3w, 576, 64, e);
    }
    net->prof_tag_cur = net->shared_trunk ? PT_DENSE_FWD : PT_PER_AGENT;
    // bits != nullptr: the epilogue also keeps the sign bits of what it stores (the data gradient above reads them instead of the tensor)
    auto dense = [&](const float *in, int K, const float *w, const float *b, int N, float *out, unsigned long long *bits) {
        DenseRows g{in, n, K, K};
        GemmTimer t(net, 2.0 * n * K * N);
        if (bits) {
            EpiBiasActBits e{out, N, b, bits, N >> 6};
            launch_rowk<128, 128, kW128M, kW128N, DenseRows, EpiBiasActBits>(net, dim3(N / 128, (n + 127) / 128), st, g, w, K, N, e);
        } else {
            EpiBiasAct e{out, N, b, ACT_RELU};
            launch_rowk<128, 128, kW128M, kW128N, DenseRows, EpiBiasAct>(net, dim3(N / 128, (n + 127) / 128), st, g, w, K, N, e);
        }
    };
    if (!net->shared_trunk) dense(net->a3, 3136, PT + ConvOffsets::d1w, P + ConvOffsets::d1b, 512, net->d1, nullptr);
    dense(net->d1, 512, PT + ConvOffsets::d2w, P + ConvOffsets::d2b, 256, net->d2, net->mb_d2);
    {   // pol1 and v1 in one launch: N = 1024 over [pol1_w | v1_w] (A tile d2 staged once, one launch less per forward pass)
        DenseRows g{net->d2, n, 256, 256};
        GemmTimer t(net, 2.0 * n * 256 * 1024);
        EpiBiasActSplit e{net->p1, net->v1, 512, 512, P + ConvOffsets::p1b, P + net->ho.v1b, net->mb_v1, 8};
        launch_rowk<128, 128, kW128M, kW128N, DenseRows, EpiBiasActSplit, true, true, 1, true>(net, dim3(1024 / 128, (n + 127) / 128), st, g, net->wpvT, 256, 1024, e);
    }
    dense(net->v1, 512, PT + net->ho.v2w, P + net->ho.v2b, 256, net->v2, nullptr);
    }
    if (!skip_heads)      // (the gradient step over a resident rollout has the heads' outputs of every step already)
    GRL_HEADS_DISPATCH(net->ho.A, hipLaunchKernelGGL(heads_forward_kernel<kA>, dim3((n + 4 * kHeadRows - 1) / (4 * kHeadRows)), dim3(256), 0, st, net->p1, net->v2, P, net->ho, n,
                                                      net->cfg.scale, mu, sigma, vs));
    NET_HIP(net, hipGetLastError());
    return GRL_OK;
}

// forward over n_envs envs in chunks; obs pointers are DEVICE pointers; outputs device (B,2)(B,2)(B)
static int forward_all(grl_net *net, const uint8_t *lb, const uint8_t *ab, const uint8_t *pos, int n_envs, float *mu, float *sigma,
                       float *vs, long slot0 = -1) {
    const int ce = net->chunk / 10;
    int rc = lanes_fork(net);
    if (rc) return rc;
    const int nl = lanes_active(net);
    for (int e0 = 0; e0 < n_envs; e0 += ce) {
        int ne = n_envs - e0 < ce ? n_envs - e0 : ce;
        use_lane(net, (e0 / ce) % nl);      // independent chunks alternate between the two streams
        net->last_lane = net->cur_lane;
        bind_activations(net, slot0 < 0 ? -1 : slot0 + e0 / ce);
        rc = forward_chunk(net, lb + (size_t)e0 * 160, ab + (size_t)e0 * 20, pos + (size_t)e0 * 20, ne,
                           mu + (size_t)e0 * 10 * net->ho.A, sigma + (size_t)e0 * 10 * net->ho.A, vs + (size_t)e0 * 10);
        if (rc) { (void)lanes_join(net); return rc; }
    }
    return lanes_join(net);
}

// forward-pass workspace of the lane currently loaded into *net
static int alloc_lane_forward(grl_net *n) {
    const size_t c = n->chunk;
    int rc = GRL_OK;
    auto A = [&](float **p, size_t cnt) { if (rc == GRL_OK) rc = nalloc(n, p, cnt); };
    A(&n->grads, n->ho.total);
    // a2: the per-agent conv2 tensor of the dense form; in shared-trunk mode the workspace of conv3's gather form (net_patch.inc:
    // canonical cell blocks of ctiles * 256 rows, then the items' descriptors, tap masks, tile masks and sort counters)
    const size_t gather_ws = (size_t)n->ctiles * 256 * 576 + (c * 25 + 256) * 4 + (c * 25 + 1024) + 2 * ((c + 255) / 256 + 1) * 64 + 4096;
    A(&n->a2, std::max(c * 5184, gather_ws)); A(&n->d1, c * 512); A(&n->d2, c * 256); A(&n->p1, c * 512); A(&n->v1, c * 512); A(&n->v2, c * 256);
    if (!n->shared_trunk) { A(&n->a1, c * 12800); A(&n->a3, c * 3136); }       // per-agent tensors the shared evaluation never forms
    A(&n->sraw, (c / 10) * 12800); A(&n->z2sh, (c / 10) * 5184);
    A(&n->a2sh, (c / 10) * 5184); A(&n->d2s, c * 9 * 64); if (rc == GRL_OK) rc = nalloc(n, &n->m2s, c * 9); A(&n->z3sh, (c / 10) * 3136);
    if (rc == GRL_OK) rc = nalloc(n, &n->ulist, c * 9);
    A(&n->a3sh, (c / 10) * 3136); A(&n->d3, c * 1600); if (rc == GRL_OK) rc = nalloc(n, &n->m3, c * 25); A(&n->ysh, (c / 10) * 512);
    n->ws_a3 = n->a3; n->ws_d1 = n->d1; n->ws_d2 = n->d2; n->ws_p1 = n->p1; n->ws_v1 = n->v1; n->ws_v2 = n->v2;
    n->ws_a3sh = n->a3sh; n->ws_d3 = n->d3; n->ws_m3 = n->m3;
    if (rc == GRL_OK) rc = nalloc(n, &n->ws_mb, c * 12);
    n->mb_d2 = n->ws_mb; n->mb_v1 = n->ws_mb + c * 4;
    n->ws_sraw = n->sraw; n->ws_a2sh = n->a2sh; n->ws_d2s = n->d2s; n->ws_m2s = n->m2s; n->ws_ulist = n->ulist;
    if (rc == GRL_OK) rc = nalloc(n, &n->perm, (size_t)n->ptiles * 256);
    if (rc == GRL_OK) rc = nalloc(n, &n->goffp, 32);
    if (rc == GRL_OK) rc = nalloc(n, &n->sbeg, n->pslices);
    if (rc == GRL_OK) rc = nalloc(n, &n->send, n->pslices);
    if (rc == GRL_OK) rc = nalloc(n, &n->sgrp, n->pslices);
    if (rc == GRL_OK) rc = nalloc(n, &n->blkcnt, ((c + 255) / 256) * PATCH_KEYS);
    if (rc == GRL_OK) rc = nalloc(n, &n->blkoff, ((c + 255) / 256) * PATCH_KEYS);
    if (rc == GRL_OK) rc = nalloc(n, &n->binbase, PATCH_KEYS);
    if (rc == GRL_OK) rc = nalloc(n, &n->pkey, c);
    if (rc == GRL_OK) rc = nalloc(n, &n->prank, c);
    if (rc == GRL_OK) rc = nalloc(n, &n->smask, c);
    if (rc == GRL_OK) rc = nalloc(n, &n->tmask, (size_t)n->ptiles * 2 + 2);
    if (rc == GRL_OK) rc = nalloc(n, &n->zmask, (size_t)n->pslices);
    if (rc == GRL_OK) rc = nalloc(n, &n->wmask, c);
    if (rc == GRL_OK) rc = nalloc(n, &n->slot_of, c);
    if (rc == GRL_OK) rc = nalloc(n, &n->tilegroup, (size_t)n->ptiles);
    if (rc == GRL_OK) rc = nalloc(n, &n->org, c);
    if (rc == GRL_OK) rc = nalloc(n, &n->sbase, c + 1);
    if (rc == GRL_OK) rc = nalloc(n, &n->rowagent, c * 9);
    if (rc == GRL_OK) rc = nalloc(n, &n->rowdesc, c * 9);
    if (rc == GRL_OK) rc = nalloc(n, &n->sblk, 1024);
    if (rc == GRL_OK) rc = nalloc(n, &n->skey, c * 9);
    if (rc == GRL_OK) rc = nalloc(n, &n->srank, c * 9);
    if (rc == GRL_OK) rc = nalloc(n, &n->stap, c * 9);
    if (rc == GRL_OK) rc = nalloc(n, &n->sblkcnt, ((c * 9 + 1023) / 1024 + 1) * 25);
    if (rc == GRL_OK) rc = nalloc(n, &n->sblkoff, ((c * 9 + 1023) / 1024 + 1) * 25);
    if (rc == GRL_OK) rc = nalloc(n, &n->sbinbase, 32);
    if (rc == GRL_OK) rc = nalloc(n, &n->sperm, c * 9 + 256);
    if (rc == GRL_OK) rc = nalloc(n, &n->stmask, (c * 9 + 255) / 256 + 1);
    if (rc == GRL_OK) rc = nalloc(n, &n->szmask, 1024);
    if (rc == GRL_OK) rc = nalloc(n, &n->tamask, (c / 10) * 3 + 3);
    if (rc == GRL_OK) rc = nalloc(n, &n->tbmask, (c / 10) * 4 + 4);
    if (rc == GRL_OK) rc = nalloc(n, &n->tcmask, (c / 10) * 2 + 2);
    if (rc == GRL_OK) rc = nalloc(n, &n->tnmask, (c / 10) * 4 + 4);
    if (rc == GRL_OK) rc = nalloc(n, &n->trowlist, (c / 10) * 81 + 256);
    if (rc == GRL_OK) rc = nalloc(n, &n->tblklist, (c / 10) * 100 + 256);
    if (rc == GRL_OK) rc = nalloc(n, &n->tc3list, (c / 10) * 49 + 256);
    if (rc == GRL_OK) rc = nalloc(n, &n->trows_n, 4);
    if (rc == GRL_OK) rc = nalloc(n, &n->twgcnt, ((c / 10 + TRUNK_ENVS - 1) / TRUNK_ENVS + 1) * 3);
    if (rc == GRL_OK) rc = nalloc(n, &n->twgoff, ((c / 10 + TRUNK_ENVS - 1) / TRUNK_ENVS + 1) * 3);
    if (rc == GRL_OK) rc = nalloc(n, &n->tumask, 4);
    if (rc == GRL_OK) rc = nalloc(n, &n->tneed2, 4);
    if (rc == GRL_OK) rc = nalloc(n, &n->tubias, 512);
    if (rc == GRL_OK) rc = nalloc(n, &n->tug, 512);
    if (rc == GRL_OK) rc = nalloc(n, &n->tuspix, 49 * 64);
    if (rc == GRL_OK) rc = nalloc(n, &n->tslab, 2 * 2048 * 64);
    if (rc == GRL_OK) rc = nalloc(n, &n->tsums, 256);
    A(&n->carow, c * 128);
    if (rc == GRL_OK) rc = nalloc(n, &n->cperm, (size_t)n->ctiles * 256);
    if (rc == GRL_OK) rc = nalloc(n, &n->cblkcnt, ((c + 255) / 256) * 4);
    if (rc == GRL_OK) rc = nalloc(n, &n->cblkoff, ((c + 255) / 256) * 4);
    if (rc == GRL_OK) rc = nalloc(n, &n->cgoff, 8);
    if (rc == GRL_OK) rc = nalloc(n, &n->cslot, c * 9);
    if (rc == GRL_OK) rc = nalloc(n, &n->cinv, c);
    if (rc == GRL_OK) rc = nalloc(n, &n->ctilegroup, (size_t)n->ctiles);
    if (rc == GRL_OK) {
        grl_net *net = n;
        int k = 0;
#define X(m, bytes) net->ws_idx[k++] = (void *)net->m;
        GRL_IDX_LIST(X)
#undef X
        static_assert(sizeof(((NetLane *)nullptr)->ws_idx) / sizeof(void *) >= 21, "ws_idx holds every list of GRL_IDX_LIST");
    }
    return rc;
}

static int ensure_tmp_obs(grl_net *net, int n_envs) {
    if (net->tmp_envs >= n_envs) return GRL_OK;
    int rc;
    if ((rc = nalloc(net, &net->tmp_lb, (size_t)n_envs * 160))) return rc;
    if ((rc = nalloc(net, &net->tmp_ab, (size_t)n_envs * 20))) return rc;
    if ((rc = nalloc(net, &net->tmp_pos, (size_t)n_envs * 20))) return rc;
    float *m, *s, *v;
    if ((rc = nalloc(net, &m, (size_t)n_envs * 10 * net->ho.A))) return rc;
    if ((rc = nalloc(net, &s, (size_t)n_envs * 10 * net->ho.A))) return rc;
    if ((rc = nalloc(net, &v, (size_t)n_envs * 10))) return rc;
    net->mu = m; net->sigma = s; net->vs = v;
    net->tmp_envs = n_envs;
    return GRL_OK;
}

}  // namespace grl

using namespace grl;

// a workgroup that waits `ticks` of the 100 MHz constant clock (side_stream_pick)
__global__ void grl_wait_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// picks a stream that runs BESIDE lane 0 (see grl_net_create) into n->side_stream[0], or leaves it null
static int side_stream_pick(grl_net *n) {
    hipStream_t lane = n->lane_stream[0];
    hipEvent_t e0 = nullptr, e1 = nullptr, ec = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventCreateWithFlags(&ec, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return GRL_E_HIP;
    }
    const long long ticks = 6000;      // 60 us
    hipStream_t cand[5] = {};
    int pick = -1;
    for (int c = 0; c < 5 && pick < 0; ++c) {
        if (hipStreamCreateWithFlags(&cand[c], hipStreamNonBlocking) != hipSuccess) { cand[c] = nullptr; break; }
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {      // the shortest of three: a wait can only be lengthened by others' work
            (void)hipStreamSynchronize(lane);
            (void)hipStreamSynchronize(cand[c]);
            (void)hipEventRecord(e0, lane);
            hipLaunchKernelGGL(grl_wait_kernel, dim3(1), dim3(64), 0, lane, ticks);
            hipLaunchKernelGGL(grl_wait_kernel, dim3(1), dim3(64), 0, cand[c], ticks);
            (void)hipEventRecord(ec, cand[c]);
            (void)hipStreamWaitEvent(lane, ec, 0);
            (void)hipEventRecord(e1, lane);
            (void)hipStreamSynchronize(lane);
            float ms = 1e9f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best) best = ms;
        }
        if (best < 0.095f) pick = c;      // 60 us side by side; 120 one behind the other
    }
    for (int c = 0; c < 5; ++c)
        if (cand[c] && c != pick) { (void)hipStreamSynchronize(cand[c]); (void)hipStreamDestroy(cand[c]); }
    if (pick >= 0) n->side_stream[0] = cand[pick];
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ec);
    (void)hipGetLastError();
    return GRL_OK;
}

extern "C" {

int grl_net_config_default(int32_t kind, grl_net_config *cfg) {
    if (!cfg || kind != GRL_NET_CONV_SINGLE_AGENT) return GRL_E_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(grl_net_config);
    cfg->kind = kind;
    cfg->max_chunk_samples = 40960;
    cfg->scale = 1000.f;          // train_paac_conv.py:113
    cfg->entropy_beta = 0.02f;    // :103
    cfg->clip_norm = 40.f;        // :104
    cfg->gamma = 0.99f;           // :106
    cfg->num_actions = 2;         // SwarmEnvironmentCreator.num_actions (environment_creator.py:17-20)
    return GRL_OK;
}

static int range_flag_init(grl_net *n);

int grl_net_create(grl_handle *h, const grl_net_config *cfg, grl_net **out) {
    if (!h || !cfg || !out) return GRL_E_INVALID;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(grl_net_config)) return fail(h, GRL_E_INVALID, "grl_net_create: grl_net_config size mismatch");
    if (cfg->kind != GRL_NET_CONV_SINGLE_AGENT) return fail(h, GRL_E_INVALID, "grl_net_create: unknown kind");
    if (h->cfg.env_kind != GRL_ENV_SWARM || h->cfg.grid_size != 84)
        return fail(h, GRL_E_INVALID, "grl_net_create: the conv net needs a Swarm handle with grid_size 84 (84x84x3 input)");
    if (cfg->max_chunk_samples < 10 || cfg->max_chunk_samples % 10 || cfg->max_chunk_samples > 131072)
        return fail(h, GRL_E_INVALID, "grl_net_create: max_chunk_samples must be a multiple of 10 in 10..131072");
    if (cfg->num_actions < 1 || cfg->num_actions > GRL_MAX_ACTIONS)
        return fail(h, GRL_E_INVALID, "grl_net_create: num_actions must be in 1..4");
    hipSetDevice(h->cfg.device_id);
    grl_net *n = new grl_net();
    n->h = h; n->cfg = *cfg; n->chunk = cfg->max_chunk_samples; n->adam_t = 0;
    n->ho = head_offsets(cfg->num_actions);
    n->T = 0; n->B = 0; n->tmp_envs = 0; n->act_counter = 0; n->prof_on = false; n->prof_used = 0; n->prof_flops = 0; n->last_n = 0; n->prof_tag_cur = 0;
    n->ro_lb = nullptr; n->slab_floats = 0; n->slab_used = 0; n->slabb_floats = 0; n->slabb_used = 0; n->w3t = n->w2t = nullptr;
    n->mu = n->sigma = n->vs = nullptr;
    n->keep_level = 0;
    n->ar_ev0 = n->ar_ev1 = nullptr; n->ar_pending = 0; n->ar_calls = 0; n->ht_rollouts = n->ht_updates = 0; n->ht_rollout_ms = n->ht_train_enq_ms = n->ht_train_wait_ms = 0.0; n->ht_train_t0 = 0.0; n->ar_ms_total = 0.0; n->ar_ms_last = 0.f;
    n->keep = nullptr; n->keep_slots = 0; n->param_version = 0; n->keep_version = -1;
    n->shared_trunk = (cfg->reserved & GRL_NET_F_PER_AGENT_TRUNK) ? 0 : 1;     // the plain per-agent evaluation is the A/B reference
    n->cur_lane = 0; n->last_lane = 0;
    for (int k = 0; k < GRL_MAX_LANES; ++k) { n->lane_stream[k] = nullptr; n->ev_join[k] = nullptr; n->side_stream[k] = nullptr; n->ev_side0[k] = n->ev_side1[k] = n->ev_side2[k] = nullptr; }
    n->idx_side = 0; n->side_now = 0; n->pitem_on_side = 0;
    n->lane_stream[0] = h->stream; n->ev_fork = nullptr; n->nlanes = 1;
    const size_t c = n->chunk;
    n->ptiles = (int)((c + 255) / 256) + 9;
    n->ctiles = (int)((c + 255) / 256) + 4;
    {   // A/B switch of the conv2 corrections (default: GEMM)
        const char *lsk = getenv("GRL_NET_LOSS_SCALE");      // "off": S = 1, to show what the scale is for (tests)
        n->loss_scale_on = (lsk && strcmp(lsk, "off") == 0) ? 0 : 1;
        const char *gm = getenv("GRL_NET_GEMM"), *rf = getenv("GRL_NET_RANGE_FALLBACK");
        n->gemm_f32 = (gm && strcmp(gm, "f32") == 0) ? 1 : 0;                    // force the fp32-MFMA form from the start (tests, A/B)
        n->range_fallback_on = (rf && strcmp(rf, "off") == 0) ? 0 : 1;            // off: a range violation fails the call (GRL_E_RANGE) and nothing else
        n->range_fallbacks = 0; n->range_bits_last = 0; n->update_skipped_last = 0;
        n->f32_by_fallback = 0; n->f32_clean_passes = 0; n->range_returns = 0; n->absmax_last = 0.f;
        n->range_return_k = 4;
        if (const char *rr = getenv("GRL_NET_RANGE_RETURN")) n->range_return_k = strcmp(rr, "off") == 0 ? 0 : std::max(0, atoi(rr));
        const char *psk = getenv("GRL_PATCH_SKIP");
        n->patch_skip = (psk && strcmp(psk, "off") == 0) ? 0 : 1;
        const char *tsk = getenv("GRL_TRUNK_SKIP");
        n->trunk_skip = (tsk && strcmp(tsk, "off") == 0) ? 0 : 1;
        n->pfrac[0] = n->pfrac[1] = n->pfrac[2] = 1.0; n->sfrac = 1.0;
        const char *e2 = getenv("GRL_NET_EXPAND2");
        n->expand2_gemm = (e2 && strcmp(e2, "lds") == 0) ? 0 : 1;
        { const char *e3 = getenv("GRL_NET_EXPAND3"); n->expand3_gather = (e3 && strcmp(e3, "prod") == 0) ? 0 : 1;
          const char *a1 = getenv("GRL_NET_ACC1"); n->acc1 = (a1 && strcmp(a1, "off") == 0) ? 0 : 1; }
    }
    n->pslice_rows = PATCH_SLICE_ROWS; n->pwgrad_xcd = 1;      // a row slice per XCD: A and B of a slice fetched once (under the support masks and the lighter traffic of round 3 this order wins by 0.7 %; 2 was the choice for the plain patch)
    if (const char *e = getenv("GRL_PATCH_SLICE")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024 || v == 2048 || v == 4096 || v == 8192) n->pslice_rows = v; }
    if (const char *e = getenv("GRL_PATCH_WGRAD_XCD")) n->pwgrad_xcd = atoi(e);
    n->pwgrad_pair = 1;      // 128 x 128 tiles over pairs of live patch pixels (PatchRowsPair, net_gemm.h); off: the 64 x 128 per-pixel tiles of round 3 (A/B, equality test)
    if (const char *e = getenv("GRL_PATCH_WGRAD_PAIR")) n->pwgrad_pair = strcmp(e, "off") != 0 && strcmp(e, "0") != 0;
    n->pdgrad_pair = 1;      // the same for the data gradient's N axis (PatchRowsPairN): 128 x 128 tiles on eight waves; off: 256 x 64 per pixel
    if (const char *e = getenv("GRL_PATCH_DGRAD_PAIR")) n->pdgrad_pair = strcmp(e, "off") != 0 && strcmp(e, "0") != 0;
    n->swgrad_pair = 1;      // conv3's slot weight gradient on 128 x 64 tiles of two live taps (SlotGatherT3PPair); off: 64 x 64 per tap
    if (const char *e = getenv("GRL_SLOT_WGRAD_PAIR")) n->swgrad_pair = strcmp(e, "off") != 0 && strcmp(e, "0") != 0;
    n->slots_xcd = 1;        // the sorted gather GEMMs keep a contiguous range of row tiles on one XCD (gemm_rowk, XCD_ORDER = 2); GRL_SLOTS_XCD=off: round-robin tiles
    if (const char *e = getenv("GRL_SLOTS_XCD")) n->slots_xcd = strcmp(e, "off") != 0 && strcmp(e, "0") != 0;
    n->swgrad_chunks = 0;    // tuning knob: row ranges of that launch (0: the default of the call site)
    if (const char *e = getenv("GRL_SLOT_WGRAD_CHUNKS")) n->swgrad_chunks = std::max(0, atoi(e));
    n->pdgrad_xcd = 1;      // M tile per XCD (A fetched once): 40.6 -> 38.5 ms per update at 81 920-sample chunks; spread (0) was faster at 40 960
    n->tn_wgs = 1024; n->tn_wgs_dense = 512;
    if (const char *e = getenv("GRL_TN_WGS")) { const int v = atoi(e); if (v >= 256 && v <= 4096) n->tn_wgs = v; }
    if (const char *e = getenv("GRL_TN_WGS_DENSE")) { const int v = atoi(e); if (v >= 128 && v <= 4096) n->tn_wgs_dense = v; }
    if (const char *e = getenv("GRL_PATCH_DGRAD_XCD")) n->pdgrad_xcd = atoi(e) != 0;
    n->pslices = (int)((c + n->pslice_rows - 1) / n->pslice_rows) + 9;
    n->npad = (int)((c + 255) / 256 * 256);
    int rc = GRL_OK;
    auto A = [&](float **p, size_t cnt) { if (rc == GRL_OK) rc = nalloc(n, p, cnt); };
    A(&n->params, n->ho.total); A(&n->paramsT, n->ho.total); A(&n->adam_m, n->ho.total); A(&n->adam_v, n->ho.total);
    A(&n->w3f, 576 * 64); A(&n->stats, 16); A(&n->wpvT, (size_t)1024 * 256); A(&n->w2corr, 4 * 576 * 128);
    A(&n->tbgimg, 12800); A(&n->tbgz, 128); A(&n->tbgimg3, 5184); A(&n->tbgz3, 128); A(&n->tybg, 49 * 512);
    if (rc == GRL_OK) rc = nalloc(n, &n->tbglist, 4);
    int nlanes = (cfg->reserved & GRL_NET_F_SINGLE_STREAM) ? 1 : 4;      // measured at 32 768 envs: 1.41 / 1.22 / 1.16 / 1.13 / 1.14 s per update with 1 / 2 / 3 / 4 / 8
    if (const char *env = getenv("GRL_NET_LANES")) { int v = atoi(env); if (v >= 1 && v <= GRL_MAX_LANES) nlanes = v; }      // tuning knob
    for (int k = 0; k < nlanes && rc == GRL_OK; ++k) {       // lane k's forward workspace (allocated into *n, then parked)
        static_cast<NetLane &>(*n) = NetLane{};
        rc = alloc_lane_forward(n);
        n->lanes[k] = static_cast<NetLane &>(*n);
    }
    static_cast<NetLane &>(*n) = n->lanes[0];
    if (rc == GRL_OK && nlanes > 1) {
        bool ok = hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int k = 1; k < nlanes && ok; ++k)
            ok = hipStreamCreateWithFlags(&n->lane_stream[k], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&n->ev_join[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) rc = nfail(n, GRL_E_HIP, "creating the lane streams/events failed");
    }
    if (rc == GRL_OK) n->nlanes = nlanes;
    // The index side stream of lane 0 (used by a rollout with ONE chunk per step: grl_net_rollout; not in the single-stream profiling
    // configuration; GRL_NET_IDX_SIDE=off: never).  It must sit on ANOTHER hardware queue than lane 0: the runtime deals the streams of
    // a process onto four queues by their use counts, so after other handles and nets have come and gone a new stream may land on
    // the lane's own queue -- the side work then serialises with the lane and the events cost on top (49.3 instead of 43.4 ms per
    // 4 096-env update inside bench.py's shard leg, where lane 0 and its side stream shared queue 3; without it 45.7).  The HIP API does
    // not name a stream's queue, so up to five candidates are TIMED against the lane -- two 60 us waits side by side take 60 us on two
    // queues and 120 on one -- and the first that runs beside it is kept.  (Streams of another priority, or more than four queues,
    // are no way out: kernels of extra queues of one process interleave with ~40 us per 5 us kernel: 72 and 61 ms.)
    {
        const char *e = getenv("GRL_NET_IDX_SIDE");
        const bool want = !(cfg->reserved & GRL_NET_F_SINGLE_STREAM) && !(e && (!strcmp(e, "off") || !strcmp(e, "0"))) && rc == GRL_OK;
        if (want && side_stream_pick(n) == GRL_OK && n->side_stream[0]) {
            const bool ok = hipEventCreateWithFlags(&n->ev_side0[0], hipEventDisableTiming) == hipSuccess &&
                            hipEventCreateWithFlags(&n->ev_side1[0], hipEventDisableTiming) == hipSuccess &&
                            hipEventCreateWithFlags(&n->ev_side2[0], hipEventDisableTiming) == hipSuccess;
            n->idx_side = ok ? 1 : 0;
        }
        (void)hipGetLastError();
    }
    if (rc == GRL_OK && hipFuncSetAttribute((const void *)expand_conv2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)EXP2_LDS_BYTES) != hipSuccess)
        rc = nfail(n, GRL_E_HIP, "hipFuncSetAttribute(expand_conv2_kernel)");
    if (rc == GRL_OK) rc = ensure_tmp_obs(n, h->E);
    if (rc == GRL_OK) rc = range_flag_init(n);
    if (rc != GRL_OK) {
        fail(h, rc, "grl_net_create: " + n->err);
        grl_net_destroy(n);
        return rc;
    }
    hipStreamSynchronize(h->stream);
    *out = n;
    return GRL_OK;
}

int grl_net_destroy(grl_net *n) {
    if (!n) return GRL_OK;
    const bool alive = grl_handle_alive(n->h);      // the handle may have been destroyed first (finaliser order of a host binding)
    grl_sync_for_destroy(n->h);
    if (n->comm) {
        ncclCommDestroy((ncclComm_t)n->comm);
        (void)hipGetLastError();   // RCCL teardown may leave a stale HIP error on this thread
    }
    if (alive) use_lane(n, 0);      // restores the handle's stream
    for (int k = 1; k < GRL_MAX_LANES; ++k) {
        if (n->lane_stream[k]) { hipStreamSynchronize(n->lane_stream[k]); hipStreamDestroy(n->lane_stream[k]); }
        if (n->ev_join[k]) hipEventDestroy(n->ev_join[k]);
    }
    for (int k = 0; k < GRL_MAX_LANES; ++k) {
        if (n->side_stream[k]) { hipStreamSynchronize(n->side_stream[k]); hipStreamDestroy(n->side_stream[k]); }
        if (n->ev_side0[k]) hipEventDestroy(n->ev_side0[k]);
        if (n->ev_side1[k]) hipEventDestroy(n->ev_side1[k]);
        if (n->ev_side2[k]) hipEventDestroy(n->ev_side2[k]);
    }
    if (n->ev_fork) hipEventDestroy(n->ev_fork);
    if (n->ar_ev0) { hipEventDestroy(n->ar_ev0); hipEventDestroy(n->ar_ev1); }
    for (void *p : n->allocs) hipFree(p);
    if (n->keep) hipFree(n->keep);
    for (hipEvent_t ev : n->prof_ev) hipEventDestroy(ev);
    delete n;
    return GRL_OK;
}

const char *grl_net_last_error(const grl_net *n) { return n ? n->err.c_str() : ""; }
int64_t grl_net_num_params(const grl_net *n) { return n ? n->ho.total : 0; }

int grl_net_set_params(grl_net *n, const float *host, int64_t cnt) {
    if (!n || !host) return GRL_E_INVALID;
    if (cnt != n->ho.total) return nfail(n, GRL_E_SIZE, "grl_net_set_params: expected " + std::to_string((long)n->ho.total) + " floats");
    hipSetDevice(n->h->cfg.device_id);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    NET_HIP(n, hipMemcpy(n->params, host, cnt * 4, hipMemcpyHostToDevice));
    n->param_version += 1;
    refresh_transposes(n);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    return GRL_OK;
}

static int get_flat(grl_net *n, const float *src, float *host, int64_t cnt) {
    if (!n || !host) return GRL_E_INVALID;
    if (cnt != n->ho.total) return nfail(n, GRL_E_SIZE, "expected " + std::to_string((long)n->ho.total) + " floats");
    hipSetDevice(n->h->cfg.device_id);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    NET_HIP(n, hipMemcpy(host, src, cnt * 4, hipMemcpyDeviceToHost));
    return GRL_OK;
}
int grl_net_get_params(grl_net *n, float *host, int64_t cnt) { return get_flat(n, n ? n->params : nullptr, host, cnt); }
int grl_net_get_grads(grl_net *n, float *host, int64_t cnt) { return get_flat(n, n ? n->grads : nullptr, host, cnt); }

int grl_net_get_optimizer_state(grl_net *n, float *m_host, float *v_host, int64_t cnt, int64_t *step_out) {
    if (!n || !m_host || !v_host || !step_out) return GRL_E_INVALID;
    int rc = get_flat(n, n->adam_m, m_host, cnt);
    if (rc == GRL_OK) rc = get_flat(n, n->adam_v, v_host, cnt);
    *step_out = n->adam_t;
    return rc;
}

int grl_net_set_optimizer_state(grl_net *n, const float *m_host, const float *v_host, int64_t cnt, int64_t step) {
    if (!n || !m_host || !v_host || step < 0) return GRL_E_INVALID;
    if (cnt != n->ho.total) return nfail(n, GRL_E_SIZE, "grl_net_set_optimizer_state: expected " + std::to_string((long)n->ho.total) + " floats");
    hipSetDevice(n->h->cfg.device_id);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    NET_HIP(n, hipMemcpy(n->adam_m, m_host, cnt * 4, hipMemcpyHostToDevice));
    NET_HIP(n, hipMemcpy(n->adam_v, v_host, cnt * 4, hipMemcpyHostToDevice));
    n->adam_t = (long)step;
    return GRL_OK;
}

// After a synchronisation point: did any GEMM output of the work just finished leave the fp16 range (net_gemm.h: g_gemm_range_flag)?
// The results of that work are then wrong (inf / NaN operands, which ReLU turns into zeros), so the call fails instead of returning them.
// One word per DEVICE (the __device__ pointer g_gemm_range_flag has one instance per device; a process that creates nets on two
// devices gets two words); nets on the same device share it.  Never freed.
constexpr int kMaxDevices = 64;
static int *g_range_flag_dev[kMaxDevices] = {};
static int range_flag_init(grl_net *n) {
    const int dev = n->h->cfg.device_id;
    if (dev < 0 || dev >= kMaxDevices) return nfail(n, GRL_E_INVALID, "device ordinal beyond 63");
    if (g_range_flag_dev[dev]) return GRL_OK;
    NET_HIP(n, hipSetDevice(dev));
    int *p = nullptr;
    NET_HIP(n, hipMalloc((void **)&p, 2 * sizeof(int)));      // [0] the range flag, [1] the fp32 form's largest |output| (float bits)
    NET_HIP(n, hipMemset(p, 0, 2 * sizeof(int)));
    NET_HIP(n, hipMemcpyToSymbol(HIP_SYMBOL(grl::g_gemm_range_flag), &p, sizeof(p)));      // this device's instance of the symbol
    unsigned *pa = reinterpret_cast<unsigned *>(p + 1);
    NET_HIP(n, hipMemcpyToSymbol(HIP_SYMBOL(grl::g_gemm_absmax), &pa, sizeof(pa)));
    (void)hipGetLastError();      // the symbol lookup may probe other ordinals and leave a stale error on this thread
    g_range_flag_dev[dev] = p;
    return GRL_OK;
}
static int *range_flag_ptr(grl_net *n) { return g_range_flag_dev[n->h->cfg.device_id]; }
static unsigned *range_absmax_ptr(grl_net *n) { return reinterpret_cast<unsigned *>(g_range_flag_dev[n->h->cfg.device_id] + 1); }
// flag bits: 1 = some GEMM output of the work since the last check left the fp16 range; 2 = that was already so when the last
// rollout ended (rollout_range_mark_kernel), i.e. the rollout's own actions / values come from invalid operands
static int range_check(grl_net *n, const char *where) {
    int flag = 0;
    NET_HIP(n, hipMemcpy(&flag, range_flag_ptr(n), sizeof(int), hipMemcpyDeviceToHost));
    n->range_bits_last = flag;
    if (!flag) return GRL_OK;
    NET_HIP(n, hipMemset(range_flag_ptr(n), 0, sizeof(int)));
    NET_HIP(n, hipDeviceSynchronize());
    return nfail(n, GRL_E_RANGE, std::string(where) + ": an activation or gradient exceeded 65504, the range of the fp16 matrix-pipe GEMMs "
                                                      "(include/goldsrl_net.h, Arithmetic); the results of this call are not valid");
}

// A pass that left the fp16 range: switch the net to the fp32-MFMA GEMMs and tell the caller to run the work again.  Returns false
// when the fallback is off or the net already computes in fp32.  The net stays on the fp32 form until range_return_k updates in a
// row stayed well inside the fp16 range (range_maybe_return, called where an update ends) or grl_net_set_gemm_f32(net, 0).
static bool range_fall_back(grl_net *n) {
    if (!n->range_fallback_on || n->gemm_f32) return false;
    n->gemm_f32 = 1;
    n->range_fallbacks += 1;
    n->f32_by_fallback = 1;
    n->f32_clean_passes = 0;
    (void)hipMemset(range_absmax_ptr(n), 0, sizeof(unsigned));
    (void)trunk_background(n);      // the background rows in the arithmetic the list GEMMs now use
    return true;
}

// End of an applied update on the fp32 form (stream drained): the largest |value| any gemm_rowk tile handed on since the last
// update -- rollout, gradient step and the background rows of the NEW parameters; with RCCL the word was max-reduced beside the
// range flag, and what came after the collective ran on identical parameters, so every rank reads the same number.  A spike is
// transient: range_return_k clean updates in a row (below a quarter of the fp16 range: a margin of 4 against coming straight back)
// take the net back to the three-product form, whose background rows are rebuilt; should THAT pass leave the range the net is back
// on the fp32 form before anyone has used it.
static int range_maybe_return(grl_net *n) {
    if (!n->gemm_f32 || !n->f32_by_fallback || n->range_return_k <= 0) return GRL_OK;
    unsigned bits = 0;
    NET_HIP(n, hipMemcpy(&bits, range_absmax_ptr(n), sizeof(bits), hipMemcpyDeviceToHost));
    NET_HIP(n, hipMemset(range_absmax_ptr(n), 0, sizeof(unsigned)));
    float amax;
    memcpy(&amax, &bits, sizeof(amax));
    n->absmax_last = amax;
    if (amax < kF16Max / 4.f) n->f32_clean_passes += 1;      // (inf and NaN compare false)
    else n->f32_clean_passes = 0;
    if (n->f32_clean_passes < n->range_return_k) return GRL_OK;
    n->gemm_f32 = 0;
    n->f32_by_fallback = 0;
    n->f32_clean_passes = 0;
    n->range_returns += 1;
    int rc = trunk_background(n);
    if (rc) return rc;
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    int flag = 0;
    NET_HIP(n, hipMemcpy(&flag, range_flag_ptr(n), sizeof(int), hipMemcpyDeviceToHost));
    if (flag) {
        NET_HIP(n, hipMemset(range_flag_ptr(n), 0, sizeof(int)));
        n->range_returns -= 1;
        (void)range_fall_back(n);
        NET_HIP(n, hipStreamSynchronize(n->h->stream));
    }
    return GRL_OK;
}

static int download_heads(grl_net *n, int B, float *mu_host, float *sigma_host, float *vs_host) {
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    if (int rc = range_check(n, "grl_net_predict")) return rc;
    if (mu_host) NET_HIP(n, hipMemcpy(mu_host, n->mu, (size_t)B * n->ho.A * 4, hipMemcpyDeviceToHost));
    if (sigma_host) NET_HIP(n, hipMemcpy(sigma_host, n->sigma, (size_t)B * n->ho.A * 4, hipMemcpyDeviceToHost));
    if (vs_host) NET_HIP(n, hipMemcpy(vs_host, n->vs, (size_t)B * 4, hipMemcpyDeviceToHost));
    return GRL_OK;
}

int grl_net_predict(grl_net *n, float *mu_host, float *sigma_host, float *vs_host) {
    if (!n) return GRL_E_INVALID;
    hipSetDevice(n->h->cfg.device_id);
    grl_handle *h = n->h;
    int rc = forward_all(n, h->sw.lbins, h->sw.abins, h->sw.pos, h->E, n->mu, n->sigma, n->vs);
    if (rc) return rc;
    rc = download_heads(n, h->E * 10, mu_host, sigma_host, vs_host);
    if (rc == GRL_E_RANGE && range_fall_back(n)) {      // once more on the fp32 form
        if ((rc = forward_all(n, h->sw.lbins, h->sw.abins, h->sw.pos, h->E, n->mu, n->sigma, n->vs))) return rc;
        rc = download_heads(n, h->E * 10, mu_host, sigma_host, vs_host);
    }
    return rc;
}

int grl_net_predict_obs(grl_net *n, int32_t n_envs, const uint8_t *lb, const uint8_t *ab, const uint8_t *pos, float *mu_host,
                        float *sigma_host, float *vs_host) {
    if (!n || !lb || !ab || !pos || n_envs <= 0) return GRL_E_INVALID;
    hipSetDevice(n->h->cfg.device_id);
    int rc = ensure_tmp_obs(n, n_envs);
    if (rc) return rc;
    NET_HIP(n, hipMemcpyAsync(n->tmp_lb, lb, (size_t)n_envs * 160, hipMemcpyHostToDevice, n->h->stream));
    NET_HIP(n, hipMemcpyAsync(n->tmp_ab, ab, (size_t)n_envs * 20, hipMemcpyHostToDevice, n->h->stream));
    NET_HIP(n, hipMemcpyAsync(n->tmp_pos, pos, (size_t)n_envs * 20, hipMemcpyHostToDevice, n->h->stream));
    rc = forward_all(n, n->tmp_lb, n->tmp_ab, n->tmp_pos, n_envs, n->mu, n->sigma, n->vs);
    if (rc) return rc;
    rc = download_heads(n, n_envs * 10, mu_host, sigma_host, vs_host);
    if (rc == GRL_E_RANGE && range_fall_back(n)) {
        if ((rc = forward_all(n, n->tmp_lb, n->tmp_ab, n->tmp_pos, n_envs, n->mu, n->sigma, n->vs))) return rc;
        rc = download_heads(n, n_envs * 10, mu_host, sigma_host, vs_host);
    }
    return rc;
}

int grl_net_range_info(grl_net *n, int32_t *gemm_f32, int32_t *fallbacks, int32_t *update_skipped) {
    if (!n) return GRL_E_INVALID;
    if (gemm_f32) *gemm_f32 = n->gemm_f32;
    if (fallbacks) *fallbacks = n->range_fallbacks;
    if (update_skipped) *update_skipped = n->update_skipped_last;
    return GRL_OK;
}

int grl_net_set_gemm_f32(grl_net *n, int32_t on) {
    if (!n) return GRL_E_INVALID;
    hipSetDevice(n->h->cfg.device_id);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    n->gemm_f32 = on ? 1 : 0;
    n->f32_by_fallback = 0;      // the caller's choice stands until the caller (or a range violation) changes it
    n->f32_clean_passes = 0;
    return trunk_background(n);
}

int grl_net_range_return_info(grl_net *n, int32_t *returns_out, int32_t *clean_passes_out, int32_t *needed_out, float *absmax_last_out) {
    if (!n) return GRL_E_INVALID;
    if (returns_out) *returns_out = n->range_returns;
    if (clean_passes_out) *clean_passes_out = n->f32_clean_passes;
    if (needed_out) *needed_out = n->range_return_k;
    if (absmax_last_out) *absmax_last_out = n->absmax_last;
    return GRL_OK;
}

int grl_net_set_range_return(grl_net *n, int32_t clean_passes) {
    if (!n || clean_passes < 0) return GRL_E_INVALID;
    n->range_return_k = clean_passes;
    n->f32_clean_passes = 0;
    return GRL_OK;
}

static int read_activation_impl(grl_net *n, const char *which, float *host, size_t bytes);
int grl_net_read_activation(grl_net *n, const char *which, float *host, size_t bytes) {
    if (!n || !which || !host) return GRL_E_INVALID;
    hipSetDevice(n->h->cfg.device_id);
    use_lane(n, n->last_lane);      // the last chunk's tensors live in the lane that ran it
    int rc = read_activation_impl(n, which, host, bytes);
    use_lane(n, 0);
    return rc;
}
static int read_activation_impl(grl_net *n, const char *which, float *host, size_t bytes) {
    std::string w(which);
    const float *src = nullptr;
    size_t per = 0;
    if (w == "a1") {
        if (n->shared_trunk) return nfail(n, GRL_E_INVALID, "grl_net_read_activation: 'a1' is not materialised in shared-trunk mode (read 'a1sh')");
        src = n->a1; per = 12800;
    }
    else if (w == "a1sh" || w == "sraw") {     // per ENV: (n/10, 20, 20, 32)
        size_t need_e = (size_t)(n->last_n / 10) * 12800 * 4;
        if (bytes != need_e) return nfail(n, GRL_E_SIZE, "grl_net_read_activation: need " + std::to_string(need_e) + " bytes");
        if (n->trunk_skip && n->expand2_gemm)      // list form: the blocks nobody reads were not written; they hold b1 (debug/test access)
            hipLaunchKernelGGL(materialize_sraw_kernel, dim3(n->last_n / 10), dim3(256), 0, n->h->stream, n->tnmask, n->params + ConvOffsets::c1b, n->sraw);
        NET_HIP(n, hipStreamSynchronize(n->h->stream));
        NET_HIP(n, hipMemcpy(host, n->sraw, bytes, hipMemcpyDeviceToHost));
        if (w == "a1sh") {     // relu(sraw): not materialised on the device
            float *hp = static_cast<float *>(host);
            for (size_t i = 0; i < need_e / 4; ++i) hp[i] = hp[i] > 0.f ? hp[i] : 0.f;
        }
        return GRL_OK;
    }
    else if (w == "a2" && n->shared_trunk) {     // not materialised in shared-trunk mode: expand it on demand (debug/test access)
        size_t need_a = (size_t)n->last_n * 5184 * 4;
        if (bytes != need_a) return nfail(n, GRL_E_SIZE, "grl_net_read_activation: need " + std::to_string(need_a) + " bytes");
        if (n->last_n <= 0) return nfail(n, GRL_E_STATE, "grl_net_read_activation: no forward pass yet");
        float *tmp = nullptr;
        NET_HIP(n, hipMalloc((void **)&tmp, need_a));
        if (n->trunk_skip && n->expand2_gemm)      // list form: the pixels no consumer reads were not filled; they hold the background row (debug/test access)
            hipLaunchKernelGGL(trunk_fill_kernel, dim3((n->last_n / 10 * 81 * 16 + 255) / 256), dim3(256), 0, n->h->stream, (const unsigned *)n->tamask, 3, 81,
                               (const float *)(n->tbgz + 64), n->last_n / 10 * 81, n->a2sh, (const float *)nullptr, (float *)nullptr, 0, (const unsigned *)nullptr);
        hipLaunchKernelGGL(materialize_a2_kernel, dim3(n->last_n), dim3(256), 0, n->h->stream, n->a2sh, n->d2s, n->m2s, n->ulist, n->sbase, tmp);
        hipError_t e1 = hipStreamSynchronize(n->h->stream), e2 = hipMemcpy(host, tmp, bytes, hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e1 != hipSuccess || e2 != hipSuccess) return nfail(n, GRL_E_HIP, "grl_net_read_activation: expanding a2 failed");
        return GRL_OK;
    }
    else if (w == "a2") { src = n->a2; per = 5184; }
    else if (w == "a3" && n->shared_trunk) {     // likewise: a3sh with the agent's 5x5 patch replaced by its own values
        size_t need_a = (size_t)n->last_n * 3136 * 4;
        if (bytes != need_a) return nfail(n, GRL_E_SIZE, "grl_net_read_activation: need " + std::to_string(need_a) + " bytes");
        if (n->last_n <= 0) return nfail(n, GRL_E_STATE, "grl_net_read_activation: no forward pass yet");
        float *tmp = nullptr;
        NET_HIP(n, hipMalloc((void **)&tmp, need_a));
        if (n->trunk_skip && n->expand2_gemm)
            hipLaunchKernelGGL(materialize_a3sh_kernel, dim3(n->last_n / 10), dim3(256), 0, n->h->stream, (const unsigned *)n->tumask, (const float *)(n->tbgz3 + 64), n->a3sh);
        hipLaunchKernelGGL(materialize_a3_kernel, dim3(n->last_n), dim3(256), 0, n->h->stream, n->a3sh, n->d3, n->m3, n->org, n->patch_skip ? n->smask : nullptr, tmp);
        hipError_t e1 = hipStreamSynchronize(n->h->stream), e2 = hipMemcpy(host, tmp, bytes, hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e1 != hipSuccess || e2 != hipSuccess) return nfail(n, GRL_E_HIP, "grl_net_read_activation: expanding a3 failed");
        return GRL_OK;
    }
    else if (w == "a3") { src = n->a3; per = 3136; }
    else if (w == "d1") { src = n->d1; per = 512; }
    else if (w == "d2") { src = n->d2; per = 256; }
    else if (w == "p1") { src = n->p1; per = 512; }
    else if (w == "v1") { src = n->v1; per = 512; }
    else if (w == "v2") { src = n->v2; per = 256; }
    else if (w == "d3") { src = n->d3; per = 1600; }                                                   // compact patch differences (net_patch.inc)
    else if (w == "smask") { src = reinterpret_cast<const float *>(n->smask); per = 1; }          // 25-bit patch supports, raw words
    else if (w == "gp1") { src = n->gp1; per = 512; }
    else if (w == "gv2") { src = n->gv2; per = 256; }
    else if (w == "gd2") { src = n->gd2; per = 256; }
    else if (w == "gv1") { src = n->gv1; per = 512; }
    else if (w == "gd1") { src = n->gd1; per = 512; }
    else return nfail(n, GRL_E_INVALID, "grl_net_read_activation: unknown tensor '" + w + "'");
    size_t need = (size_t)n->last_n * per * 4;
    if (bytes != need) return nfail(n, GRL_E_SIZE, "grl_net_read_activation: need " + std::to_string(need) + " bytes");
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    NET_HIP(n, hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost));
    return GRL_OK;
}

int grl_net_profile_enable(grl_net *n, int32_t on) {
    if (!n) return GRL_E_INVALID;
    hipSetDevice(n->h->cfg.device_id);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    if (on && n->prof_ev.empty()) {
        n->prof_ev.resize(131072);   // 65 536 bracketed launches (~13 full updates at 32 768 envs)
        n->prof_tag.assign(65536, 0);
        n->prof_launch_flops.assign(65536, 0.0);
        for (auto &ev : n->prof_ev) NET_HIP(n, hipEventCreate(&ev));
    }
    n->prof_on = on != 0; n->prof_used = 0; n->prof_flops = 0;
    return GRL_OK;
}

int grl_net_profile_read(grl_net *n, int32_t *launches_out, float *total_ms_out, double *flops_out) {
    if (!n || !launches_out || !total_ms_out || !flops_out) return GRL_E_INVALID;
    hipSetDevice(n->h->cfg.device_id);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    float total = 0.f;
    for (size_t i = 0; i + 1 < n->prof_used; i += 2) {
        float ms = 0.f;
        NET_HIP(n, hipEventElapsedTime(&ms, n->prof_ev[i], n->prof_ev[i + 1]));
        total += ms;
    }
    *launches_out = (int32_t)(n->prof_used / 2); *total_ms_out = total; *flops_out = n->prof_flops;
    return GRL_OK;
}

int grl_net_profile_read_tags(grl_net *n, int32_t ntags, int32_t *launches, float *ms, double *flops) {
    if (!n || !launches || !ms || !flops || ntags < PT_COUNT) return GRL_E_INVALID;
    hipSetDevice(n->h->cfg.device_id);
    NET_HIP(n, hipStreamSynchronize(n->h->stream));
    for (int t = 0; t < ntags; ++t) { launches[t] = 0; ms[t] = 0.f; flops[t] = 0.0; }
    for (size_t i = 0; i + 1 < n->prof_used; i += 2) {
        float d = 0.f;
        NET_HIP(n, hipEventElapsedTime(&d, n->prof_ev[i], n->prof_ev[i + 1]));
        const int t = n->prof_tag[i / 2] < PT_COUNT ? n->prof_tag[i / 2] : 0;
        launches[t] += 1; ms[t] += d; flops[t] += n->prof_launch_flops[i / 2];
    }
    return GRL_OK;
}

}  // extern "C"

#include "net_train.inc"
