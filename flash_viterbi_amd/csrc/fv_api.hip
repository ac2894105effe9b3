// fv_api.hip — the C-ABI of include/flashvit.h over the gfx950 kernels.
//
// Everything a decode needs is enqueued on one HIP stream without a single host
// round trip: the task tree depends only on (T, n_split), and every state index a
// later pass consumes (Ans[L-1], Ans[R]) stays in device memory.  One sync at the end.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "flashvit.h"
#include "fv_schedule.h"
#include "fv_kernels.hip.inc"
#include "fv_beam_kernels.hip.inc"

namespace {

using clk = std::chrono::steady_clock;
inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t want)
    {
        if (want <= n) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc(&p, want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    size_t bytes() const { return n * sizeof(T); }
};

}  // namespace

struct fv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // FLASH-BS: the passes of a generation are dealt to up to 1 + BEAM_AUX streams, so that the selects / exact replays
    // of one group (a CU each) run under the step kernels of the others
    static constexpr int BEAM_AUX = 3;
    hipStream_t aux[BEAM_AUX] = { nullptr, nullptr, nullptr };
    hipEvent_t ev_fork = nullptr, ev_join[BEAM_AUX] = { nullptr, nullptr, nullptr };
    bool fork_active = false;     // a forked generation of the current decode has been queued
    hipStream_t lstream = nullptr; // stream the next full-state step launch goes to (nullptr: `stream`)
    bool forked_batches = false;  // the launches of this generation alternate between streams: co-resident workgroups wanted
    int num_cus = 256;       // multiProcessorCount of the device (MI355X: 256)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_top = nullptr, ev_s0 = nullptr, ev_s1 = nullptr;
    std::string detail;

    // model
    int K = 0, M = 0, nrows = 0;
    bool full_ok = false;    // every full-state kernel can take this K (one float32 score row fits LDS: K <= ~40100)
    bool u16_ok = false;     // the packed 16-bit kernel can (one row of 16-bit score codes fits LDS: K <= 65536)
    bool logs_nonpositive = false;
    DevBuf<float> LA32, LB32T;
    DevBuf<unsigned short> LA16, LAQ16;
    DevBuf<uint4> SPdata;    // sparse CSC-Q16 table (fv_kernels.hip.inc, trellis_step_sparse)
    DevBuf<int> SPoff, SPnwb;
    double density = 1.0;    // finite fraction of log A
    float window16 = 0.0f;   // 2 * max |half(L) - L| over the finite table entries
    float windowq = 0.0f, qscale = -1.0f;   // same for the fixed-point table; value = code * qscale
    DevBuf<double> LA64, LB64T, LPi64;

    // workspace
    DevBuf<int> d_ob, d_ans, d_bp, d_gather;
    DevBuf<float> d_rows, d_score, d_ckpt;            // d_ckpt: kept score rows of fv_decode_checkpoint
    DevBuf<unsigned long long> d_counters;
    // beam workspace
    DevBuf<float> d_hval, d_scores, d_slot_val;      // [T][B] members, [T][K] scores, [T][B] exact layout
    DevBuf<int> d_hstate, d_slot_state, d_flags;
    DevBuf<double> LA64R;                            // row-gather copy of the float64 table (built on first beam decode)
    DevBuf<unsigned short> LAQ16R;                   // row-major fixed-point table of beam_step_q16 (same moment; only when every log <= 0)
    DevBuf<unsigned long long> d_qaux;               // [0] lmax bits, [1] dmax bits, then {qscale, window} as floats (q16_params)
    DevBuf<int2> d_tie_list;
    DevBuf<float> d_cut;         // [T][CUT_W] theta, duplicate flag, predicted lower bound of the next cut (topb_select)
    DevBuf<fvb::HNode> d_cand;   // [T][cand_cap] candidate lists of the selects (beam_step epilogue)
    DevBuf<int> d_cand_count;    // [T]
    float opt_sel_margin = 0.5f; // FV_OPT_SEL_MARGIN (in 1/1000): margin of the predicted cut bound in beam spreads
    DevBuf<int> d_dupwin;        // [T]
    DevBuf<int> d_needfull;      // [1] a pass's back-track met a tied cell: rebuild the layouts of the generation (beam_end_backtrack)
    DevBuf<int> d_passL;         // first position of every pass of the generation in flight (beam decodes)
    std::vector<int> h_passL;
    DevBuf<unsigned int> d_tie_count;

    // options
    int opt_kernel = FV_KERNEL_AUTO;
    int opt_max_batch = fvk::MAX_BATCH;
    int opt_profile = 0;
    int vanilla = 0;         // set for the duration of fv_decode_vanilla
    int opt_debug = 0;       // FV_OPT_DEBUG bits: 1 skip refine (timing only), 2 no reverse sweep, 4 alternate unroll, 8 full last step,
                             // 16 launch only / 32 no score-row staging (sparse walk), 64 hipGraph replay, 256 / 512 beam step kernel: float64 / 16-bit
    std::vector<hipEvent_t> prof_events;
    std::vector<int> h_ob;
    std::vector<hipGraphExec_t> graphs;     // experiment (FV_OPT_DEBUG bit 6): destroyed after the decode's sync

    // comm
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;

    fv_stats stats{};
};

namespace {

#define FV_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ctx->detail = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return e_ == hipErrorOutOfMemory ? FV_ERR_NOMEM : FV_ERR_DEVICE;                  \
        }                                                                                     \
    } while (0)

constexpr int FV_NCOUNTERS = 16;   // device statistics words (fv_kernels.hip.inc / fv_beam_kernels.hip.inc say which is which)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

size_t device_bytes(const fv_ctx *c)
{
    return c->LA32.bytes() + c->LA16.bytes() + c->LAQ16.bytes() + c->SPdata.bytes() + c->SPoff.bytes() + c->SPnwb.bytes() + c->LB32T.bytes() + c->LA64.bytes() + c->LB64T.bytes() + c->LPi64.bytes() +
           c->d_ob.bytes() + c->d_ans.bytes() + c->d_bp.bytes() + c->d_gather.bytes() + c->d_rows.bytes() + c->d_ckpt.bytes() +
           c->d_score.bytes() + c->d_counters.bytes() + c->d_hval.bytes() + c->d_scores.bytes() +
           c->d_hstate.bytes() + c->d_flags.bytes() + c->d_slot_val.bytes() + c->d_slot_state.bytes() +
           c->LA64R.bytes() + c->LAQ16R.bytes() + c->d_qaux.bytes() + c->d_tie_list.bytes() + c->d_tie_count.bytes() + c->d_cut.bytes() + c->d_dupwin.bytes() + c->d_cand.bytes() + c->d_cand_count.bytes() + c->d_passL.bytes() + c->d_needfull.bytes();
}

// log() of a strided block of floats on several host threads (same libm call per entry as the reference).
template <typename F>
void parallel_rows(int rows, F &&fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<unsigned>(hw ? hw : 4, 16);
    if (rows < 256) nt = 1;
    if (nt <= 1) { fn(0, rows); return; }
    std::vector<std::thread> th;
    int per = (rows + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        int a = t * per, b = std::min(rows, a + per);
        if (a >= b) break;
        th.emplace_back([=, &fn] { fn(a, b); });
    }
    for (auto &x : th) x.join();
}

int pick_kernel(const fv_ctx *ctx)
{
    if (ctx->opt_kernel == FV_KERNEL_F64_STREAM) return FV_KERNEL_F64_STREAM;
    // the filter kernels' error bracket needs every log <= 0 (no cancellation between score and log A)
    if (!ctx->logs_nonpositive) return FV_KERNEL_F64_STREAM;
    // Measured at K=3965 (us per step of the whole-sequence pass): q16 9.5, f32 11.1, f16 12.8, f64 20.7.
    // binary16's 2^-11 relative spacing makes its window ~0.008 wide (~430 extra candidates and ~7 lane
    // rescans per step); 16-bit fixed point has a window of ~3e-4 (~21 and 0.4) at the same 2 B/cell.
    if (ctx->opt_kernel == FV_KERNEL_F16_REFINE || ctx->opt_kernel == FV_KERNEL_F32_REFINE ||
        ctx->opt_kernel == FV_KERNEL_Q16_REFINE || ctx->opt_kernel == FV_KERNEL_U16_REFINE)
        return ctx->opt_kernel;
    if (ctx->opt_kernel == FV_KERNEL_SPARSE_Q16) return ctx->SPdata.p ? FV_KERNEL_SPARSE_Q16 : FV_KERNEL_U16_REFINE;
    // AUTO: the sparse walk visits only finite entries; it wins clearly below ~1/3 density
    return (ctx->SPdata.p && ctx->density <= 0.35) ? FV_KERNEL_SPARSE_Q16 : FV_KERNEL_U16_REFINE;
}

// Kernel variants: chunks of U 16-byte loads per lane, double-buffered in registers.
// "Upfront" (one register buffer holding the wave's whole share of the tile, requested before the
// score row is staged) measured SLOWER for the f32 table at K=3965 (16.3 vs 12.9 us/step): with every
// workgroup's whole tile in flight the L2 lines kept from the previous (opposite-direction) sweep are
// evicted before they are re-read.  Kept behind FV_OPT_DEBUG bit 2 for experiments.
constexpr int U_UP = 16, U_DB32 = 4, U_DB64 = 2, U_DB16 = 2;

template <typename TA, int NB, int U, bool DB>
int launch_variant(fv_ctx *ctx, const fvk::StepArgs<NB> &a, size_t lds)
{
    hipLaunchKernelGGL((fvk::trellis_step<TA, NB, U, DB>), dim3(a.tiles_per_xcd * 8), dim3(fvk::BLOCK), lds, ctx->lstream ? ctx->lstream : ctx->stream, a);
    FV_HIP(hipGetLastError());
    return 0;
}

template <typename TA, int NB>
int launch_step_nb(fv_ctx *ctx, const fvk::TaskSlot *slots, int nb, int reverse)
{
    fvk::StepArgs<NB> a;
    if constexpr (std::is_same<TA, double>::value) { a.LA = ctx->LA64.p; a.window = 0.0f; }
    else if constexpr (std::is_same<TA, float>::value) { a.LA = ctx->LA32.p; a.window = 0.0f; }
    else if constexpr (std::is_same<TA, fvk::q16_t>::value) { a.LA = ctx->LAQ16.p; a.window = ctx->windowq; }
    else { a.LA = ctx->LA16.p; a.window = ctx->window16; }
    a.qscale = ctx->qscale;
    a.vanilla = ctx->vanilla;
    a.LA64 = ctx->LA64.p;
    a.counters = ctx->d_counters.p;
    a.K = ctx->K;
    a.reverse = (ctx->opt_debug & 2) ? 0 : reverse;
    a.debug = ctx->opt_debug;
    a.nrows = ctx->nrows;
    a.ntiles = (ctx->K + fvk::TILE_W - 1) / fvk::TILE_W;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    a.nb = nb;
    for (int t = 0; t < NB; ++t) a.t[t] = slots[t < nb ? t : 0];
    const size_t lds = fvk::step_lds_bytes<NB>(ctx->nrows);
    constexpr int RBR = 4 * fvk::Tab<TA>::R;
    const int nj_max = (ctx->nrows / RBR + fvk::NWAVES - 1) / fvk::NWAVES;
    if constexpr (std::is_same<TA, double>::value) {
        return launch_variant<TA, NB, U_DB64, true>(ctx, a, lds);
    } else if constexpr (std::is_same<TA, float>::value) {
        if constexpr (NB <= 2) {
            if (nj_max <= U_UP && (ctx->opt_debug & 4)) return launch_variant<TA, NB, U_UP, false>(ctx, a, lds);
        }
        return launch_variant<TA, NB, U_DB32, true>(ctx, a, lds);
    } else {
        // 16-bit tables: an XCD's slab (3.9 MB at K=3965) nearly fits its L2, so requesting the whole
        // tile before staging the score row wins (9.5 vs 9.8 us/step); FV_OPT_DEBUG bit 2 turns it off.
        // That variant holds the tile in 86 VGPRs: one workgroup per CU.  With more tiles than CUs the
        // double-buffered one (46 VGPRs, two workgroups per CU) keeps the grid in one round
        // (K=5632: 18.3 vs 25.3 us/step, K=8192: 25.8 vs 39.8).
        if constexpr (NB <= 2) {
            if (nj_max <= U_UP && a.ntiles <= ctx->num_cus && !(ctx->opt_debug & 4)) return launch_variant<TA, NB, U_UP, false>(ctx, a, lds);
        }
        // (8-wave workgroups for the batched launches, as the packed 16-bit kernel uses, were measured: 2.14 vs 2.10 ms of
        // right-hand passes at cfg2, 65.6 vs 62.7 ms at cfg3 — the f32 sweep needs its four waves per SIMD)
        return launch_variant<TA, NB, U_DB16, true>(ctx, a, lds);
    }
}

template <typename TA>
int launch_step(fv_ctx *ctx, const fvk::TaskSlot *slots, int nb, int reverse)
{
    if (nb <= 1) return launch_step_nb<TA, 1>(ctx, slots, nb, reverse);
    if (nb <= 2) return launch_step_nb<TA, 2>(ctx, slots, nb, reverse);
    if (nb <= 4) return launch_step_nb<TA, 4>(ctx, slots, nb, reverse);
    return launch_step_nb<TA, 8>(ctx, slots, nb, reverse);
}

template <int NB>
int launch_sparse_nb(fv_ctx *ctx, const fvk::TaskSlot *slots, int nb)
{
    fvk::SparseArgs<NB> a;
    a.data = ctx->SPdata.p; a.tile_off = ctx->SPoff.p; a.tile_nwb = ctx->SPnwb.p;
    a.LA64 = ctx->LA64.p; a.counters = ctx->d_counters.p;
    a.K = ctx->K; a.nrows = ctx->nrows;
    a.ntiles = (ctx->K + fvk::TILE_W - 1) / fvk::TILE_W;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    a.nb = nb; a.debug = ctx->opt_debug;
    a.window = ctx->windowq; a.qscale = ctx->qscale;
    for (int t = 0; t < NB; ++t) a.t[t] = slots[t < nb ? t : 0];
    hipLaunchKernelGGL((fvk::trellis_step_sparse<NB>), dim3(a.tiles_per_xcd * 8), dim3(fvk::SP_BLOCK),
                       fvk::sparse_lds_bytes<NB>(ctx->nrows), ctx->lstream ? ctx->lstream : ctx->stream, a);
    FV_HIP(hipGetLastError());
    return 0;
}

int launch_sparse(fv_ctx *ctx, const fvk::TaskSlot *slots, int nb)
{
    if (nb <= 1) return launch_sparse_nb<1>(ctx, slots, nb);
    if (nb <= 2) return launch_sparse_nb<2>(ctx, slots, nb);
    if (nb <= 4) return launch_sparse_nb<4>(ctx, slots, nb);
    return launch_sparse_nb<8>(ctx, slots, nb);
}

template <int NB, int U, bool DB, int NWV>
int launch_u16_variant(fv_ctx *ctx, const fvk::StepArgs<NB> &a)
{
    const size_t lds = fvk::u16_lds_bytes<NB, NWV>(ctx->nrows);
    hipLaunchKernelGGL((fvk::trellis_step_u16<NB, U, DB, NWV>), dim3(a.tiles_per_xcd * 8), dim3(NWV * 64), lds, ctx->lstream ? ctx->lstream : ctx->stream, a);
    FV_HIP(hipGetLastError());
    return 0;
}

template <int NB>
int launch_u16_nb(fv_ctx *ctx, const fvk::TaskSlot *slots, int nb, int reverse)
{
    fvk::StepArgs<NB> a;
    a.LA = ctx->LAQ16.p; a.window = ctx->windowq; a.qscale = ctx->qscale; a.vanilla = 0;
    a.LA64 = ctx->LA64.p; a.counters = ctx->d_counters.p;
    a.K = ctx->K; a.reverse = (ctx->opt_debug & 2) ? 0 : reverse; a.debug = ctx->opt_debug;
    a.nrows = ctx->nrows;
    a.ntiles = (ctx->K + fvk::TILE_W - 1) / fvk::TILE_W;
    a.tiles_per_xcd = (a.ntiles + 7) / 8;
    a.nb = nb;
    for (int t = 0; t < NB; ++t) a.t[t] = slots[t < nb ? t : 0];
    const int nq = ctx->nrows / 32;
    // 8 waves per workgroup (FV_OPT_DEBUG bit 13: 16): everything outside the sweep — quantisation, reductions, refine —
    // is executed by every wave, so fewer, longer waves spend fewer issue slots on it
    if (ctx->opt_debug & 8192) {
        if (a.ntiles <= ctx->num_cus && (nq + 15) / 16 <= 8 && !(ctx->opt_debug & 4)) return launch_u16_variant<NB, 8, false, 16>(ctx, a);
        return launch_u16_variant<NB, U_DB16, true, 16>(ctx, a);
    }
    // (forked batches: the double-buffered form, 81 VGPRs at NB = 4 — three 8-wave workgroups of three streams share a CU;
    // the whole-tile form's 137 would leave room for one)
    if (a.ntiles <= ctx->num_cus && (nq + 7) / 8 <= 16 && !(ctx->opt_debug & 4) && !ctx->forked_batches) return launch_u16_variant<NB, 16, false, 8>(ctx, a);
    return launch_u16_variant<NB, U_DB16, true, 8>(ctx, a);
}

int launch_u16(fv_ctx *ctx, const fvk::TaskSlot *slots, int nb, int reverse)
{
    if (nb <= 1) return launch_u16_nb<1>(ctx, slots, nb, reverse);
    if (nb <= 2) return launch_u16_nb<2>(ctx, slots, nb, reverse);
    if (nb <= 4) return launch_u16_nb<4>(ctx, slots, nb, reverse);
    return launch_u16_nb<8>(ctx, slots, nb, reverse);
}

int launch_step_kernel(fv_ctx *ctx, int kernel, const fvk::TaskSlot *slots, int nb, int reverse)
{
    switch (kernel) {
    case FV_KERNEL_U16_REFINE:
        // Both filters read the same 16-bit table and give the same bits, so the choice is per launch: the packed
        // 16-bit filter for single-task launches (the whole-sequence pass: 9.1 vs 9.3 us per step at K=3965/T=256,
        // 10.4 vs 14.2 at T=4096 where its window is the narrower one) and for models whose float32 rows do not fit
        // LDS; the f32 filter for batched launches, where the 16-bit kernel's per-task prologue (row maximum,
        // quantisation) costs what its cheaper sweep saves.  FV_OPT_DEBUG bit 14: packed 16-bit for every launch.
        if (nb <= 1 || !ctx->full_ok || (ctx->opt_debug & 16384) || ctx->forked_batches) return launch_u16(ctx, slots, nb, reverse);
        return launch_step<fvk::q16_t>(ctx, slots, nb, reverse);
    case FV_KERNEL_SPARSE_Q16: return launch_sparse(ctx, slots, nb);
    case FV_KERNEL_F64_STREAM: return launch_step<double>(ctx, slots, nb, reverse);
    case FV_KERNEL_F32_REFINE: return launch_step<float>(ctx, slots, nb, reverse);
    case FV_KERNEL_Q16_REFINE: return launch_step<fvk::q16_t>(ctx, slots, nb, reverse);
    default: return launch_step<fvk::half_t>(ctx, slots, nb, reverse);
    }
}

template <typename K>
int set_big_lds(fv_ctx *ctx, K kernel)
{
    FV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return 0;
}

template <typename TA, int NB>
int allow_big_lds(fv_ctx *ctx)
{
    int rc = 0;
    if constexpr (std::is_same<TA, double>::value) {
        rc = set_big_lds(ctx, &fvk::trellis_step<TA, NB, U_DB64, true>);
    } else {
        constexpr int U = std::is_same<TA, float>::value ? U_DB32 : U_DB16;
        rc = set_big_lds(ctx, &fvk::trellis_step<TA, NB, U, true>);
        if constexpr (NB <= 2) { if (!rc) rc = set_big_lds(ctx, &fvk::trellis_step<TA, NB, U_UP, false>); }
    }
    return rc;
}

// largest batch whose score rows fit LDS next to the reduction scratch
int max_batch_for(int nrows, bool u16_only)
{
    if (u16_only) {             // models beyond the float32 kernels' limit: rows of 16-bit codes
        if (fvk::u16_lds_bytes<8, 8>(nrows) <= 160 * 1024) return 8;
        if (fvk::u16_lds_bytes<4, 8>(nrows) <= 160 * 1024) return 4;
        if (fvk::u16_lds_bytes<2, 8>(nrows) <= 160 * 1024) return 2;
        return 1;
    }
    int nb = fvk::MAX_BATCH;
    while (nb > 1) {
        size_t need = nb == 8 ? fvk::step_lds_bytes<8>(nrows) : nb == 4 ? fvk::step_lds_bytes<4>(nrows)
                                                                        : fvk::step_lds_bytes<2>(nrows);
        if (need <= 160 * 1024) break;
        nb >>= 1;
    }
    return nb;
}

int ensure_workspace(fv_ctx *ctx, int T, size_t rows_needed)
{
    FV_HIP(ctx->d_ob.ensure(T));
    FV_HIP(ctx->d_ans.ensure(T));
    FV_HIP(ctx->d_bp.ensure((size_t)T * ctx->K));
    {
        const size_t want = rows_needed * 2 * (size_t)ctx->nrows;
        if (want > ctx->d_rows.n) {
            FV_HIP(ctx->d_rows.ensure(want));
            FV_HIP(hipMemsetAsync(ctx->d_rows.p, 0, want * sizeof(float), ctx->stream));   // row pads stay zero
        }
    }
    FV_HIP(ctx->d_score.ensure(4));
    FV_HIP(ctx->d_counters.ensure(FV_NCOUNTERS));
    if (ctx->comm) FV_HIP(ctx->d_gather.ensure((size_t)T * ctx->nranks));
    return 0;
}

struct ProfRange { size_t first, count; };

int prof_event(fv_ctx *ctx, size_t idx, hipEvent_t *out)
{
    while (ctx->prof_events.size() <= idx) {
        hipEvent_t e;
        FV_HIP(hipEventCreate(&e));
        ctx->prof_events.push_back(e);
    }
    *out = ctx->prof_events[idx];
    return 0;
}

// Runs every pass of one generation in lock-step: at lock-step s each still-active pass advances
// from time L+s-1 to L+s.  Passes are sorted longest first so the active set is a prefix.
int run_generation_full(fv_ctx *ctx, std::vector<fv::Pass> &passes, int kernel, size_t &nprof)
{
    const int K = ctx->K;
    const int np = (int)passes.size();
    if (np == 0) return 0;
    std::stable_sort(passes.begin(), passes.end(),
                     [](const fv::Pass &a, const fv::Pass &b) { return a.R - a.L > b.R - b.L; });
    // init rows
    for (int base = 0; base < np; base += fvk::PASS_CHUNK) {
        fvk::PassChunk ch;
        ch.n = std::min(fvk::PASS_CHUNK, np - base);
        for (int q = 0; q < ch.n; ++q) {
            const fv::Pass &p = passes[base + q];
            ch.p[q] = fvk::PassDesc{ p.L, p.R, p.from_pi ? 1 : 0, p.whole ? 1 : 0, (long long)(base + q) * 2 * ctx->nrows };
        }
        hipLaunchKernelGGL(fvk::init_rows, dim3((K + 255) / 256, ch.n), dim3(256), 0, ctx->stream, ch,
                           ctx->LA64.p, ctx->nrows, ctx->LB64T.p, ctx->LPi64.p, ctx->d_ob.p, ctx->d_ans.p,
                           ctx->d_rows.p, K);
        FV_HIP(hipGetLastError());
    }
    const int maxlen = passes[0].R - passes[0].L;
    int cap = std::max(1, std::min(ctx->opt_max_batch, max_batch_for(ctx->nrows, !ctx->full_ok)));
    const bool whole_gen = passes[0].whole;       // generation 0: bracket its step launches for the stats
    // The batches of a lock-step are independent, and a step launch is latency-bound at both ends (staging the score
    // rows; reductions and refine): the right-hand generations of the packed 16-bit kernel therefore run as batches of
    // FORK_CAP tasks dealt to FORK_STREAMS streams, small enough (36 KB of LDS, 81 VGPRs, 8 waves) for three workgroups
    // of different launches to share a CU — one launch's head and tail run under the sweeps of the others.  cfg2
    // right-hand passes 2.10 -> 1.85 ms, cfg3 62.7 -> 44.7 ms; the f32 filter in two co-resident 8-wave workgroups gave
    // 1.96 / 53.0, batches of two tasks and four streams were slower.  The sparse walk gains the same way (cfg3 right-hand
    // 33.4 -> 23.1 ms; at cfg2's 31-step passes nothing, so short generations stay on one stream and the host keeps
    // running ahead).  FV_OPT_DEBUG bit 18: off.
    constexpr int FORK_STREAMS = 3, FORK_CAP = 4;
    const bool two = ((kernel == FV_KERNEL_U16_REFINE && ctx->u16_ok) || (kernel == FV_KERNEL_SPARSE_Q16 && maxlen >= 64)) &&
                     !(ctx->opt_debug & 262144) && !whole_gen && np > FORK_CAP &&
                     !ctx->opt_profile && !(ctx->opt_debug & 64);
    const int nbatches = (np + FORK_CAP - 1) / FORK_CAP;
    const int nstreams = two ? std::min(FORK_STREAMS, nbatches) : 1;      // (batches of three tasks were slower: 2.16 / 54.9 ms)
    if (two) {
        cap = std::min(cap, FORK_CAP);
        // no packets may wait on the other queues while a serial generation runs (decode_beam_impl has the measurement)
        if (!ctx->fork_active) FV_HIP(hipStreamSynchronize(ctx->stream));
        ctx->fork_active = true;
        FV_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int q = 1; q < nstreams; ++q) FV_HIP(hipStreamWaitEvent(ctx->aux[q - 1], ctx->ev_fork, 0));
    }
    ctx->forked_batches = two;
    struct Unfork { fv_ctx *c; ~Unfork() { c->forked_batches = false; c->lstream = nullptr; } } unfork{ ctx };
    auto join = [&]() -> int {
        for (int q = 1; q < nstreams; ++q) {
            FV_HIP(hipEventRecord(ctx->ev_join[q - 1], ctx->aux[q - 1]));
            FV_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[q - 1], 0));
        }
        return 0;
    };
    const bool col_last = !(ctx->opt_debug & 8);  // FV_OPT_DEBUG bit 3: run every last step as a full step
    // FV_OPT_DEBUG bit 6 (experiment): capture this generation's step launches into a hipGraph and replay it
    const bool use_graph = (ctx->opt_debug & 64) && !ctx->opt_profile;
    if (use_graph) FV_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    if (whole_gen && !use_graph) FV_HIP(hipEventRecord(ctx->ev_s0, ctx->stream));
    // passes are sorted longest first: at lock-step s the passes with len >= s are a prefix; those with
    // len == s are finishing and (unless they are the whole-sequence pass) only need one column
    int active = np;
    for (int s = 1; s <= maxlen; ++s) {
        while (active > 0 && passes[active - 1].R - passes[active - 1].L < s) --active;
        int full = active;                           // passes [0, full) take a full step
        if (col_last) while (full > 0 && passes[full - 1].R - passes[full - 1].L == s && !passes[full - 1].whole) --full;
        auto row = [&](int q, int parity) { return ctx->d_rows.p + (size_t)q * 2 * ctx->nrows + (size_t)parity * ctx->nrows; };
        for (int base = 0; base < full; base += cap) {
            const int nb = std::min(cap, full - base);
            fvk::TaskSlot slots[fvk::MAX_BATCH];
            for (int q = 0; q < nb; ++q) {
                const fv::Pass &p = passes[base + q];
                slots[q].t1_in = row(base + q, (s - 1) & 1);
                slots[q].t1_out = row(base + q, s & 1);
                slots[q].tmp_row = ctx->LB32T.p + (size_t)ctx->h_ob[p.L + s] * K;
                slots[q].tmp64_row = ctx->LB64T.p + (size_t)ctx->h_ob[p.L + s] * K;
                slots[q].bp_out = ctx->d_bp.p + (size_t)(p.L + s) * K;
            }
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (ctx->opt_profile) {
                int rc = prof_event(ctx, nprof, &e0); if (rc) return rc;
                rc = prof_event(ctx, nprof + 1, &e1); if (rc) return rc;
                nprof += 2;
                FV_HIP(hipEventRecord(e0, ctx->stream));
            }
            const int other = two ? (base / cap) % nstreams : 0;         // batch b keeps its stream for the whole generation
            ctx->lstream = other ? ctx->aux[other - 1] : ctx->stream;
            int rc = launch_step_kernel(ctx, kernel, slots, nb, s & 1);
            ctx->lstream = nullptr;
            if (rc) return rc;
            if (ctx->opt_profile) FV_HIP(hipEventRecord(e1, ctx->stream));
            ctx->stats.step_launches += 1;
            ctx->stats.task_steps += nb;
        }
        if (two && full < active) {      // the single-column last steps read rows every stream has written; the finished
            int rc = join();              // passes are the tail of the list, so the batches that go on keep their streams
            if (rc) return rc;
        }
        for (int base = full; base < active; base += fvk::COL_CHUNK) {
            fvk::ColArgs c;
            c.LA64 = ctx->LA64.p; c.ans = ctx->d_ans.p; c.K = K; c.nrows = ctx->nrows;
            c.n = std::min(fvk::COL_CHUNK, active - base);
            for (int q = 0; q < c.n; ++q) {
                const fv::Pass &p = passes[base + q];
                c.p[q] = fvk::ColJob{ row(base + q, (s - 1) & 1), ctx->LB32T.p + (size_t)ctx->h_ob[p.R] * K,
                                      ctx->d_bp.p + (size_t)p.R * K, p.R };
            }
            hipLaunchKernelGGL(fvk::last_column, dim3(c.n), dim3(256), 0, ctx->stream, c);
            FV_HIP(hipGetLastError());
            ctx->stats.column_steps += c.n;
        }
    }
    if (use_graph) {
        hipGraph_t g = nullptr;
        FV_HIP(hipStreamEndCapture(ctx->stream, &g));
        hipGraphExec_t ge = nullptr;
        FV_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        if (whole_gen) FV_HIP(hipEventRecord(ctx->ev_s0, ctx->stream));
        FV_HIP(hipGraphLaunch(ge, ctx->stream));
        ctx->graphs.push_back(ge);
        (void)hipGraphDestroy(g);
    }
    if (whole_gen) FV_HIP(hipEventRecord(ctx->ev_s1, ctx->stream));
    if (two) { int rc = join(); if (rc) return rc; }
    // end states + chains
    for (int q = 0; q < np; ++q) {
        if (!passes[q].whole) continue;
        const int len = passes[q].R - passes[q].L;
        const float *last = ctx->d_rows.p + (size_t)q * 2 * ctx->nrows + (size_t)(len & 1) * ctx->nrows;
        hipLaunchKernelGGL(fvk::final_argmax, dim3(1), dim3(1024), 0, ctx->stream, last, K,
                           ctx->d_ans.p + passes[q].R, ctx->d_score.p);
        FV_HIP(hipGetLastError());
    }
    for (int base = 0; base < np; base += fvk::PASS_CHUNK) {
        fvk::PassChunk ch;
        ch.n = std::min(fvk::PASS_CHUNK, np - base);
        for (int q = 0; q < ch.n; ++q) {
            const fv::Pass &p = passes[base + q];
            ch.p[q] = fvk::PassDesc{ p.L, p.R, p.from_pi ? 1 : 0, p.whole ? 1 : 0, 0 };
        }
        hipLaunchKernelGGL(fvk::backtrack, dim3(ch.n), dim3(64), 0, ctx->stream, ch, ctx->d_bp.p, K, ctx->d_ans.p);
        FV_HIP(hipGetLastError());
    }
    return 0;
}

// Merge of the per-rank answer arrays after the all-gather: position j is taken from the rank that
// owns the top-level segment containing it (segment end points are fixed by the whole-sequence
// pass, identical on every rank).
void merge_gathered(const fv::Plan &plan, const std::vector<int> &gathered, int T, int nranks, int *path)
{
    for (int j = 0; j < T; ++j) path[j] = gathered[j];   // rank 0's copy: whole-pass values
    for (size_t s = 0; s < plan.seg_L.size(); ++s) {
        const int r = plan.seg_owner[s] % nranks;
        for (int j = plan.seg_L[s]; j < plan.seg_R[s]; ++j) path[j] = gathered[(size_t)r * T + j];
    }
}

int finish_decode(fv_ctx *ctx, const fv::Plan &plan, int T, int *path_out, float *score_out, clk::time_point t0,
                  size_t nprof, bool beam)
{
    std::vector<int> host;
    if (ctx->comm && !plan.seg_L.empty()) {
        ncclResult_t nr = ncclAllGather(ctx->d_ans.p, ctx->d_gather.p, (size_t)T, ncclInt32, ctx->comm, ctx->stream);
        if (nr != ncclSuccess) { ctx->detail = std::string("ncclAllGather: ") + ncclGetErrorString(nr); return FV_ERR_COMM; }
        host.resize((size_t)T * ctx->nranks);
        FV_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
        FV_HIP(hipMemcpyAsync(host.data(), ctx->d_gather.p, host.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    } else {
        FV_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
        FV_HIP(hipMemcpyAsync(path_out, ctx->d_ans.p, (size_t)T * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    float score = 0.f;
    unsigned long long counters[FV_NCOUNTERS] = {};
    FV_HIP(hipMemcpyAsync(&score, ctx->d_score.p, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(hipMemcpyAsync(counters, ctx->d_counters.p, sizeof counters, hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(hipStreamSynchronize(ctx->stream));
    for (hipGraphExec_t ge : ctx->graphs) (void)hipGraphExecDestroy(ge);
    ctx->graphs.clear();
    if (!host.empty()) merge_gathered(plan, host, T, ctx->nranks, path_out);
    if (score_out) *score_out = score;

    fv_stats &st = ctx->stats;
    st.decode_ms = ms_since(t0);
    float ms = 0.f;
    FV_HIP(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop)); st.gpu_ms = ms;
    FV_HIP(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_top)); st.top_pass_ms = ms;
    FV_HIP(hipEventElapsedTime(&ms, ctx->ev_s0, ctx->ev_s1)); st.top_steps_ms = ms;
    st.step_kernel_ms = 0;
    for (size_t i = 0; i + 1 < nprof; i += 2) {
        FV_HIP(hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]));
        st.step_kernel_ms += ms;
    }
    st.refine_near = (long long)counters[0];
    st.refine_rescan = (long long)counters[1];
    st.beam_exact_sets = (long long)counters[2];
    st.beam_dup_cols = (long long)counters[3];
    st.beam_dup_steps = (long long)counters[4];
    st.beam_ties = (long long)counters[6];
    st.beam_cand_selects = (long long)counters[7];
    st.refine_saturated = (long long)counters[8];
    if (counters[5]) { ctx->detail = "heap replay: producer/consumer hand-shake timed out"; return FV_ERR_DEVICE; }
    st.device_bytes = (long long)device_bytes(ctx);
    st.ranks = ctx->nranks;
    bool neg = false;
    for (int j = 0; j < T; ++j) neg |= path_out[j] < 0;
    if (neg) return beam ? FV_WARN_BEAM_MISS : FV_ERR_NO_PRED;
    return FV_OK;
}

// A decode that fails after its first enqueue must not leave kernels running on ctx->stream: the next call
// may grow (hipFree + hipMalloc) a workspace buffer they still use.  Every entry point funnels its error
// returns through here: end a capture left open, wait for the stream, drop the captured graphs.
int drained(fv_ctx *ctx, int rc)
{
    if (rc >= 0) return rc;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(ctx->stream, &g);
        if (g) (void)hipGraphDestroy(g);
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t a : ctx->aux) if (a) (void)hipStreamSynchronize(a);
    for (hipGraphExec_t ge : ctx->graphs) (void)hipGraphExecDestroy(ge);
    ctx->graphs.clear();
    (void)hipGetLastError();          // the failure is reported through rc / detail, not left sticky
    return rc;
}

}  // namespace

// ------------------------------------------------------------------ C ABI

extern "C" int fv_create(fv_ctx **out, int device)
{
    if (!out) return FV_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FV_ERR_DEVICE;
    fv_ctx *ctx = new (std::nothrow) fv_ctx();
    if (!ctx) return FV_ERR_NOMEM;
    ctx->device = device;
    auto fail = [&](int rc) { fv_destroy(ctx); return rc; };
    if (hipSetDevice(device) != hipSuccess) return fail(FV_ERR_DEVICE);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return fail(FV_ERR_DEVICE);
    for (int q = 0; q < fv_ctx::BEAM_AUX; ++q)
        if (hipStreamCreateWithFlags(&ctx->aux[q], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_join[q], hipEventDisableTiming) != hipSuccess) return fail(FV_ERR_DEVICE);
    if (hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(FV_ERR_DEVICE);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0) ctx->num_cus = cus;
    }
    if (hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess ||
        hipEventCreate(&ctx->ev_top) != hipSuccess || hipEventCreate(&ctx->ev_s0) != hipSuccess ||
        hipEventCreate(&ctx->ev_s1) != hipSuccess)
        return fail(FV_ERR_DEVICE);
    int rc = 0;
    if ((rc = allow_big_lds<float, 1>(ctx)) || (rc = allow_big_lds<float, 2>(ctx)) || (rc = allow_big_lds<float, 4>(ctx)) ||
        (rc = allow_big_lds<float, 8>(ctx)) || (rc = allow_big_lds<double, 1>(ctx)) || (rc = allow_big_lds<double, 2>(ctx)) ||
        (rc = allow_big_lds<double, 4>(ctx)) || (rc = allow_big_lds<double, 8>(ctx)) ||
        (rc = allow_big_lds<fvk::half_t, 1>(ctx)) || (rc = allow_big_lds<fvk::half_t, 2>(ctx)) ||
        (rc = allow_big_lds<fvk::half_t, 4>(ctx)) || (rc = allow_big_lds<fvk::half_t, 8>(ctx)) ||
        (rc = allow_big_lds<fvk::q16_t, 1>(ctx)) || (rc = allow_big_lds<fvk::q16_t, 2>(ctx)) ||
        (rc = allow_big_lds<fvk::q16_t, 4>(ctx)) || (rc = allow_big_lds<fvk::q16_t, 8>(ctx)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_sparse<1>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_sparse<2>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_sparse<4>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_sparse<8>)))
        return fail(rc);
    if ((rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, U_DB16, true, 8>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, U_DB16, true, 8>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, U_DB16, true, 8>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, U_DB16, true, 8>)))
        return fail(rc);
    if ((rc = fvb::allow_big_lds(ctx->detail))) return fail(rc);
    *out = ctx;
    return FV_OK;
}

extern "C" void fv_destroy(fv_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) ncclCommDestroy(ctx->comm);
    ctx->LA32.release(); ctx->LA16.release(); ctx->LAQ16.release(); ctx->SPdata.release(); ctx->SPoff.release(); ctx->SPnwb.release(); ctx->LB32T.release(); ctx->LA64.release(); ctx->LB64T.release(); ctx->LPi64.release();
    ctx->d_ob.release(); ctx->d_ans.release(); ctx->d_bp.release(); ctx->d_gather.release(); ctx->d_rows.release(); ctx->d_ckpt.release();
    ctx->d_score.release(); ctx->d_counters.release(); ctx->d_hval.release(); ctx->d_scores.release();
    ctx->d_hstate.release(); ctx->d_flags.release(); ctx->d_slot_val.release(); ctx->d_slot_state.release();
    ctx->LA64R.release(); ctx->LAQ16R.release(); ctx->d_qaux.release(); ctx->d_tie_list.release(); ctx->d_tie_count.release(); ctx->d_cut.release(); ctx->d_dupwin.release(); ctx->d_cand.release(); ctx->d_cand_count.release(); ctx->d_passL.release(); ctx->d_needfull.release();
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->ev_top) (void)hipEventDestroy(ctx->ev_top);
    if (ctx->ev_s0) (void)hipEventDestroy(ctx->ev_s0);
    if (ctx->ev_s1) (void)hipEventDestroy(ctx->ev_s1);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (int q = 0; q < fv_ctx::BEAM_AUX; ++q) {
        if (ctx->ev_join[q]) (void)hipEventDestroy(ctx->ev_join[q]);
        if (ctx->aux[q]) (void)hipStreamDestroy(ctx->aux[q]);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int fv_set_model(fv_ctx *ctx, const float *A, const float *B, const float *Pi, int K, int M)
{
    if (!ctx || !A || !B || !Pi || K < 1 || M < 1) return FV_ERR_ARG;
    auto t0 = clk::now();
    FV_HIP(hipSetDevice(ctx->device));
    const int nrows = round_up(K, fvk::ROW_ALIGN);
    const int ntiles = (K + fvk::TILE_W - 1) / fvk::TILE_W;
    // the full-state step kernels keep one score row in LDS: beyond that K only the beam path is available
    // (it needs the float64 table alone, which also keeps the host footprint at 8 B per entry)
    const bool full_ok = fvk::step_lds_bytes<1>(nrows) <= 160 * 1024;
    const bool u16_ok = fvk::u16_lds_bytes<1, 8>(nrows) <= 160 * 1024 && K <= 65536;

    const size_t tab = (size_t)ntiles * nrows * fvk::TILE_W;
    std::vector<double> h64;
    std::vector<float> h32;
    std::vector<unsigned short> h16;
    try {
        h64.assign(tab, -HUGE_VAL);
        if (full_ok) { h32.assign(tab, -HUGE_VALF); h16.assign(tab, 0xFC00u /* -inf */); }
    } catch (...) { return FV_ERR_NOMEM; }
    std::vector<double> dmax_row(K, 0.0);
    bool ok_range = true;
    std::vector<char> bad(K, 0), big(K, 0);
    parallel_rows(K, [&](int a, int b) {
        for (int k = a; k < b; ++k) {
            const float *src = A + (size_t)k * K;
            for (int i = 0; i < K; ++i) {
                const float x = src[i];
                if (!(x >= 0.0f) || std::isinf(x)) bad[k] = 1;
                if (x > 1.0f) big[k] = 1;
                const double l = std::log((double)x);
                const size_t e = fvk::tab_index<4>(k, i, nrows);
                h64[e] = l;
                if (!full_ok) continue;
                h32[e] = (float)l;
                const _Float16 hl = (_Float16)l;          // round to nearest even; -inf stays -inf
                unsigned short hb;
                std::memcpy(&hb, &hl, 2);
                h16[fvk::tab_index<8>(k, i, nrows)] = hb;
                if (std::isfinite(l)) {
                    // a finite log that overflows binary16 (< -65504) would become -inf: cannot happen for
                    // float32 inputs (log >= -104), but keep the bound honest
                    const double d = std::isfinite((double)hl) ? std::fabs((double)hl - l) : HUGE_VAL;
                    if (d > dmax_row[k]) dmax_row[k] = d;
                }
            }
        }
    });
    std::vector<double> b64((size_t)M * K), pi64(K);
    std::vector<float> b32((size_t)M * K);
    for (int i = 0; i < K; ++i) {
        for (int o = 0; o < M; ++o) {
            const float x = B[(size_t)i * M + o];
            if (!(x >= 0.0f) || std::isinf(x)) ok_range = false;
            if (x > 1.0f) big[0] = 1;
            const double l = std::log((double)x);
            b64[(size_t)o * K + i] = l; b32[(size_t)o * K + i] = (float)l;
        }
        const float x = Pi[i];
        if (!(x >= 0.0f) || std::isinf(x)) ok_range = false;
        if (x > 1.0f) big[0] = 1;
        pi64[i] = std::log((double)x);
    }
    bool any_big = false;
    for (int k = 0; k < K; ++k) { if (bad[k]) ok_range = false; if (big[k]) any_big = true; }
    if (!ok_range) { ctx->detail = "model entries must be finite and >= 0"; return FV_ERR_ARG; }
    // From here on device tables are released and overwritten in place: until every upload has succeeded the
    // context holds NO model (K = 0 makes every decode return FV_ERR_STATE), so a failure half way (e.g. NOMEM
    // on a larger second model) can never pair the old sizes with partly new tables.
    ctx->K = 0; ctx->M = 0; ctx->nrows = 0; ctx->full_ok = false; ctx->u16_ok = false;
    (void)hipStreamSynchronize(ctx->stream);
    ctx->LA64R.release(); ctx->LAQ16R.release();

    double dmax = 0.0;
    for (int k = 0; k < K; ++k) dmax = std::max(dmax, dmax_row[k]);
    ctx->window16 = std::nextafter((float)(2.0 * dmax), HUGE_VALF);     // rounded up
    // fixed-point table: step = (largest finite |log A|) / 65534, code = round(-L/step), 0xffff = -inf.
    // The kernel evaluates fma(code, -step, s) with step as a float, so the error is measured against
    // exactly that product (code * (double)(float)step is exact in double).
    ctx->SPdata.release(); ctx->SPoff.release(); ctx->SPnwb.release();
    ctx->density = 1.0;
    std::vector<unsigned short> hq;
    if (full_ok) {
    try { hq.assign(tab, 0xFFFFu); } catch (...) { return FV_ERR_NOMEM; }
    double lmax = 0.0;
    for (size_t e = 0; e < tab; ++e) if (std::isfinite(h64[e])) lmax = std::max(lmax, -h64[e]);
    const float stepf = lmax > 0.0 ? (float)(lmax / 65534.0) : 1.0f;
    const double stepd = (double)stepf;
    std::vector<double> dq_row(K, 0.0);
    parallel_rows(K, [&](int a, int b) {
        for (int k = a; k < b; ++k)
            for (int i = 0; i < K; ++i) {
                const double l = h64[fvk::tab_index<4>(k, i, nrows)];
                if (!std::isfinite(l)) continue;
                double q = std::nearbyint(-l / stepd);
                if (q < 0) q = 0;
                if (q > 65534.0) q = 65534.0;
                hq[fvk::tab_index<8>(k, i, nrows)] = (unsigned short)q;
                const double d = std::fabs(-q * stepd - l);
                if (d > dq_row[k]) dq_row[k] = d;
            }
    });
    double dqmax = 0.0;
    for (int k = 0; k < K; ++k) dqmax = std::max(dqmax, dq_row[k]);
    ctx->windowq = std::nextafter((float)(2.0 * dqmax), HUGE_VALF);
    ctx->qscale = -stepf;
    FV_HIP(ctx->LAQ16.ensure(tab));
    FV_HIP(hipMemcpy(ctx->LAQ16.p, hq.data(), tab * sizeof(unsigned short), hipMemcpyHostToDevice));
    // sparse form of the same codes (CSC-Q16, see trellis_step_sparse): per tile, per column, the finite
    // entries in ascending k as (k << 16 | code), 4 per lane load, columns padded to the tile's longest
    if (K <= 65536) {
        std::vector<int> off(ntiles), nwbv(ntiles);
        std::vector<std::vector<uint32_t>> cols((size_t)ntiles * fvk::TILE_W);
        parallel_rows(ntiles, [&](int a, int b) {
            for (int tl = a; tl < b; ++tl)
                for (int c = 0; c < fvk::TILE_W; ++c) {
                    const int i = tl * fvk::TILE_W + c;
                    if (i >= K) continue;
                    std::vector<uint32_t> &v = cols[(size_t)tl * fvk::TILE_W + c];
                    for (int k = 0; k < K; ++k) {
                        const unsigned short code = hq[fvk::tab_index<8>(k, i, nrows)];
                        if (code != 0xFFFFu) v.push_back(((uint32_t)k << 16) | code);
                    }
                }
        });
        size_t nnz = 0, total = 0;
        for (int tl = 0; tl < ntiles; ++tl) {
            size_t longest = 0;
            for (int c = 0; c < fvk::TILE_W; ++c) {
                const size_t n = cols[(size_t)tl * fvk::TILE_W + c].size();
                nnz += n; longest = std::max(longest, n);
            }
            const int nch = (int)((longest + 3) / 4);
            nwbv[tl] = (nch + 3) / 4;
            off[tl] = (int)total;
            total += (size_t)nwbv[tl] * 4 * fvk::TILE_W;
        }
        ctx->density = (double)nnz / ((double)K * K);
        if (total < (size_t)1 << 30 && total > 0) {
            std::vector<uint4> sp(total);
            const uint32_t pad = 0x0000FFFFu;           // k = 0, code = 0xffff (-inf)
            parallel_rows(ntiles, [&](int a, int b) {
                for (int tl = a; tl < b; ++tl)
                    for (int j = 0; j < nwbv[tl] * 4; ++j)
                        for (int c = 0; c < fvk::TILE_W; ++c) {
                            const std::vector<uint32_t> &v = cols[(size_t)tl * fvk::TILE_W + c];
                            uint32_t e[4];
                            for (int q = 0; q < 4; ++q) e[q] = (size_t)(4 * j + q) < v.size() ? v[4 * j + q] : pad;
                            sp[(size_t)off[tl] + (size_t)j * fvk::TILE_W + c] = make_uint4(e[0], e[1], e[2], e[3]);
                        }
            });
            FV_HIP(ctx->SPdata.ensure(total));
            FV_HIP(ctx->SPoff.ensure(ntiles));
            FV_HIP(ctx->SPnwb.ensure(ntiles));
            FV_HIP(hipMemcpy(ctx->SPdata.p, sp.data(), total * sizeof(uint4), hipMemcpyHostToDevice));
            FV_HIP(hipMemcpy(ctx->SPoff.p, off.data(), ntiles * sizeof(int), hipMemcpyHostToDevice));
            FV_HIP(hipMemcpy(ctx->SPnwb.p, nwbv.data(), ntiles * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    }   // full_ok
    FV_HIP(ctx->LA64.ensure(tab));
    if (full_ok) {
        FV_HIP(ctx->LA32.ensure(tab));
        FV_HIP(ctx->LA16.ensure(tab));
        FV_HIP(hipMemcpy(ctx->LA16.p, h16.data(), tab * sizeof(unsigned short), hipMemcpyHostToDevice));
        FV_HIP(hipMemcpy(ctx->LA32.p, h32.data(), tab * sizeof(float), hipMemcpyHostToDevice));
    } else {
        ctx->LA32.release(); ctx->LA16.release(); ctx->LAQ16.release();
    }
    FV_HIP(ctx->LB64T.ensure((size_t)M * K));
    FV_HIP(ctx->LB32T.ensure((size_t)M * K));
    FV_HIP(ctx->LPi64.ensure(K));
    FV_HIP(hipMemcpy(ctx->LA64.p, h64.data(), tab * sizeof(double), hipMemcpyHostToDevice));
    FV_HIP(hipMemcpy(ctx->LB64T.p, b64.data(), b64.size() * sizeof(double), hipMemcpyHostToDevice));
    FV_HIP(hipMemcpy(ctx->LB32T.p, b32.data(), b32.size() * sizeof(float), hipMemcpyHostToDevice));
    FV_HIP(hipMemcpy(ctx->LPi64.p, pi64.data(), pi64.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->K = K; ctx->M = M; ctx->nrows = nrows; ctx->full_ok = full_ok; ctx->u16_ok = u16_ok;     // LA64R / LAQ16R: rebuilt on the next beam decode
    ctx->logs_nonpositive = !any_big;
    ctx->stats = fv_stats{};
    ctx->stats.set_model_ms = ms_since(t0);
    ctx->stats.device_bytes = (long long)device_bytes(ctx);
    return FV_OK;
}

extern "C" int fv_set_option(fv_ctx *ctx, int key, long long value)
{
    if (!ctx) return FV_ERR_ARG;
    switch (key) {
    case FV_OPT_KERNEL:
        if (value < FV_KERNEL_AUTO || value > FV_KERNEL_U16_REFINE) return FV_ERR_ARG;
        ctx->opt_kernel = (int)value; return FV_OK;
    case FV_OPT_MAX_BATCH:
        if (value < 1 || value > fvk::MAX_BATCH) return FV_ERR_ARG;
        ctx->opt_max_batch = (int)value; return FV_OK;
    case FV_OPT_PROFILE:
        ctx->opt_profile = value ? 1 : 0; return FV_OK;
    case FV_OPT_SEL_MARGIN:
        if (value < 0 || value > 100000) return FV_ERR_ARG;
        ctx->opt_sel_margin = (float)value * 1e-3f; return FV_OK;
    case FV_OPT_DEBUG:
        ctx->opt_debug = (int)value; return FV_OK;
    default: return FV_ERR_ARG;
    }
}

namespace {
int decode_full_impl(fv_ctx *ctx, const int *ob, int T, int n_split, int mode, int *path_out, float *score_out);
int decode_beam_impl(fv_ctx *ctx, const int *ob, int T, int n_split, int beam_width, int mode, int *path_out, float *score_out);
int decode_checkpoint_impl(fv_ctx *ctx, const int *ob, int T, int step, int *path_out, float *score_out);
}  // namespace

extern "C" int fv_decode_full(fv_ctx *ctx, const int *ob, int T, int n_split, int mode, int *path_out, float *score_out)
{
    if (!ctx) return FV_ERR_ARG;
    return drained(ctx, decode_full_impl(ctx, ob, T, n_split, mode, path_out, score_out));
}

namespace {
int decode_full_impl(fv_ctx *ctx, const int *ob, int T, int n_split, int mode, int *path_out, float *score_out)
{
    if (!ctx || !ob || !path_out || T < 2 || n_split < 1) return FV_ERR_ARG;
    if (ctx->K == 0) return FV_ERR_STATE;
    // Beyond the float32 kernels' LDS limit only the packed 16-bit kernel fits (a row of 16-bit score codes is half
    // the bytes): it needs every model entry in [0,1] and its table is built on the device on first use.
    const bool big = !ctx->full_ok;
    if (big && !(ctx->u16_ok && ctx->logs_nonpositive && (ctx->opt_kernel == FV_KERNEL_AUTO || ctx->opt_kernel == FV_KERNEL_U16_REFINE))) {
        ctx->detail = "full-state decode of K > ~40100 needs the packed 16-bit kernel (FV_KERNEL_AUTO / FV_KERNEL_U16_REFINE, K <= 65536, model entries in [0,1])";
        return FV_ERR_UNSUPPORTED;
    }
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= ctx->M) return FV_ERR_ARG;
    if (ctx->opt_kernel >= FV_KERNEL_F32_REFINE && !ctx->logs_nonpositive) {
        ctx->detail = "the filter+refine kernels need every model entry in [0,1]";
        return FV_ERR_UNSUPPORTED;
    }
    auto t0 = clk::now();
    FV_HIP(hipSetDevice(ctx->device));
    fv::Plan plan;
    int rc = fv::build_plan(T, n_split, mode, ctx->nranks, plan);
    if (rc) return rc;
    const int kernel = big ? FV_KERNEL_U16_REFINE : pick_kernel(ctx);
    if (big && !ctx->LAQ16.p) {
        const int ntiles = (ctx->K + fvk::TILE_W - 1) / fvk::TILE_W;
        const size_t tab = (size_t)ntiles * ctx->nrows * fvk::TILE_W;
        FV_HIP(ctx->LAQ16.ensure(tab));
        FV_HIP(ctx->d_qaux.ensure(3));
        FV_HIP(hipMemsetAsync(ctx->d_qaux.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(fvk::q16_tile_range, dim3(2048), dim3(256), 0, ctx->stream, ctx->LA64.p, tab, ctx->d_qaux.p);
        hipLaunchKernelGGL(fvk::q16_tile_codes, dim3(4096), dim3(256), 0, ctx->stream, ctx->LA64.p, ctx->LAQ16.p, ctx->K, ctx->nrows,
                           ntiles, ctx->d_qaux.p, ctx->d_qaux.p + 1);
        FV_HIP(hipGetLastError());
        unsigned long long bits[2] = { 0, 0 };
        FV_HIP(hipMemcpyAsync(bits, ctx->d_qaux.p, sizeof bits, hipMemcpyDeviceToHost, ctx->stream));
        FV_HIP(hipStreamSynchronize(ctx->stream));
        double lmax, dqmax;
        std::memcpy(&lmax, &bits[0], 8); std::memcpy(&dqmax, &bits[1], 8);
        const float stepf = lmax > 0.0 ? (float)(lmax / 65534.0) : 1.0f;
        ctx->windowq = std::nextafter((float)(2.0 * dqmax), HUGE_VALF);
        ctx->qscale = -stepf;
    }

    // generations of passes this rank runs
    std::vector<std::vector<fv::Pass>> gens(plan.generations());
    size_t most = 1;
    for (const fv::Pass &p : plan.passes)
        if (p.owner < 0 || p.owner % ctx->nranks == ctx->rank) gens[p.generation].push_back(p);
    for (auto &g : gens) most = std::max(most, g.size());
    if ((rc = ensure_workspace(ctx, T, most))) return rc;

    const double keep_model_ms = ctx->stats.set_model_ms;
    ctx->stats = fv_stats{};
    ctx->stats.set_model_ms = keep_model_ms;
    ctx->stats.kernel = kernel;
    ctx->stats.generations = plan.generations();
    ctx->stats.table_bytes_per_step = (long long)((ctx->K + fvk::TILE_W - 1) / fvk::TILE_W) * ctx->nrows * fvk::TILE_W * (kernel == FV_KERNEL_F64_STREAM ? 8 : kernel == FV_KERNEL_F32_REFINE ? 4 : 2);
    if (kernel == FV_KERNEL_SPARSE_Q16) ctx->stats.table_bytes_per_step = (long long)ctx->SPdata.bytes();
    ctx->stats.density = ctx->density;

    ctx->h_ob.assign(ob, ob + T);
    FV_HIP(hipMemcpyAsync(ctx->d_ob.p, ctx->h_ob.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FV_HIP(hipMemsetAsync(ctx->d_counters.p, 0, FV_NCOUNTERS * sizeof(unsigned long long), ctx->stream));
    FV_HIP(hipMemsetAsync(ctx->d_ans.p, 0, (size_t)T * sizeof(int), ctx->stream));
    FV_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
    size_t nprof = 0;
    ctx->fork_active = false;
    for (size_t g = 0; g < gens.size(); ++g) {
        ctx->stats.passes += (int)gens[g].size();
        if ((rc = run_generation_full(ctx, gens[g], kernel, nprof))) return rc;
        if (g == 0) FV_HIP(hipEventRecord(ctx->ev_top, ctx->stream));
    }
    ctx->stats.cells = ctx->stats.task_steps * (long long)ctx->K * ctx->K + ctx->stats.column_steps * (long long)ctx->K;
    ctx->stats.alg_bytes = 4 * ctx->stats.cells;
    return finish_decode(ctx, plan, T, path_out, score_out, t0, nprof, false);
}
}  // namespace

namespace {

// row pitch of the row-major float64 table the beam kernels gather from
inline int beam_ld(int K) { return (K + fvb::BEAM_COLS - 1) / fvb::BEAM_COLS * fvb::BEAM_COLS; }
inline int beam_ldq(int K) { return (K + fvb::BEAMQ_COLS - 1) / fvb::BEAMQ_COLS * fvb::BEAMQ_COLS; }   // ... of the 16-bit one

// One generation of beam passes in lock-step (same shape as run_generation_full).  Buffers are indexed
// by absolute time j (passes of one generation cover disjoint time ranges): scores_all[j] = the K
// scores after consuming ob[j] (j = L: the init row), set_*[j] = the members of the heap built from
// them (order-free), slot_*[j] = its exact array layout (rebuilt after the lock-step loop).
// Streams a generation of np passes is dealt to (pass i of the length-sorted list goes to group i % n).
// Only for big steps (cfg5: 64 M cells per pass and step; FV_OPT_DEBUG bit 17 forces it): once the auxiliary queues
// have carried work, every dispatch on the main stream takes ~2 us longer (measured, also with the main stream at
// high priority) — 1 ms over the whole-sequence pass of cfg4, more than the overlap returns there.
inline int beam_groups(const fv_ctx *ctx, int np, int beam)
{
    if ((ctx->opt_debug & 65536) || np < 2) return 1;            // FV_OPT_DEBUG bit 16: one stream
    if (!(ctx->opt_debug & 131072) && (double)beam * ctx->K < 16e6) return 1;
    return std::min(1 + fv_ctx::BEAM_AUX, np);
}

int run_generation_beam(fv_ctx *ctx, const std::vector<fv::Pass> &passes, size_t pass_off, int beam, int T)
{
    const int K = ctx->K, np = (int)passes.size();
    if (np == 0) return 0;
    FV_HIP(hipMemsetAsync(ctx->d_tie_count.p, 0, sizeof(unsigned int), ctx->stream));
    const int cand_cap = (ctx->opt_debug & 1024) ? 0 : fvb::cand_cap_for(K, beam);     // FV_OPT_DEBUG bit 10: no candidate lists
    // passes arrive group by group (decode_beam_impl), longest first inside a group; their first positions are in
    // d_passL[pass_off ..] in the same order
    fvb::BeamBase bb;
    bb.scores_all = ctx->d_scores.p; bb.hval = ctx->d_hval.p; bb.hstate = ctx->d_hstate.p;
    bb.cut = ctx->d_cut.p; bb.passL = ctx->d_passL.p + pass_off;
    // members of the heaps of passes [first, first + count) at lock-step s: one launch
    auto select = [&](int first, int count, int s, hipStream_t st) -> int {
        fvb::SelArgs a;
        a.counters = ctx->d_counters.p; a.K = K; a.beam = beam; a.s = s;
        a.no_wave = (ctx->opt_debug & 32768) ? 1 : 0;
        a.margin = ctx->opt_sel_margin; a.cand_cap = cand_cap;
        a.cand = ctx->d_cand.p; a.cand_count = ctx->d_cand_count.p; a.b = bb;
        a.b.passL = bb.passL + first;
        const bool listed = count <= fvb::BEAM_CHUNK;
        for (int q = 0; listed && q < count; ++q) {
            const int j = passes[first + q].L + s;
            a.p[q] = fvb::SelJob{ ctx->d_scores.p + (size_t)j * K, ctx->d_hval.p + (size_t)j * beam, ctx->d_hstate.p + (size_t)j * beam,
                                  ctx->d_cut.p + (size_t)j * fvb::CUT_W, s >= 1 ? ctx->d_cut.p + (size_t)(j - 1) * fvb::CUT_W : nullptr,
                                  (cand_cap && s >= 1) ? ctx->d_cand.p + (size_t)j * cand_cap : nullptr, ctx->d_cand_count.p + j };
        }
        // steps >= 2 of a pass have a candidate list (the predictor needs two cut values)
        fvb::SelKernel lean = s >= 2 ? fvb::sel_cand_kernel_for(K, cand_cap, listed) : nullptr;
        hipLaunchKernelGGL(lean ? lean : fvb::sel_kernel_for(K, listed), dim3(count), dim3(fvb::SEL_BLOCK), fvb::sel_lds(beam), st, a);
        FV_HIP(hipGetLastError());
        return 0;
    };
    // init scores (the same rows the full variant starts from, FLASH_BS:407-427)
    for (int base = 0; base < np; base += fvk::PASS_CHUNK) {
        fvk::PassChunk ch;
        ch.n = std::min(fvk::PASS_CHUNK, np - base);
        for (int q = 0; q < ch.n; ++q) {
            const fv::Pass &p = passes[base + q];
            ch.p[q] = fvk::PassDesc{ p.L, p.R, p.from_pi ? 1 : 0, p.whole ? 1 : 0, (long long)p.L * K };
        }
        hipLaunchKernelGGL(fvk::init_rows, dim3((K + 255) / 256, ch.n), dim3(256), 0, ctx->stream, ch,
                           ctx->LA64.p, ctx->nrows, ctx->LB64T.p, ctx->LPi64.p, ctx->d_ob.p, ctx->d_ans.p,
                           ctx->d_scores.p, K);
        FV_HIP(hipGetLastError());
    }
    // Every group runs its passes in lock-step on its own stream: first heaps' members, then step + select per position.
    // A select launch lasts as long as its slowest exact replay and keeps one CU per pass busy; the step kernels of
    // the other groups fill the rest of the chip meanwhile.
    const int ng = beam_groups(ctx, np, beam);
    if (ng > 1) FV_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    int rc = 0;
    for (int g = 0, first = 0; g < ng; ++g) {
        const int gn = (np - g + ng - 1) / ng;                 // passes g, g + ng, ... of the sorted list
        hipStream_t st = g == 0 ? ctx->stream : ctx->aux[g - 1];
        if (g > 0) FV_HIP(hipStreamWaitEvent(st, ctx->ev_fork, 0));
        if ((rc = select(first, gn, 0, st))) return rc;
        const int maxlen = passes[first].R - passes[first].L;
        int active = gn;
        for (int s = 1; s <= maxlen; ++s) {
            while (active > 0 && passes[first + active - 1].R - passes[first + active - 1].L < s) --active;
            for (int base = 0; base < active; base += fvb::BEAM_CHUNK) {
                fvb::BeamStepArgs a;
                a.LA64R = ctx->LA64R.p; a.tie_count = ctx->d_tie_count.p; a.tie_list = ctx->d_tie_list.p;
                a.tie_cap = (unsigned int)ctx->d_tie_list.n;
                a.counters = ctx->d_counters.p;
                a.K = K; a.ld = beam_ld(K); a.ldq = beam_ldq(K); a.beam = beam;
                a.LAQ16R = ctx->LAQ16R.p; a.qpar = ctx->LAQ16R.p ? reinterpret_cast<const float *>(ctx->d_qaux.p + 2) : nullptr;
                a.cand = ctx->d_cand.p; a.cand_count = ctx->d_cand_count.p; a.cand_cap = cand_cap;
                a.n = std::min(fvb::BEAM_CHUNK, active - base);
                for (int q = 0; q < a.n; ++q) {
                    const int j = passes[first + base + q].L + s;
                    a.p[q].sval = ctx->d_hval.p + (size_t)(j - 1) * beam;
                    a.p[q].sstate = ctx->d_hstate.p + (size_t)(j - 1) * beam;
                    a.p[q].scores = ctx->d_scores.p + (size_t)j * K;
                    a.p[q].bp_row = ctx->d_bp.p + (size_t)j * K;
                    a.p[q].tmp_row = ctx->LB32T.p + (size_t)ctx->h_ob[j] * K;
                    a.p[q].j = j;
                    a.p[q].cut = ctx->d_cut.p + (size_t)(j - 1) * fvb::CUT_W;
                    a.p[q].dupwin = ctx->d_dupwin.p + j;
                }
                // The 16-bit filter kernel moves a quarter of the bytes but has two more dependent phases (window,
                // float64 refine): measured at K = 16384, B = 256 it takes 13.7 us + 2.6 us per extra pass of the
                // launch against 10.6 + 5.0 for the float64 kernel, so it is used from ~80 MB of float64 rows per
                // launch on (cfg5: 537 MB per pass).  FV_OPT_DEBUG bit 8: never, bit 9: always.
                const bool use_q16 = ctx->LAQ16R.p && !(ctx->opt_debug & 256) &&
                                     ((ctx->opt_debug & 512) || (double)a.n * beam * K * 8.0 >= 80e6);
                if (use_q16)
                    hipLaunchKernelGGL(fvb::beam_step_q16, dim3(beam_ldq(K) / fvb::BEAMQ_COLS, a.n), dim3(fvb::BEAM_BLOCK),
                                       fvb::beam_step_q16_lds(beam), st, a);
                else
                    hipLaunchKernelGGL(fvb::beam_step, dim3(beam_ld(K) / fvb::BEAM_COLS, a.n), dim3(fvb::BEAM_BLOCK), fvb::beam_step_lds(beam),
                                       st, a);
                FV_HIP(hipGetLastError());
                ctx->stats.step_launches += 1;
                ctx->stats.task_steps += a.n;
            }
            if ((rc = select(first, active, s, st))) return rc;
        }
        if (g > 0) {
            FV_HIP(hipEventRecord(ctx->ev_join[g - 1], st));
            FV_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[g - 1], 0));
        }
        first += gn;
    }
    // Pass ends.  First attempt on the provisional back-pointers (beam_end_backtrack); only the whole-sequence pass
    // needs a layout for that — its last heap's.  The exact layouts of every step's heap and the tie fix-up are queued
    // behind it but run only if the walk met a tied cell (FV_OPT_DEBUG bit 19: always).
    const bool lazy = !(ctx->opt_debug & 524288);
    FV_HIP(hipMemsetAsync(ctx->d_needfull.p, 0, sizeof(int), ctx->stream));
    auto layouts = [&](bool last_only, const int *gate) -> int {
        for (int base = 0; base < np; base += fvb::HEAP_CHUNK) {
            fvb::HeapAllArgs h;
            h.scores_all = ctx->d_scores.p; h.slot_val = ctx->d_slot_val.p; h.slot_state = ctx->d_slot_state.p;
            h.err_counter = ctx->d_counters.p + 5; h.gate = gate;
            h.K = K; h.beam = beam; h.n = 0;
            int longest = 0;
            for (int q = 0; q < std::min(fvb::HEAP_CHUNK, np - base); ++q) {
                const fv::Pass &p = passes[base + q];
                if (last_only && !p.whole) continue;
                h.p[h.n++] = fvb::HeapRange{ last_only ? p.R : p.L, p.R };
                longest = std::max(longest, last_only ? 1 : p.R - p.L + 1);
            }
            if (h.n == 0) continue;
            hipLaunchKernelGGL(fvb::heap_build_all, dim3(longest, h.n), dim3(128), fvb::heap_lds(beam), ctx->stream, h);
            FV_HIP(hipGetLastError());
        }
        return 0;
    };
    auto ends = [&](int lazy_walk) -> int {
        for (int base = 0; base < np; base += fvb::BEAM_CHUNK) {
            fvb::BeamEndArgs e;
            e.K = K; e.beam = beam; e.n = std::min(fvb::BEAM_CHUNK, np - base);
            e.lazy = lazy_walk; e.flag = ctx->d_needfull.p;
            for (int q = 0; q < e.n; ++q) e.p[q] = fvb::BeamEnd{ passes[base + q].L, passes[base + q].R, passes[base + q].whole ? 1 : 0 };
            hipLaunchKernelGGL(fvb::beam_end_backtrack, dim3(e.n), dim3(64), 0, ctx->stream, e, ctx->d_slot_val.p,
                               ctx->d_slot_state.p, ctx->d_hstate.p, ctx->d_bp.p, ctx->d_ans.p, ctx->d_score.p);
            FV_HIP(hipGetLastError());
        }
        return 0;
    };
    if (lazy) {
        if ((rc = layouts(true, nullptr))) return rc;
        if ((rc = ends(1))) return rc;
    } else {
        int one = 1;
        FV_HIP(hipMemcpyAsync(ctx->d_needfull.p, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        FV_HIP(hipStreamSynchronize(ctx->stream));       // (`one` is a local; this is the experiment path)
    }
    if ((rc = layouts(false, ctx->d_needfull.p))) return rc;
    {
        fvb::FixArgs f;
        f.LA64R = ctx->LA64R.p; f.LB32T = ctx->LB32T.p; f.ob = ctx->d_ob.p;
        f.tie_count = ctx->d_tie_count.p; f.tie_list = ctx->d_tie_list.p; f.tie_cap = (unsigned int)ctx->d_tie_list.n;
        f.slot_val = ctx->d_slot_val.p; f.slot_state = ctx->d_slot_state.p; f.bp = ctx->d_bp.p;
        f.K = K; f.ld = beam_ld(K); f.beam = beam; f.total = ctx->d_counters.p + 6; f.gate = ctx->d_needfull.p;
        hipLaunchKernelGGL(fvb::tie_fixup, dim3(512), dim3(256), 0, ctx->stream, f);
        FV_HIP(hipGetLastError());
    }
    if ((rc = ends(0))) return rc;
    (void)T;
    return 0;
}

}  // namespace

extern "C" int fv_decode_beam(fv_ctx *ctx, const int *ob, int T, int n_split, int beam_width, int mode,
                              int *path_out, float *score_out)
{
    if (!ctx) return FV_ERR_ARG;
    return drained(ctx, decode_beam_impl(ctx, ob, T, n_split, beam_width, mode, path_out, score_out));
}

namespace {
int decode_beam_impl(fv_ctx *ctx, const int *ob, int T, int n_split, int beam_width, int mode, int *path_out, float *score_out)
{
    if (!ctx || !ob || !path_out || T < 2 || n_split < 1) return FV_ERR_ARG;
    if (ctx->K == 0) return FV_ERR_STATE;
    // beam > K reads uninitialised heap slots in the reference
    if (beam_width < 2 || beam_width > ctx->K) return FV_ERR_ARG;
    if (fvb::beam_step_lds(beam_width) > 150 * 1024 || fvb::beam_step_q16_lds(beam_width) > 150 * 1024 ||
        fvb::heap_lds(beam_width) > 150 * 1024) return FV_ERR_UNSUPPORTED;
    if (ctx->K > fvb::SEL_MAX_ROUNDS * fvb::SEL_BLOCK) return FV_ERR_UNSUPPORTED;      // topb_select: one bit per round
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= ctx->M) return FV_ERR_ARG;
    auto t0 = clk::now();
    FV_HIP(hipSetDevice(ctx->device));
    fv::Plan plan;
    int rc = fv::build_plan(T, n_split, mode, ctx->nranks, plan);
    if (rc) return rc;
    std::vector<std::vector<fv::Pass>> gens(plan.generations());
    size_t most = 1;
    for (const fv::Pass &p : plan.passes)
        if (p.owner < 0 || p.owner % ctx->nranks == ctx->rank) gens[p.generation].push_back(p);
    for (auto &g : gens) most = std::max(most, g.size());
    if ((rc = ensure_workspace(ctx, T, 1))) return rc;
    (void)most;
    FV_HIP(ctx->d_scores.ensure((size_t)T * ctx->K));
    FV_HIP(ctx->d_hval.ensure((size_t)T * beam_width));
    FV_HIP(ctx->d_hstate.ensure((size_t)T * beam_width));
    FV_HIP(ctx->d_slot_val.ensure((size_t)T * beam_width));
    FV_HIP(ctx->d_slot_state.ensure((size_t)T * beam_width));
    FV_HIP(ctx->d_tie_list.ensure((size_t)T * ctx->K));
    FV_HIP(ctx->d_tie_count.ensure(4));
    FV_HIP(ctx->d_cut.ensure((size_t)T * fvb::CUT_W));
    FV_HIP(ctx->d_cand_count.ensure(T));
    FV_HIP(hipMemsetAsync(ctx->d_cand_count.p, 0, (size_t)T * sizeof(int), ctx->stream));
    if (const int cap = fvb::cand_cap_for(ctx->K, beam_width)) FV_HIP(ctx->d_cand.ensure((size_t)T * cap));
    FV_HIP(ctx->d_dupwin.ensure(T));
    FV_HIP(ctx->d_needfull.ensure(4));
    FV_HIP(hipMemsetAsync(ctx->d_dupwin.p, 0, (size_t)T * sizeof(int), ctx->stream));
    if (!ctx->LA64R.p) {
        const int ld = beam_ld(ctx->K);
        FV_HIP(ctx->LA64R.ensure((size_t)ctx->K * ld));
        hipLaunchKernelGGL(fvb::relayout_rows, dim3(2048), dim3(256), 0, ctx->stream, ctx->LA64.p, ctx->LA64R.p, ctx->K, ctx->nrows, ld);
        FV_HIP(hipGetLastError());
    }
    if (!ctx->LAQ16R.p && ctx->logs_nonpositive) {
        // filter table of beam_step_q16, quantised on the device from LA64R
        const int ld = beam_ld(ctx->K), ldq = beam_ldq(ctx->K);
        FV_HIP(ctx->LAQ16R.ensure((size_t)ctx->K * ldq));
        FV_HIP(ctx->d_qaux.ensure(3));
        FV_HIP(hipMemsetAsync(ctx->d_qaux.p, 0, 3 * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(fvb::q16_range, dim3(2048), dim3(256), 0, ctx->stream, ctx->LA64R.p, (size_t)ctx->K * ld, ctx->d_qaux.p);
        hipLaunchKernelGGL(fvb::q16_rows, dim3(2048), dim3(256), 0, ctx->stream, ctx->LA64R.p, ctx->LAQ16R.p, ctx->K, ld, ldq,
                           ctx->d_qaux.p, ctx->d_qaux.p + 1);
        hipLaunchKernelGGL(fvb::q16_params, dim3(1), dim3(1), 0, ctx->stream, ctx->d_qaux.p, ctx->d_qaux.p + 1,
                           reinterpret_cast<float *>(ctx->d_qaux.p + 2));
        FV_HIP(hipGetLastError());
    }

    const double keep_model_ms = ctx->stats.set_model_ms;
    ctx->stats = fv_stats{};
    ctx->stats.set_model_ms = keep_model_ms;
    ctx->stats.kernel = FV_KERNEL_F64_STREAM;
    ctx->stats.generations = plan.generations();
    ctx->stats.table_bytes_per_step = (long long)beam_width * ctx->K * 8;

    ctx->h_ob.assign(ob, ob + T);
    FV_HIP(hipMemcpyAsync(ctx->d_ob.p, ctx->h_ob.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    // The passes of a generation run in lock-step, longest first (the active ones are a prefix); the kernels find a
    // pass's rows from its first position, so the whole plan's pass lists go to the device once, before the clock starts.
    std::vector<size_t> pass_off(gens.size(), 0);
    ctx->h_passL.clear();
    for (size_t g = 0; g < gens.size(); ++g) {
        std::stable_sort(gens[g].begin(), gens[g].end(),
                         [](const fv::Pass &a, const fv::Pass &b) { return a.R - a.L > b.R - b.L; });
        {   // group-major: the passes of stream group q (i % ng == q), longest first, then those of group q + 1
            const int np = (int)gens[g].size(), ng = beam_groups(ctx, np, beam_width);
            std::vector<fv::Pass> byg;
            byg.reserve(gens[g].size());
            for (int q = 0; q < ng; ++q)
                for (int i = q; i < np; i += ng) byg.push_back(gens[g][i]);
            gens[g].swap(byg);
        }
        pass_off[g] = ctx->h_passL.size();
        for (const fv::Pass &p : gens[g]) ctx->h_passL.push_back(p.L);
    }
    FV_HIP(ctx->d_passL.ensure(std::max<size_t>(1, ctx->h_passL.size())));
    if (!ctx->h_passL.empty())
        FV_HIP(hipMemcpyAsync(ctx->d_passL.p, ctx->h_passL.data(), ctx->h_passL.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FV_HIP(hipMemsetAsync(ctx->d_counters.p, 0, FV_NCOUNTERS * sizeof(unsigned long long), ctx->stream));
    FV_HIP(hipMemsetAsync(ctx->d_ans.p, 0, (size_t)T * sizeof(int), ctx->stream));
    FV_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
    FV_HIP(hipEventRecord(ctx->ev_s0, ctx->stream));
    for (size_t g = 0; g < gens.size(); ++g) {
        ctx->stats.passes += (int)gens[g].size();
        // Work queued on the auxiliary streams slows every dispatch of the main one while it waits there for its fork
        // event (the command processor keeps re-examining the blocked queues: +2 us per launch, 1 ms over the
        // whole-sequence pass of cfg4).  The host therefore does not run ahead of a serial generation into a forked one.
        if (g > 0 && beam_groups(ctx, (int)gens[g].size(), beam_width) > 1 && beam_groups(ctx, (int)gens[g - 1].size(), beam_width) == 1)
            FV_HIP(hipStreamSynchronize(ctx->stream));
        if ((rc = run_generation_beam(ctx, gens[g], pass_off[g], beam_width, T))) return rc;
        if (g == 0) { FV_HIP(hipEventRecord(ctx->ev_top, ctx->stream)); FV_HIP(hipEventRecord(ctx->ev_s1, ctx->stream)); }
    }
    ctx->stats.cells = ctx->stats.task_steps * (long long)ctx->K * beam_width;
    ctx->stats.alg_bytes = 4 * ctx->stats.cells;
    return finish_decode(ctx, plan, T, path_out, score_out, t0, 0, true);
}
}  // namespace

extern "C" int fv_decode_vanilla(fv_ctx *ctx, const int *ob, int T, int *path_out, float *score_out)
{
    if (!ctx) return FV_ERR_ARG;
    const int keep_kernel = ctx->opt_kernel;
    ctx->opt_kernel = FV_KERNEL_F64_STREAM;      // the baseline's expression has no filter form
    ctx->vanilla = 1;
    int rc = fv_decode_full(ctx, ob, T, 1, FV_MODE_SINGLE_PASS, path_out, score_out);
    ctx->vanilla = 0;
    ctx->opt_kernel = keep_kernel;
    return rc;
}

// checkpoint Viterbi.c:176-251 on the device.  First pass: T-1 steps of the baseline's recurrence with the
// score row of every step that is a multiple of `step` written straight into its checkpoint slot (the next
// step reads it from there; the arg rows of this pass are scratch).  Second pass: every segment
// [c, next checkpoint] restarts from its kept row and re-runs its steps, this time keeping the arg rows;
// the segments are independent, so they advance in lock-step and share table sweeps (up to 8 per launch)
// instead of running last-to-first as the CPU program does.  End state and back-track as in vanilla.
extern "C" int fv_decode_checkpoint(fv_ctx *ctx, const int *ob, int T, int step, int *path_out, float *score_out)
{
    if (!ctx) return FV_ERR_ARG;
    return drained(ctx, decode_checkpoint_impl(ctx, ob, T, step, path_out, score_out));
}

namespace {
int decode_checkpoint_impl(fv_ctx *ctx, const int *ob, int T, int step, int *path_out, float *score_out)
{
    if (!ctx || !ob || !path_out || T < 2) return FV_ERR_ARG;
    if (ctx->K == 0) return FV_ERR_STATE;
    if (!ctx->full_ok) { ctx->detail = "full-state decode needs one score row in LDS (K <= ~40100)"; return FV_ERR_UNSUPPORTED; }
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= ctx->M) return FV_ERR_ARG;
    if (step <= 0) step = (int)std::floor(std::sqrt(1.0 * T));        // checkpoint Viterbi.c:179-180
    auto t0 = clk::now();
    FV_HIP(hipSetDevice(ctx->device));
    const int K = ctx->K, nrows = ctx->nrows;
    const int nck = (T + step - 1) / step;
    fv::Plan plan;
    int rc = fv::build_plan(T, 1, FV_MODE_SINGLE_PASS, 1, plan);
    if (rc) return rc;
    if ((rc = ensure_workspace(ctx, T, (size_t)nck + 1))) return rc;  // two rows per segment + two for the first pass
    if (ctx->d_ckpt.n < (size_t)nck * nrows) {
        FV_HIP(ctx->d_ckpt.ensure((size_t)nck * nrows));
        FV_HIP(hipMemsetAsync(ctx->d_ckpt.p, 0, (size_t)nck * nrows * sizeof(float), ctx->stream));   // row pads stay zero
    }
    const double keep_model_ms = ctx->stats.set_model_ms;
    ctx->stats = fv_stats{};
    ctx->stats.set_model_ms = keep_model_ms;
    ctx->stats.kernel = FV_KERNEL_F64_STREAM;
    ctx->stats.generations = 2;
    ctx->stats.passes = 1 + nck;
    ctx->stats.table_bytes_per_step = (long long)((K + fvk::TILE_W - 1) / fvk::TILE_W) * nrows * fvk::TILE_W * 8;
    ctx->stats.density = ctx->density;
    ctx->h_ob.assign(ob, ob + T);
    FV_HIP(hipMemcpyAsync(ctx->d_ob.p, ctx->h_ob.data(), (size_t)T * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    FV_HIP(hipMemsetAsync(ctx->d_counters.p, 0, FV_NCOUNTERS * sizeof(unsigned long long), ctx->stream));
    FV_HIP(hipMemsetAsync(ctx->d_ans.p, 0, (size_t)T * sizeof(int), ctx->stream));
    FV_HIP(hipEventRecord(ctx->ev_start, ctx->stream));

    struct Restore { fv_ctx *c; ~Restore() { c->vanilla = 0; } } restore{ ctx };
    ctx->vanilla = 1;
    auto ckpt = [&](int c) { return ctx->d_ckpt.p + (size_t)c * nrows; };
    auto scratch = [&](int q, int parity) { return ctx->d_rows.p + ((size_t)q * 2 + parity) * nrows; };
    auto slot_for = [&](const float *in, float *out, int j) {
        fvk::TaskSlot sl;
        sl.t1_in = in; sl.t1_out = out;
        sl.tmp_row = ctx->LB32T.p + (size_t)ctx->h_ob[j] * K;
        sl.tmp64_row = ctx->LB64T.p + (size_t)ctx->h_ob[j] * K;
        sl.bp_out = ctx->d_bp.p + (size_t)j * K;
        return sl;
    };
    {   // initT1 (:119) into checkpoint 0
        fvk::PassChunk ch;
        ch.n = 1;
        ch.p[0] = fvk::PassDesc{ 0, T - 1, 1, 1, 0 };
        hipLaunchKernelGGL(fvk::init_rows, dim3((K + 255) / 256, 1), dim3(256), 0, ctx->stream, ch, ctx->LA64.p, nrows,
                           ctx->LB64T.p, ctx->LPi64.p, ctx->d_ob.p, ctx->d_ans.p, ctx->d_ckpt.p, K);
        FV_HIP(hipGetLastError());
    }
    // first pass (:213-232)
    auto row_after = [&](int j) -> float * { return j % step == 0 ? ckpt(j / step) : scratch(nck, j & 1); };
    FV_HIP(hipEventRecord(ctx->ev_s0, ctx->stream));
    for (int j = 1; j < T; ++j) {
        fvk::TaskSlot sl = slot_for(row_after(j - 1), row_after(j), j);
        if ((rc = launch_step_kernel(ctx, FV_KERNEL_F64_STREAM, &sl, 1, j & 1))) return rc;
        ctx->stats.step_launches += 1;
        ctx->stats.task_steps += 1;
    }
    FV_HIP(hipEventRecord(ctx->ev_s1, ctx->stream));
    FV_HIP(hipEventRecord(ctx->ev_top, ctx->stream));
    // second pass (:236-248, subroutine :121-174): segment c = times c*step .. min((c+1)*step, T-1)
    std::vector<int> order(nck);
    for (int c = 0; c < nck; ++c) order[c] = c;
    auto seg_len = [&](int c) { return std::min((c + 1) * step, T - 1) - c * step; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return seg_len(a) > seg_len(b); });
    const int cap = std::max(1, std::min(ctx->opt_max_batch, max_batch_for(nrows, false)));
    const int maxlen = seg_len(order[0]);
    int active = nck;
    for (int s = 1; s <= maxlen; ++s) {
        while (active > 0 && seg_len(order[active - 1]) < s) --active;
        for (int base = 0; base < active; base += cap) {
            const int nb = std::min(cap, active - base);
            fvk::TaskSlot slots[fvk::MAX_BATCH];
            for (int q = 0; q < nb; ++q) {
                const int c = order[base + q];
                slots[q] = slot_for(s == 1 ? ckpt(c) : scratch(c, (s - 1) & 1), scratch(c, s & 1), c * step + s);
            }
            if ((rc = launch_step_kernel(ctx, FV_KERNEL_F64_STREAM, slots, nb, s & 1))) return rc;
            ctx->stats.step_launches += 1;
            ctx->stats.task_steps += nb;
        }
    }
    // end state (:152-165) from the first pass's last row, then the back-track through the kept arg rows (:167-171)
    hipLaunchKernelGGL(fvk::final_argmax, dim3(1), dim3(1024), 0, ctx->stream, row_after(T - 1), K, ctx->d_ans.p + (T - 1),
                       ctx->d_score.p);
    FV_HIP(hipGetLastError());
    {
        fvk::PassChunk ch;
        ch.n = 1;
        ch.p[0] = fvk::PassDesc{ 0, T - 1, 1, 1, 0 };
        hipLaunchKernelGGL(fvk::backtrack, dim3(1), dim3(64), 0, ctx->stream, ch, ctx->d_bp.p, K, ctx->d_ans.p);
        FV_HIP(hipGetLastError());
    }
    ctx->stats.cells = ctx->stats.task_steps * (long long)K * K;
    ctx->stats.alg_bytes = 4 * ctx->stats.cells;
    return finish_decode(ctx, plan, T, path_out, score_out, t0, 0, false);
}
}  // namespace

extern "C" long long fv_checkpoint_memory_bytes(int K, int T, int step)
{
    if (step <= 0) step = (int)std::floor(std::sqrt(1.0 * T));
    const long long nck = (T + step - 1) / step;
    const long long last = T - (nck - 1) * step;
    const long long tsub = nck > 1 && step + 1 > last ? step + 1 : last;       // T_sub, checkpoint Viterbi.c:123
    return 4LL * K + 4LL * K * nck + 4LL * K + 4LL * (T / step + 1) + 8LL * K * tsub;   // :250
}

#ifdef FV_REPLAY_PROF
// Experiment builds only: read and reset the replay profile (fv_beam_kernels.hip.inc, replay_prof).
extern "C" int fv_debug_replay_prof(unsigned long long *out8)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(fvb::replay_prof), sizeof z) != hipSuccess) return FV_ERR_DEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fvb::replay_prof), z, sizeof z) != hipSuccess) return FV_ERR_DEVICE;
    return FV_OK;
}
#endif

extern "C" int fv_last_stats(const fv_ctx *ctx, fv_stats *out)
{
    if (!ctx || !out) return FV_ERR_ARG;
    *out = ctx->stats;
    return FV_OK;
}

extern "C" const char *fv_last_error_detail(const fv_ctx *ctx) { return ctx ? ctx->detail.c_str() : ""; }

extern "C" const char *fv_strerror(int rc)
{
    switch (rc) {
    case FV_OK: return "ok";
    case FV_WARN_BEAM_MISS: return "beam miss: path holds -1 entries, as the reference prints";
    case FV_ERR_ARG: return "bad argument";
    case FV_ERR_NOMEM: return "out of memory";
    case FV_ERR_NO_PRED: return "decoded entry has no finite predecessor";
    case FV_ERR_DEVICE: return "HIP error";
    case FV_ERR_STATE: return "call out of order (no model / no communicator)";
    case FV_ERR_UNSUPPORTED: return "size or option not supported by this build";
    case FV_ERR_COMM: return "RCCL error";
    default: return "unknown flashvit error";
    }
}

extern "C" long long fv_reference_memory_bytes(int K, int T, int n_split, int beam_width)
{
    // sizeof(ThreadPool) on x86-64 glibc: mutex 40 + cond 48 + N pthread_t + 3 ints, padded to 8
    const long long N = n_split;
    const long long pool = ((40 + 48 + 8 * N + 12 + 7) / 8) * 8;
    long long mem = 0, tmp;
    const bool nway = N > 2 && T >= 2 * N;
    if (beam_width <= 0) {
        if (nway) mem = 4 * (N - 1) + 2LL * K * 4 + 2 * (N - 1) * (long long)K * 4;    // FLASH:355
        tmp = N * (2LL * K * 4 + 2LL * K * 4);                                         // :364
    } else {
        if (nway) mem = 4 * (N - 1) + 2 * (N - 1) * (long long)(beam_width + 1) * 12;  // FLASH_BS:564
        tmp = N * (2LL * (beam_width + 1) * 12);                                       // :573
    }
    if (tmp > mem) mem = tmp;
    return mem + pool + 8;     // + sizeof(ThreadPool) + sizeof(size_t) (the sizeof(expr) quirk, FLASH:367)
}

extern "C" int fv_merge_paths(int T, int n_split, int nranks, const int *gathered, int *path_out)
{
    if (!gathered || !path_out || nranks < 1) return FV_ERR_ARG;
    fv::Plan plan;
    int rc = fv::build_plan(T, n_split, FV_MODE_REFERENCE, nranks, plan);
    if (rc) return rc;
    std::vector<int> g(gathered, gathered + (size_t)T * nranks);
    merge_gathered(plan, g, T, nranks, path_out);
    return FV_OK;
}

extern "C" int fv_comm_unique_id(void *id_out)
{
    if (!id_out) return FV_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) <= FV_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return FV_ERR_COMM;
    std::memset(id_out, 0, FV_UNIQUE_ID_BYTES);
    std::memcpy(id_out, &id, sizeof id);
    return FV_OK;
}

extern "C" int fv_set_partition(fv_ctx *ctx, int rank, int nranks)
{
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) return FV_ERR_ARG;
    if (ctx->comm) return FV_ERR_STATE;
    ctx->rank = rank; ctx->nranks = nranks;
    return FV_OK;
}

extern "C" int fv_comm_init(fv_ctx *ctx, int rank, int nranks, const void *id)
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return FV_ERR_ARG;
    if (ctx->comm) return FV_ERR_STATE;
    FV_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t nr = ncclCommInitRank(&ctx->comm, nranks, uid, rank);
    if (nr != ncclSuccess) { ctx->detail = std::string("ncclCommInitRank: ") + ncclGetErrorString(nr); ctx->comm = nullptr; return FV_ERR_COMM; }
    ctx->rank = rank; ctx->nranks = nranks;
    return FV_OK;
}
