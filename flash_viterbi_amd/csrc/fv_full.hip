// fv_full.hip — the full-state path of libflashvit.so: step-kernel launchers, the lock-step generation driver and
// the decode entry points fv_decode_full / fv_decode_vanilla / fv_decode_checkpoint.
//
// Everything a decode needs is enqueued on one HIP stream without a single host
// round trip: the task tree depends only on (T, n_split), and every state index a
// later pass consumes (Ans[L-1], Ans[R]) stays in device memory.  One sync at the end.
#include "fv_internal.h"
#include "fv_kernels.hip.inc"

namespace {

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

// source rows per launch of trellis_step: all of them if NB score rows fit LDS next to the reduction scratch
template <int NB>
int slab_rows(const fv_ctx *ctx)
{
    int slab = ctx->nrows;
    if (ctx->opt_debug & (1 << 21)) slab = std::max(64, (ctx->nrows / 3 + 63) / 64 * 64);
    while (slab > 64 && fvk::step_lds_bytes<NB>(slab) > 160 * 1024) slab = (slab / 2 + 63) / 64 * 64;
    return slab;
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
    a.row_lo = 0; a.srows = ctx->nrows; a.merge = 0;
    for (int t = 0; t < NB; ++t) a.t[t] = slots[t < nb ? t : 0];
    const size_t lds = fvk::step_lds_bytes<NB>(ctx->nrows);
    constexpr int RBR = 4 * fvk::Tab<TA>::R;
    const int nj_max = (ctx->nrows / RBR + fvk::NWAVES - 1) / fvk::NWAVES;
    // One launch when the score rows fit LDS; otherwise (K > ~40100 at NB = 1: the route of every model the packed 16-bit
    // kernel cannot take — K > 65536, or entries above 1) one launch per slab of source rows, ascending, each merging
    // into what the slabs below it left in t1_out / bp_out.  FV_OPT_DEBUG bit 21 forces three slabs (tests).
    const int slab = slab_rows<NB>(ctx);
    if (slab < ctx->nrows) {
        constexpr int U = std::is_same<TA, double>::value ? U_DB64 : std::is_same<TA, float>::value ? U_DB32 : U_DB16;
        a.reverse = 0;
        for (int lo = 0; lo < ctx->nrows; lo += slab) {
            a.row_lo = lo; a.srows = std::min(slab, ctx->nrows - lo); a.merge = lo > 0 ? 1 : 0;
            const int rc = launch_variant<TA, NB, U, true>(ctx, a, fvk::step_lds_bytes<NB>(a.srows));
            if (rc) return rc;
        }
        return 0;
    }
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
    a.row_lo = 0; a.srows = ctx->nrows; a.merge = 0;
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
    // (the float64 kernel beyond one LDS row sweeps slabs of source rows: four tasks per launch keep a slab at ~10000 rows)
    const bool slabbed = (kernel == FV_KERNEL_F64_STREAM || kernel == FV_KERNEL_Q16_REFINE) && !ctx->full_ok;
    int cap = std::max(1, std::min(ctx->opt_max_batch, slabbed ? 4 : max_batch_for(ctx->nrows, !ctx->full_ok)));
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
    // (Round 3, measured and not kept: the co-resident batches in ONE launch, gridDim.y = three batches of four, instead
    // of three launches on three streams — no fork / join, a third of the host launches — cfg2 right-hand 1.64 -> 1.79 ms,
    // cfg3 43.8 -> 51.1 ms: workgroups of one launch run in phase, all in their prologue or all in their sweep; what the
    // streams provide is the stagger.)
    const bool two = ((kernel == FV_KERNEL_U16_REFINE && ctx->u16_ok) || (kernel == FV_KERNEL_SPARSE_Q16 && maxlen >= 64)) &&
                     !(ctx->opt_debug & 262144) && !whole_gen && np > FORK_CAP &&
                     !ctx->opt_profile && !(ctx->opt_debug & 64);
    const int nbatches = (np + FORK_CAP - 1) / FORK_CAP;
    const int nstreams = two ? std::min(FORK_STREAMS, nbatches) : 1;      // (batches of three tasks were slower: 2.16 / 54.9 ms)
    auto fork = [&]() -> int {
        FV_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int q = 1; q < nstreams; ++q) FV_HIP(hipStreamWaitEvent(ctx->aux[q - 1], ctx->ev_fork, 0));
        return 0;
    };
    if (two) {
        cap = std::min(cap, FORK_CAP);
        // no packets may wait on the other queues while a serial generation runs (decode_beam_impl has the measurement)
        // (polled: a blocking hipStreamSynchronize wakes the host ~30 us after the stream has drained, and the chip idles
        // until the first forked launch arrives)
        if (!ctx->fork_active) {
            hipError_t qe;
            while ((qe = hipStreamQuery(ctx->stream)) == hipErrorNotReady) std::this_thread::yield();
            FV_HIP(qe);
        }
        ctx->fork_active = true;
        { int rc = fork(); if (rc) return rc; }
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
        // Single-column last steps of the passes that finish at this lock-step.  A pass's row was written by the stream its
        // batch keeps for the whole generation (or, at lock-step 1, by init_rows on the main stream), so its last step goes
        // to that very stream: no cross-stream join in the middle of a generation (a join cost ~10 us of host time, and
        // the dependency between queues left the chip idle for 15-50 us at each of the 24 of a cfg2 decode).
        for (int sid = 0; sid < nstreams; ++sid) {
            fvk::ColArgs c;
            c.LA64 = ctx->LA64.p; c.ans = ctx->d_ans.p; c.K = K; c.nrows = ctx->nrows; c.n = 0;
            hipStream_t cst = sid ? ctx->aux[sid - 1] : ctx->stream;
            auto flush = [&]() -> int {
                if (c.n == 0) return 0;
                hipLaunchKernelGGL(fvk::last_column, dim3(c.n), dim3(256), 0, cst, c);
                FV_HIP(hipGetLastError());
                ctx->stats.column_steps += c.n;
                c.n = 0;
                return 0;
            };
            for (int q = full; q < active; ++q) {
                const int owner = (two && s > 1) ? (q / cap) % nstreams : 0;
                if (owner != sid) continue;
                const fv::Pass &p = passes[q];
                c.p[c.n++] = fvk::ColJob{ row(q, (s - 1) & 1), ctx->LB32T.p + (size_t)ctx->h_ob[p.R] * K, ctx->d_bp.p + (size_t)p.R * K, p.R };
                if (c.n == fvk::COL_CHUNK) { int rc = flush(); if (rc) return rc; }
            }
            int rc = flush();
            if (rc) return rc;
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

}  // namespace

namespace fvi {

// every full-state step kernel may use the whole 160 KB of LDS
int full_setup(fv_ctx *ctx)
{
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
        return rc;
    if ((rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, 8, false, 16>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, U_DB16, true, 16>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<1, U_DB16, true, 8>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<2, U_DB16, true, 8>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<4, U_DB16, true, 8>)) ||
        (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, 16, false, 8>)) || (rc = set_big_lds(ctx, &fvk::trellis_step_u16<8, U_DB16, true, 8>)))
        return rc;
    return 0;
}

int launch_init_rows(fv_ctx *ctx, const fvk::PassChunk &ch, float *rows)
{
    hipLaunchKernelGGL(fvk::init_rows, dim3((ctx->K + 255) / 256, ch.n), dim3(256), 0, ctx->stream, ch,
                       ctx->LA64.p, ctx->nrows, ctx->LB64T.p, ctx->LPi64.p, ctx->d_ob.p, ctx->d_ans.p, rows, ctx->K);
    FV_HIP(hipGetLastError());
    return 0;
}

}  // namespace fvi

namespace {
int decode_full_impl(fv_ctx *ctx, const int *ob, int T, int n_split, int mode, int *path_out, float *score_out);
int decode_checkpoint_impl(fv_ctx *ctx, const int *ob, int T, int step, int *path_out, float *score_out);
}  // namespace

extern "C" int fv_decode_full(fv_ctx *ctx, const int *ob, int T, int n_split, int mode, int *path_out, float *score_out)
{
    if (!ctx) return FV_ERR_ARG;
    if (ctx->group && ctx->group_rank == 0 && !ctx->vanilla) {
        // multi-device context: every member decodes its share on its own device and host thread, one gather merges them
        if (!path_out || T < 2) return FV_ERR_ARG;
        return fvi::group_run(ctx, T, path_out, score_out, [&](fv_ctx *m, int *path, float *score) {
            return fvi::drained(m, decode_full_impl(m, ob, T, n_split, mode, path, score));
        });
    }
    return fvi::drained(ctx, decode_full_impl(ctx, ob, T, n_split, mode, path_out, score_out));
}

namespace {
int decode_full_impl(fv_ctx *ctx, const int *ob, int T, int n_split, int mode, int *path_out, float *score_out)
{
    if (!ctx || !ob || !path_out || T < 2 || n_split < 1) return FV_ERR_ARG;
    if (ctx->K == 0) return FV_ERR_STATE;
    // Beyond the float32 kernels' LDS limit (one score row of K floats) two routes remain:
    //   * the packed 16-bit kernel — a row of 16-bit score codes is half the bytes (K <= 65536); it needs every model entry
    //     in [0,1], and its table is built on the device on first use;
    //   * the float64 kernel in slabs of source rows (launch_step_nb): any K the float64 table fits the device for, any
    //     model; 8 B per cell instead of 2.
    const bool wide = !ctx->full_ok;
    const bool big = wide && ctx->u16_ok && ctx->logs_nonpositive && !ctx->vanilla &&
                     (ctx->opt_kernel == FV_KERNEL_AUTO || ctx->opt_kernel == FV_KERNEL_U16_REFINE);
    // ... and the same slabs for the f32 filter on the 16-bit table (2 B per cell; model entries in [0,1]): what AUTO takes
    // beyond K = 65536
    const bool wide_q16 = wide && !big && ctx->logs_nonpositive && !ctx->vanilla &&
                          (ctx->opt_kernel == FV_KERNEL_AUTO || ctx->opt_kernel == FV_KERNEL_Q16_REFINE);
    if (wide && !big && !wide_q16 && !(ctx->opt_kernel == FV_KERNEL_AUTO || ctx->opt_kernel == FV_KERNEL_F64_STREAM)) {
        ctx->detail = "full-state decode of K > ~40100: FV_KERNEL_AUTO, FV_KERNEL_U16_REFINE (K <= 65536), FV_KERNEL_Q16_REFINE (both: model entries in [0,1]) or FV_KERNEL_F64_STREAM";
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
    const int kernel = big ? FV_KERNEL_U16_REFINE : wide_q16 ? FV_KERNEL_Q16_REFINE : wide ? FV_KERNEL_F64_STREAM : pick_kernel(ctx);
    if ((big || wide_q16) && !ctx->laq16_ready) {      // (the flag, not the pointer: a build that failed half way leaves the buffer allocated)
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
        ctx->laq16_ready = true;
    }

    // generations of passes this rank runs
    std::vector<std::vector<fv::Pass>> gens(plan.generations());
    size_t most = 1;
    for (const fv::Pass &p : plan.passes)
        if (p.owner < 0 || p.owner % ctx->nranks == ctx->rank) gens[p.generation].push_back(p);
    for (auto &g : gens) most = std::max(most, g.size());
    if ((rc = fvi::ensure_workspace(ctx, T, most))) return rc;

    const double keep_model_ms = ctx->stats.set_model_ms;
    ctx->stats = fv_stats{};
    ctx->stats.set_model_ms = keep_model_ms;
    ctx->stats.kernel = kernel;
    ctx->stats.generations = plan.generations();
    ctx->stats.table_bytes_per_step = (long long)((ctx->K + fvk::TILE_W - 1) / fvk::TILE_W) * ctx->nrows * fvk::TILE_W * (kernel == FV_KERNEL_F64_STREAM ? 8 : kernel == FV_KERNEL_F32_REFINE ? 4 : 2);
    if (kernel == FV_KERNEL_SPARSE_Q16) ctx->stats.table_bytes_per_step = (long long)ctx->SPdata.bytes();
    ctx->stats.density = ctx->density;

    if ((rc = fvi::begin_decode(ctx, ob, T))) return rc;
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
    return fvi::finish_decode(ctx, plan, T, path_out, score_out, t0, nprof, false);
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
    return fvi::drained(ctx, decode_checkpoint_impl(ctx, ob, T, step, path_out, score_out));
}

namespace {
int decode_checkpoint_impl(fv_ctx *ctx, const int *ob, int T, int step, int *path_out, float *score_out)
{
    if (!ctx || !ob || !path_out || T < 2) return FV_ERR_ARG;
    if (ctx->K == 0) return FV_ERR_STATE;
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= ctx->M) return FV_ERR_ARG;
    if (step <= 0) step = (int)std::floor(std::sqrt(1.0 * T));        // checkpoint Viterbi.c:179-180
    auto t0 = clk::now();
    FV_HIP(hipSetDevice(ctx->device));
    const int K = ctx->K, nrows = ctx->nrows;
    const int nck = (T + step - 1) / step;
    fv::Plan plan;
    int rc = fv::build_plan(T, 1, FV_MODE_SINGLE_PASS, 1, plan);
    if (rc) return rc;
    if ((rc = fvi::ensure_workspace(ctx, T, (size_t)nck + 1))) return rc;  // two rows per segment + two for the first pass
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
    if ((rc = fvi::begin_decode(ctx, ob, T))) return rc;
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
    const int cap = std::max(1, std::min(ctx->opt_max_batch, ctx->full_ok ? max_batch_for(nrows, false) : 4));
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
    return fvi::finish_decode(ctx, plan, T, path_out, score_out, t0, 0, false);
}
}  // namespace
