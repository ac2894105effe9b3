// fv_beam.hip — the FLASH-BS (dynamic beam) path of libflashvit.so: fv_decode_beam and its generation driver.
#include "fv_internal.h"
#include "fv_device_common.h"
#include "fv_beam_kernels.hip.inc"

namespace {
int decode_beam_impl(fv_ctx *ctx, const int *ob, int T, int n_split, int beam_width, int mode, int *path_out, float *score_out);

// row pitch of the row-major float64 table the beam kernels gather from
inline int beam_ld(int K) { return (K + fvb::BEAM_COLS - 1) / fvb::BEAM_COLS * fvb::BEAM_COLS; }
inline int beam_ldq(int K) { static_assert(fvb::BEAMQ_COLS == 128, "fv_ldq"); return fv_ldq(K); }   // ... of the 16-bit one

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
    const int BP = fvb::beam_pitch(beam);
    FV_HIP(hipMemsetAsync(ctx->d_doubt_count.p, 0, (size_t)T * sizeof(int), ctx->stream));      // doubtful-column lists of this generation
    fvb::ResolveCtx rcx;
    rcx.counters = ctx->d_counters.p; rcx.K = K; rcx.beam = beam;
    rcx.no_cut = (ctx->opt_debug & (1 << 23)) ? 1 : 0;
    rcx.LA64R = ctx->LA64R.p; rcx.ld = beam_ld(K); rcx.LB32T = ctx->LB32T.p; rcx.ob = ctx->d_ob.p;
    rcx.bp = ctx->d_bp.p; rcx.doubt = ctx->d_doubt.p; rcx.doubt_count = ctx->d_doubt_count.p;
    rcx.b.scores_all = ctx->d_scores.p; rcx.b.hval = ctx->d_hval.p; rcx.b.hstate = ctx->d_hstate.p;
    rcx.b.cut = ctx->d_cut.p; rcx.b.passL = ctx->d_passL.p + pass_off;
    // members of the heaps of passes [first, first + count) at lock-step s: one launch
    auto select = [&](int first, int count, int s, hipStream_t st) -> int {
        fvb::SelArgs a;
        a.counters = ctx->d_counters.p; a.K = K; a.beam = beam; a.s = s;
        a.no_wave = (ctx->opt_debug & 32768) ? 1 : 0;
        a.eager = (ctx->opt_debug & 1048576) ? 1 : 0;             // FV_OPT_DEBUG bit 20: replay every duplicate step at once
        a.quad_dirty = (ctx->opt_debug & 16777216) ? 1 : 0;
        a.sb_rounds = (ctx->opt_debug & (1 << 22)) ? 2 : fvb::SEL_MAX_ROUNDS;
        a.T = T; a.own_pred = (ctx->opt_debug & (1 << 7)) ? 1 : 0;
        a.margin = ctx->opt_sel_margin; a.cand_cap = cand_cap;
        a.cand = ctx->d_cand.p; a.cand_count = ctx->d_cand_count.p; a.rc = rcx;
        a.rc.b.passL = rcx.b.passL + first;
        const bool listed = count <= fvb::BEAM_CHUNK;
        for (int q = 0; listed && q < count; ++q) {
            const int j = passes[first + q].L + s;
            a.p[q] = fvb::SelJob{ ctx->d_scores.p + (size_t)j * K, ctx->d_hval.p + (size_t)j * BP, ctx->d_hstate.p + (size_t)j * BP,
                                  ctx->d_cut.p + (size_t)j * fvb::CUT_W, s >= 1 ? ctx->d_cut.p + (size_t)(j - 1) * fvb::CUT_W : nullptr,
                                  (cand_cap && s >= 1) ? ctx->d_cand.p + (size_t)j * cand_cap : nullptr, ctx->d_cand_count.p + j,
                                  j, passes[first + q].L };
        }
        // steps >= 2 of a pass have a candidate list (the predictor needs two cut values)
        // K > 65536 (FV_OPT_DEBUG bit 22: any K): beyond the 64 rounds the register kernels hold, every selection runs in the
        // lean kernel — on the candidate list where there is one, otherwise over the K scores in memory
        const bool many = K > fvb::SEL_MAX_ROUNDS * fvb::SEL_BLOCK || (ctx->opt_debug & (1 << 22));
        fvb::SelKernel lean = (s >= 2 || many) ? fvb::sel_cand_kernel_for(K, cand_cap, listed, many) : nullptr;
        hipLaunchKernelGGL(lean ? lean : fvb::sel_kernel_for(K, listed), dim3(count), dim3(fvb::SEL_BLOCK), fvb::sel_lds(beam), st, a);
        FV_HIP(hipGetLastError());
        return 0;
    };
    // init scores (the same rows the full variant starts from, FLASH_BS:407-427)
    int rc0 = 0;
    for (int base = 0; base < np; base += fvk::PASS_CHUNK) {
        fvk::PassChunk ch;
        ch.n = std::min(fvk::PASS_CHUNK, np - base);
        for (int q = 0; q < ch.n; ++q) {
            const fv::Pass &p = passes[base + q];
            ch.p[q] = fvk::PassDesc{ p.L, p.R, p.from_pi ? 1 : 0, p.whole ? 1 : 0, (long long)p.L * K };
        }
        if ((rc0 = fvi::launch_init_rows(ctx, ch, ctx->d_scores.p))) return rc0;
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
                a.LAQ16R = ctx->LAQ16R.p; a.qpar = ctx->beam_q16_ready ? reinterpret_cast<const float *>(ctx->d_qaux.p + 2) : nullptr;
                a.cand = ctx->d_cand.p; a.cand_count = ctx->d_cand_count.p; a.cand_cap = cand_cap;
                a.n = std::min(fvb::BEAM_CHUNK, active - base);
                for (int q = 0; q < a.n; ++q) {
                    const int j = passes[first + base + q].L + s;
                    a.p[q].sval = ctx->d_hval.p + (size_t)(j - 1) * BP;
                    a.p[q].sstate = ctx->d_hstate.p + (size_t)(j - 1) * BP;
                    a.p[q].doubt = ctx->d_doubt.p + (size_t)j * fvb::DOUBT_CAP;
                    a.p[q].doubt_count = ctx->d_doubt_count.p + j;
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
                const bool use_q16 = ctx->beam_q16_ready && !(ctx->opt_debug & 256) &&
                                     ((ctx->opt_debug & 512) || (double)a.n * beam * K * 8.0 >= 80e6);
                // 8-wave workgroups once the launch has more 16-wave workgroups than fit the chip together (two per CU);
                // FV_OPT_DEBUG bit 25: never, bit 26: always
                const int panels = beam_ldq(K) / fvb::BEAMQ_COLS;
                const bool narrow = !(ctx->opt_debug & (1 << 25)) &&
                                    ((ctx->opt_debug & (1 << 26)) || (long long)panels * a.n > 2LL * ctx->num_cus);
                // (4-wave workgroups for launches beyond four 8-wave workgroups per CU: cfg4 right-hand 3.51 -> 3.56 ms, cfg5 95.8 -> 97.1)
                if (use_q16 && narrow)
                    hipLaunchKernelGGL(fvb::beam_step_q16<8>, dim3(panels, a.n), dim3(8 * 64), fvb::beam_step_q16_lds(beam, 8), st, a);
                else if (use_q16)
                    hipLaunchKernelGGL(fvb::beam_step_q16<16>, dim3(panels, a.n), dim3(fvb::BEAM_BLOCK),
                                       fvb::beam_step_q16_lds(beam), st, a);
                // (small beams: four waves per workgroup — K = 3965, B = 32: 4.55 -> 4.23 ms; eight waves at B = 256: 6.43 -> 6.50, not kept)
                else if (beam <= 64 && !(ctx->opt_debug & (1 << 25)))
                    hipLaunchKernelGGL(fvb::beam_step<4>, dim3(beam_ld(K) / fvb::BEAM_COLS, a.n), dim3(4 * 64), fvb::beam_step_lds(beam, 4), st, a);
                else
                    hipLaunchKernelGGL(fvb::beam_step<16>, dim3(beam_ld(K) / fvb::BEAM_COLS, a.n), dim3(fvb::BEAM_BLOCK), fvb::beam_step_lds(beam),
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
                               ctx->d_slot_state.p, ctx->d_hstate.p, ctx->d_cut.p, ctx->d_bp.p, ctx->d_ans.p, ctx->d_score.p);
            FV_HIP(hipGetLastError());
        }
        return 0;
    };
    // Undecided duplicate steps (lazy replays, fv_beam_kernels.hip.inc): mode 0 decides what the pass ends themselves read,
    // mode 1 — behind the same gate as the layouts — everything, because heap_build_all replays every step's exact scores.
    auto resolve = [&](int mode, const int *gate) -> int {
        for (int base = 0; base < np; base += fvb::BEAM_CHUNK) {
            fvb::ResolveArgs r;
            r.rc = rcx; r.mode = mode; r.gate = gate; r.ans = ctx->d_ans.p;
            r.n = std::min(fvb::BEAM_CHUNK, np - base);
            for (int q = 0; q < r.n; ++q) r.p[q] = fvb::BeamEnd{ passes[base + q].L, passes[base + q].R, passes[base + q].whole ? 1 : 0 };
            hipLaunchKernelGGL(fvb::beam_resolve, dim3(r.n), dim3(fvb::RESOLVE_BLOCK), fvb::heap_lds(beam), ctx->stream, r);
            FV_HIP(hipGetLastError());
        }
        return 0;
    };
    if ((rc = resolve(0, nullptr))) return rc;
    if (lazy) {
        if ((rc = layouts(true, nullptr))) return rc;
        if ((rc = ends(1))) return rc;
    } else {
        int one = 1;
        FV_HIP(hipMemcpyAsync(ctx->d_needfull.p, &one, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        FV_HIP(hipStreamSynchronize(ctx->stream));       // (`one` is a local; this is the experiment path)
    }
    if ((rc = resolve(1, ctx->d_needfull.p))) return rc;
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
    if (ctx->group && ctx->group_rank == 0) {
        if (!path_out || T < 2) return FV_ERR_ARG;
        return fvi::group_run(ctx, T, path_out, score_out, [&](fv_ctx *m, int *path, float *score) {
            return fvi::drained(m, decode_beam_impl(m, ob, T, n_split, beam_width, mode, path, score));
        });
    }
    return fvi::drained(ctx, decode_beam_impl(ctx, ob, T, n_split, beam_width, mode, path_out, score_out));
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
    if ((rc = fvi::ensure_workspace(ctx, T, 1))) return rc;
    (void)most;
    FV_HIP(ctx->d_scores.ensure((size_t)T * ctx->K));
    FV_HIP(ctx->d_hval.ensure((size_t)T * fvb::beam_pitch(beam_width)));
    FV_HIP(ctx->d_hstate.ensure((size_t)T * fvb::beam_pitch(beam_width)));
    FV_HIP(ctx->d_doubt.ensure((size_t)T * fvb::DOUBT_CAP));
    FV_HIP(ctx->d_doubt_count.ensure(T));
    FV_HIP(ctx->d_slot_val.ensure((size_t)T * beam_width));
    FV_HIP(ctx->d_slot_state.ensure((size_t)T * beam_width));
    FV_HIP(ctx->d_tie_list.ensure((size_t)T * ctx->K));
    FV_HIP(ctx->d_tie_count.ensure(4));
    FV_HIP(ctx->d_cut.ensure((size_t)T * fvb::CUT_W));
    FV_HIP(hipMemsetAsync(ctx->d_cut.p, 0xFF, (size_t)T * fvb::CUT_W * sizeof(float), ctx->stream));      // NaN: no earlier pass has left a cut here
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
    if (!ctx->beam_q16_ready && ctx->logs_nonpositive) {
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
        ctx->beam_q16_ready = true; ctx->rowq_ready = true;
    }

    const double keep_model_ms = ctx->stats.set_model_ms;
    ctx->stats = fv_stats{};
    ctx->stats.set_model_ms = keep_model_ms;
    ctx->stats.kernel = FV_KERNEL_F64_STREAM;
    ctx->stats.generations = plan.generations();
    ctx->stats.table_bytes_per_step = (long long)beam_width * ctx->K * 8;

    if ((rc = fvi::begin_decode(ctx, ob, T))) return rc;
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
    return fvi::finish_decode(ctx, plan, T, path_out, score_out, t0, 0, true);
}
}  // namespace

#ifdef FV_REPLAY_PROF
// Experiment builds only: read and reset the replay profile (fv_beam_kernels.hip.inc, replay_prof).
extern "C" int fv_debug_replay_prof(unsigned long long *out8)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(fvb::replay_prof), sizeof z) != hipSuccess) return FV_ERR_DEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fvb::replay_prof), z, sizeof z) != hipSuccess) return FV_ERR_DEVICE;
    return FV_OK;
}
extern "C" int fv_debug_reach_prof(unsigned long long *out8)
{
    unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(fvb::reach_prof), sizeof z) != hipSuccess) return FV_ERR_DEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(fvb::reach_prof), z, sizeof z) != hipSuccess) return FV_ERR_DEVICE;
    return FV_OK;
}
#endif

namespace fvi {
int beam_setup(fv_ctx *ctx) { return fvb::allow_big_lds(ctx->detail); }
}  // namespace fvi
