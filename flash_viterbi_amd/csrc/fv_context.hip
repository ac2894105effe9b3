// fv_context.hip — context life cycle, model upload, options, workspace and the decode epilogue of libflashvit.so.
// No kernels of its own: the full-state ones live in fv_full.hip, the FLASH-BS ones in fv_beam.hip.
#include "fv_internal.h"

namespace {

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

}  // namespace

namespace fvi {

size_t device_bytes(const fv_ctx *c)
{
    return c->LA32.bytes() + c->LA16.bytes() + c->LAQ16.bytes() + c->SPdata.bytes() + c->SPoff.bytes() + c->SPnwb.bytes() + c->LB32T.bytes() + c->LA64.bytes() + c->LB64T.bytes() + c->LPi64.bytes() +
           c->d_ob.bytes() + c->d_ans.bytes() + c->d_bp.bytes() + c->d_gather.bytes() + c->d_rows.bytes() + c->d_ckpt.bytes() +
           c->d_score.bytes() + c->d_counters.bytes() + c->d_hval.bytes() + c->d_scores.bytes() +
           c->d_hstate.bytes() + c->d_flags.bytes() + c->d_slot_val.bytes() + c->d_slot_state.bytes() +
           c->LA64R.bytes() + c->LAQ16R.bytes() + c->d_qaux.bytes() + c->d_tie_list.bytes() + c->d_tie_count.bytes() + c->d_cut.bytes() + c->d_dupwin.bytes() + c->d_cand.bytes() + c->d_cand_count.bytes() + c->d_passL.bytes() + c->d_needfull.bytes() + c->d_doubt.bytes() + c->d_doubt_count.bytes() + c->d_pack.bytes();
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
    if (ctx->comm || ctx->group) FV_HIP(ctx->d_gather.ensure((size_t)T * ctx->nranks));
    {
        const size_t want = 2 * FV_NCOUNTERS + 4 + (size_t)T * std::max(1, ctx->nranks) + (size_t)T;     // result block + the staged observations
        FV_HIP(ctx->d_pack.ensure(want));
        if (want > ctx->h_pin_n) {
            if (ctx->h_pin) { (void)hipHostFree(ctx->h_pin); ctx->h_pin = nullptr; ctx->h_pin_n = 0; }
            FV_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_pin), want * sizeof(int), hipHostMallocDefault));
            ctx->h_pin_n = want;
        }
    }
    return 0;
}

constexpr size_t PACK_HEAD = 2 * FV_NCOUNTERS + 4;        // ints in front of the answers: the counters (64-bit each), score + padding

__global__ void pack_result(const unsigned long long *counters, const float *score, const int *ans, size_t nans, int *out)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < 2 * FV_NCOUNTERS) out[tid] = reinterpret_cast<const int *>(counters)[tid];
    if (tid == 0) out[2 * FV_NCOUNTERS] = __float_as_int(*score);
    for (size_t i = tid; i < nans; i += (size_t)gridDim.x * blockDim.x) out[PACK_HEAD + i] = ans[i];
}

__global__ void clear_outputs(unsigned long long *counters, int *ans, int T)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < FV_NCOUNTERS) counters[tid] = 0ull;
    for (int i = tid; i < T; i += gridDim.x * blockDim.x) ans[i] = 0;
}

int begin_decode(fv_ctx *ctx, const int *ob, int T)
{
    ctx->h_ob.assign(ob, ob + T);
    // (the pinned block's tail is free until the epilogue: the sequence travels through it, one asynchronous copy)
    int *stage = ctx->h_pin + (ctx->h_pin_n - (size_t)T);
    std::memcpy(stage, ob, (size_t)T * sizeof(int));
    FV_HIP(hipMemcpyAsync(ctx->d_ob.p, stage, (size_t)T * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(clear_outputs, dim3(16), dim3(256), 0, ctx->stream, ctx->d_counters.p, ctx->d_ans.p, T);
    FV_HIP(hipGetLastError());
    return 0;
}

int finish_decode(fv_ctx *ctx, const fv::Plan &plan, int T, int *path_out, float *score_out, clk::time_point t0,
                  size_t nprof, bool beam)
{
    std::vector<int> host;
    const bool gathered = (ctx->comm || ctx->group) && !plan.seg_L.empty();
    if (gathered) {
        int rc = gather_answers(ctx, T);                  // one RCCL all-gather of every rank's answer array (fv_comm.hip)
        if (rc) return rc;
        host.resize((size_t)T * ctx->nranks);
    }
    FV_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
    // [counters | score | answers] in one block, one device-to-host copy into pinned memory
    const size_t nans = gathered ? host.size() : (size_t)T, total = PACK_HEAD + nans;
    hipLaunchKernelGGL(pack_result, dim3(64), dim3(256), 0, ctx->stream, ctx->d_counters.p, ctx->d_score.p,
                       gathered ? ctx->d_gather.p : ctx->d_ans.p, nans, ctx->d_pack.p);
    FV_HIP(hipGetLastError());
    FV_HIP(hipMemcpyAsync(ctx->h_pin, ctx->d_pack.p, total * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FV_HIP(hipStreamSynchronize(ctx->stream));
    for (hipGraphExec_t ge : ctx->graphs) (void)hipGraphExecDestroy(ge);
    ctx->graphs.clear();
    unsigned long long counters[FV_NCOUNTERS];
    float score;
    std::memcpy(counters, ctx->h_pin, sizeof counters);
    std::memcpy(&score, ctx->h_pin + 2 * FV_NCOUNTERS, sizeof score);
    if (gathered) {
        std::memcpy(host.data(), ctx->h_pin + PACK_HEAD, nans * sizeof(int));
        merge_gathered(plan, host, T, ctx->nranks, path_out);
    } else {
        std::memcpy(path_out, ctx->h_pin + PACK_HEAD, nans * sizeof(int));
    }
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
    st.beam_spec_steps = (long long)counters[9];
    st.beam_reach_events = (long long)counters[10];
    st.beam_list_short = (long long)counters[11];
    st.beam_list_long = (long long)counters[12];
    st.beam_list_entries = (long long)counters[13];
    st.beam_chain_cuts = (long long)counters[14];
    if (counters[5]) { ctx->detail = "heap replay: producer/consumer hand-shake timed out"; return FV_ERR_DEVICE; }
    st.device_bytes = (long long)fvi::device_bytes(ctx);
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

}  // namespace fvi


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
    if ((rc = fvi::full_setup(ctx)) || (rc = fvi::beam_setup(ctx))) return fail(rc);
    *out = ctx;
    return FV_OK;
}

extern "C" void fv_destroy(fv_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->group) {
        // the handle of a multi-device context (member 0) takes the other members and the group with it
        fv_group *g = ctx->group;
        if (ctx->group_rank != 0) return;        // members are owned by the group's handle
        for (fv_ctx *m : g->members) m->group = nullptr;
        for (size_t r = g->members.size(); r-- > 1;) fv_destroy(g->members[r]);
        for (hipEvent_t e : g->ans_ready) if (e) (void)hipEventDestroy(e);
        delete g;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) ncclCommDestroy(ctx->comm);
    ctx->LA32.release(); ctx->LA16.release(); ctx->LAQ16.release(); ctx->SPdata.release(); ctx->SPoff.release(); ctx->SPnwb.release(); ctx->LB32T.release(); ctx->LA64.release(); ctx->LB64T.release(); ctx->LPi64.release();
    ctx->d_ob.release(); ctx->d_ans.release(); ctx->d_bp.release(); ctx->d_gather.release(); ctx->d_rows.release(); ctx->d_ckpt.release();
    ctx->d_score.release(); ctx->d_counters.release(); ctx->d_hval.release(); ctx->d_scores.release();
    ctx->d_hstate.release(); ctx->d_flags.release(); ctx->d_slot_val.release(); ctx->d_slot_state.release();
    ctx->LA64R.release(); ctx->LAQ16R.release(); ctx->d_qaux.release(); ctx->d_tie_list.release(); ctx->d_tie_count.release(); ctx->d_cut.release(); ctx->d_dupwin.release(); ctx->d_cand.release(); ctx->d_cand_count.release(); ctx->d_passL.release(); ctx->d_needfull.release(); ctx->d_doubt.release(); ctx->d_doubt_count.release(); ctx->d_pack.release();
    if (ctx->h_pin) { (void)hipHostFree(ctx->h_pin); ctx->h_pin = nullptr; ctx->h_pin_n = 0; }
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

namespace fvi {

// Host side of fv_set_model: log() of every entry in double with the host libm (the calls the reference makes per
// cell, FLASH:142,150,167,170) and the table encodings.  Built once per model, uploaded to every device of a context.
int build_host_tables(const float *A, const float *B, const float *Pi, int K, int M, HostTables &h, std::string &detail)
{
    h.K = K; h.M = M;
    const int nrows = h.nrows = round_up(K, fvk::ROW_ALIGN);
    const int ntiles = h.ntiles = (K + fvk::TILE_W - 1) / fvk::TILE_W;
    // the full-state step kernels keep one score row in LDS: beyond that K only the beam path is available
    // (it needs the float64 table alone, which also keeps the host footprint at 8 B per entry)
    const bool full_ok = h.full_ok = fvk::step_lds_bytes<1>(nrows) <= 160 * 1024;
    h.u16_ok = fvk::u16_lds_bytes<1, 8>(nrows) <= 160 * 1024 && K <= 65536;

    const size_t tab = h.tab = (size_t)ntiles * nrows * fvk::TILE_W;
    std::vector<double> &h64 = h.h64;
    std::vector<float> &h32 = h.h32;
    std::vector<unsigned short> &h16 = h.h16;
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
    h.b64.assign((size_t)M * K, 0.0); h.pi64.assign(K, 0.0);
    h.b32.assign((size_t)M * K, 0.0f);
    for (int i = 0; i < K; ++i) {
        for (int o = 0; o < M; ++o) {
            const float x = B[(size_t)i * M + o];
            if (!(x >= 0.0f) || std::isinf(x)) ok_range = false;
            if (x > 1.0f) big[0] = 1;
            const double l = std::log((double)x);
            h.b64[(size_t)o * K + i] = l; h.b32[(size_t)o * K + i] = (float)l;
        }
        const float x = Pi[i];
        if (!(x >= 0.0f) || std::isinf(x)) ok_range = false;
        if (x > 1.0f) big[0] = 1;
        h.pi64[i] = std::log((double)x);
    }
    h.any_big = false;
    for (int k = 0; k < K; ++k) { if (bad[k]) ok_range = false; if (big[k]) h.any_big = true; }
    if (!ok_range) { detail = "model entries must be finite and >= 0"; return FV_ERR_ARG; }

    double dmax = 0.0;
    for (int k = 0; k < K; ++k) dmax = std::max(dmax, dmax_row[k]);
    h.window16 = std::nextafter((float)(2.0 * dmax), HUGE_VALF);     // rounded up
    // fixed-point table: step = (largest finite |log A|) / 65534, code = round(-L/step), 0xffff = -inf.
    // The kernel evaluates fma(code, -step, s) with step as a float, so the error is measured against
    // exactly that product (code * (double)(float)step is exact in double).
    h.density = 1.0; h.windowq = 0.0f; h.qscale = -1.0f;
    if (!full_ok) return FV_OK;
    std::vector<unsigned short> &hq = h.hq;
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
    h.windowq = std::nextafter((float)(2.0 * dqmax), HUGE_VALF);
    h.qscale = -stepf;
    // sparse form of the same codes (CSC-Q16, see trellis_step_sparse): per tile, per column, the finite
    // entries in ascending k as (k << 16 | code), 4 per lane load, columns padded to the tile's longest
    if (K <= 65536) {
        std::vector<int> &off = h.sp_off, &nwbv = h.sp_nwb;
        off.assign(ntiles, 0); nwbv.assign(ntiles, 0);
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
        h.density = (double)nnz / ((double)K * K);
        if (total < (size_t)1 << 30 && total > 0) {
            std::vector<uint4> &sp = h.sp;
            try { sp.resize(total); } catch (...) { return FV_ERR_NOMEM; }
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
        }
    }
    return FV_OK;
}

// Device side of fv_set_model, once per device.  Tables are released and overwritten in place: until every upload has
// succeeded the context holds NO model (K = 0 makes every decode return FV_ERR_STATE), so a failure half way (e.g.
// NOMEM on a larger second model) can never pair the old sizes with partly new tables.
int upload_tables(fv_ctx *ctx, const HostTables &h)
{
    FV_HIP(hipSetDevice(ctx->device));
    ctx->K = 0; ctx->M = 0; ctx->nrows = 0; ctx->full_ok = false; ctx->u16_ok = false; ctx->laq16_ready = false;
    ctx->rowq_ready = false; ctx->beam_q16_ready = false;
    (void)hipStreamSynchronize(ctx->stream);
    ctx->LA64R.release(); ctx->LAQ16R.release();
    ctx->SPdata.release(); ctx->SPoff.release(); ctx->SPnwb.release();
    ctx->window16 = h.window16; ctx->windowq = h.windowq; ctx->qscale = h.qscale; ctx->density = h.density;
    const size_t tab = h.tab;
    if (h.full_ok) {
        FV_HIP(ctx->LAQ16.ensure(tab));
        FV_HIP(hipMemcpy(ctx->LAQ16.p, h.hq.data(), tab * sizeof(unsigned short), hipMemcpyHostToDevice));
        if (!h.sp.empty()) {
            FV_HIP(ctx->SPdata.ensure(h.sp.size()));
            FV_HIP(ctx->SPoff.ensure(h.ntiles));
            FV_HIP(ctx->SPnwb.ensure(h.ntiles));
            FV_HIP(hipMemcpy(ctx->SPdata.p, h.sp.data(), h.sp.size() * sizeof(uint4), hipMemcpyHostToDevice));
            FV_HIP(hipMemcpy(ctx->SPoff.p, h.sp_off.data(), h.ntiles * sizeof(int), hipMemcpyHostToDevice));
            FV_HIP(hipMemcpy(ctx->SPnwb.p, h.sp_nwb.data(), h.ntiles * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    FV_HIP(ctx->LA64.ensure(tab));
    if (h.full_ok) {
        FV_HIP(ctx->LA32.ensure(tab));
        FV_HIP(ctx->LA16.ensure(tab));
        FV_HIP(hipMemcpy(ctx->LA16.p, h.h16.data(), tab * sizeof(unsigned short), hipMemcpyHostToDevice));
        FV_HIP(hipMemcpy(ctx->LA32.p, h.h32.data(), tab * sizeof(float), hipMemcpyHostToDevice));
    } else {
        ctx->LA32.release(); ctx->LA16.release(); ctx->LAQ16.release();
    }
    FV_HIP(ctx->LB64T.ensure((size_t)h.M * h.K));
    FV_HIP(ctx->LB32T.ensure((size_t)h.M * h.K));
    FV_HIP(ctx->LPi64.ensure(h.K));
    FV_HIP(hipMemcpy(ctx->LA64.p, h.h64.data(), tab * sizeof(double), hipMemcpyHostToDevice));
    FV_HIP(hipMemcpy(ctx->LB64T.p, h.b64.data(), h.b64.size() * sizeof(double), hipMemcpyHostToDevice));
    FV_HIP(hipMemcpy(ctx->LB32T.p, h.b32.data(), h.b32.size() * sizeof(float), hipMemcpyHostToDevice));
    FV_HIP(hipMemcpy(ctx->LPi64.p, h.pi64.data(), h.pi64.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->laq16_ready = h.full_ok;          // beyond the float32 limit: built on first use (fv_full.hip)
    ctx->K = h.K; ctx->M = h.M; ctx->nrows = h.nrows; ctx->full_ok = h.full_ok; ctx->u16_ok = h.u16_ok;     // LA64R / LAQ16R: rebuilt on the next beam decode
    ctx->logs_nonpositive = !h.any_big;
    ctx->stats = fv_stats{};
    ctx->stats.device_bytes = (long long)device_bytes(ctx);
    return FV_OK;
}

}  // namespace fvi

extern "C" int fv_set_model(fv_ctx *ctx, const float *A, const float *B, const float *Pi, int K, int M)
{
    if (!ctx || !A || !B || !Pi || K < 1 || M < 1) return FV_ERR_ARG;
    auto t0 = clk::now();
    fvi::HostTables h;
    int rc = fvi::build_host_tables(A, B, Pi, K, M, h, ctx->detail);
    if (rc) return rc;
    // a multi-device context uploads the same host tables to every device
    const int n = fvi::group_size(ctx);
    for (int r = 0; r < n; ++r) {
        fv_ctx *m = fvi::group_member(ctx, r);
        if ((rc = fvi::upload_tables(m, h))) { if (m != ctx) ctx->detail = m->detail; return rc; }
    }
    ctx->stats.set_model_ms = ms_since(t0);
    return FV_OK;
}

extern "C" int fv_set_option(fv_ctx *ctx, int key, long long value)
{
    if (!ctx) return FV_ERR_ARG;
    if (ctx->group && ctx->group_rank == 0) {          // a multi-device context: the same option on every member
        for (int r = fvi::group_size(ctx) - 1; r >= 1; --r) {
            const int rc = fv_set_option(fvi::group_member(ctx, r), key, value);
            if (rc) return rc;
        }
    }
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
#ifndef FV_TIMING_BUILD
        // bits 0, 4, 5, 11, 12 leave a part of a kernel out (to time the rest) and so change results: they exist in
        // the timing build only (libflashvit_timing.so, tools/).  Every bit this library accepts is speed-only.
        if (value & FV_DEBUG_TIMING_ONLY) { ctx->detail = "FV_OPT_DEBUG: result-changing timing switches need the timing build"; return FV_ERR_ARG; }
#endif
        if (value < 0 || value >= (1ll << 27)) return FV_ERR_ARG;
        ctx->opt_debug = (int)value; return FV_OK;
    default: return FV_ERR_ARG;
    }
}

extern "C" long long fv_checkpoint_memory_bytes(int K, int T, int step)
{
    if (step <= 0) step = (int)std::floor(std::sqrt(1.0 * T));
    const long long nck = (T + step - 1) / step;
    const long long last = T - (nck - 1) * step;
    const long long tsub = nck > 1 && step + 1 > last ? step + 1 : last;       // T_sub, checkpoint Viterbi.c:123
    return 4LL * K + 4LL * K * nck + 4LL * K + 4LL * (T / step + 1) + 8LL * K * tsub;   // :250
}

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
