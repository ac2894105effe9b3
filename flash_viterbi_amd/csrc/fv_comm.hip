// fv_comm.hip — multi-GPU side of libflashvit.so: partition of the top-level segments over ranks, the one RCCL
// all-gather at the merge step, the merge itself.
#include "fv_internal.h"

namespace fvi {

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

// The one exchange of a multi-rank decode: every rank's answer array (T int32) to every rank, on ctx->stream.
//  * one process per GPU (fv_comm_init) and multi-device contexts on distinct devices: ncclAllGather over xGMI;
//  * a multi-device context whose device list repeats an id (a 1-GPU lease: two RCCL ranks cannot share a device):
//    the same rendezvous with device-to-device copies in place of RCCL — every member waits for every member's
//    "answers final" event and copies that member's array into its own gather buffer.
int gather_answers(fv_ctx *ctx, int T)
{
    if (ctx->comm) {
        ncclResult_t nr = ncclAllGather(ctx->d_ans.p, ctx->d_gather.p, (size_t)T, ncclInt32, ctx->comm, ctx->stream);
        if (nr != ncclSuccess) { ctx->detail = std::string("ncclAllGather: ") + ncclGetErrorString(nr); return FV_ERR_COMM; }
        return 0;
    }
    fv_group *g = ctx->group;
    if (!g) return FV_ERR_STATE;
    FV_HIP(hipEventRecord(g->ans_ready[(size_t)ctx->group_rank], ctx->stream));
    if (!g->barrier.arrive_and_wait()) { ctx->detail = "multi-device decode: another member failed"; return FV_ERR_COMM; }
    for (size_t q = 0; q < g->members.size(); ++q) {
        const fv_ctx *peer = g->members[q];
        if (peer != ctx) FV_HIP(hipStreamWaitEvent(ctx->stream, g->ans_ready[q], 0));
        FV_HIP(hipMemcpyAsync(ctx->d_gather.p + q * (size_t)T, peer->d_ans.p, (size_t)T * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return 0;
}

bool fv_group_barrier::arrive_and_wait()
{
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    const unsigned my = generation;
    if (++waiting == n) { waiting = 0; ++generation; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return generation != my || failed; });
    return !failed;
}

void fv_group_barrier::fail()
{
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
}

void fv_group_barrier::reset(int members)
{
    std::lock_guard<std::mutex> lk(mu);
    n = members; waiting = 0; failed = false;
}

int group_run(fv_ctx *ctx, int T, int *path_out, float *score_out, const std::function<int(fv_ctx *, int *, float *)> &fn)
{
    fv_group *g = ctx->group;
    const int n = (int)g->members.size();
    g->barrier.reset(n);
    std::vector<std::vector<int>> paths((size_t)n, std::vector<int>((size_t)std::max(T, 1)));
    std::vector<float> scores((size_t)n, 0.0f);
    std::vector<int> rcs((size_t)n, 0);
    auto body = [&](int r) {
        fv_ctx *m = g->members[(size_t)r];
        int rc = hipSetDevice(m->device) == hipSuccess ? fn(m, r == 0 ? path_out : paths[(size_t)r].data(), &scores[(size_t)r]) : FV_ERR_DEVICE;
        if (rc < 0) g->barrier.fail();           // peers waiting at the gather give up instead of hanging
        rcs[(size_t)r] = rc;
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n; ++r) th.emplace_back(body, r);
    body(0);
    for (auto &t : th) t.join();
    (void)hipSetDevice(ctx->device);
    int rc = 0;
    for (int r = 0; r < n; ++r) {
        if (rcs[(size_t)r] < 0 && rc >= 0) { rc = rcs[(size_t)r]; if (r) ctx->detail = "member " + std::to_string(r) + ": " + g->members[(size_t)r]->detail; }
        else if (rcs[(size_t)r] > rc && rc >= 0) rc = rcs[(size_t)r];
    }
    if (score_out) *score_out = scores[0];
    return rc;
}

}  // namespace fvi

extern "C" int fv_merge_paths(int T, int n_split, int nranks, const int *gathered, int *path_out)
{
    if (!gathered || !path_out || nranks < 1) return FV_ERR_ARG;
    fv::Plan plan;
    int rc = fv::build_plan(T, n_split, FV_MODE_REFERENCE, nranks, plan);
    if (rc) return rc;
    std::vector<int> g(gathered, gathered + (size_t)T * nranks);
    fvi::merge_gathered(plan, g, T, nranks, path_out);
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
    if (ctx->comm || ctx->group) return FV_ERR_STATE;
    ctx->rank = rank; ctx->nranks = nranks;
    return FV_OK;
}

extern "C" int fv_comm_init(fv_ctx *ctx, int rank, int nranks, const void *id)
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return FV_ERR_ARG;
    if (ctx->comm || ctx->group) return FV_ERR_STATE;
    FV_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t nr = ncclCommInitRank(&ctx->comm, nranks, uid, rank);
    if (nr != ncclSuccess) { ctx->detail = std::string("ncclCommInitRank: ") + ncclGetErrorString(nr); ctx->comm = nullptr; return FV_ERR_COMM; }
    ctx->rank = rank; ctx->nranks = nranks;
    return FV_OK;
}

extern "C" int fv_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// One host process, one member context per listed device: the reference's shape (one process whose MAX_THREADS workers
// share one queue, FLASH:316-368) with GPUs in place of threads.
extern "C" int fv_create_multi(fv_ctx **out, const int *devices, int ndev)
{
    if (!out || !devices || ndev < 1 || ndev > 64) return FV_ERR_ARG;
    *out = nullptr;
    fv_group *g = new (std::nothrow) fv_group();
    if (!g) return FV_ERR_NOMEM;
    auto fail = [&](int rc) {
        for (size_t r = g->members.size(); r-- > 0;) { g->members[r]->group = nullptr; fv_destroy(g->members[r]); }
        for (hipEvent_t e : g->ans_ready) if (e) (void)hipEventDestroy(e);
        delete g;
        return rc;
    };
    bool distinct = true;
    for (int r = 0; r < ndev; ++r)
        for (int q = 0; q < r; ++q) distinct &= devices[q] != devices[r];
    for (int r = 0; r < ndev; ++r) {
        fv_ctx *m = nullptr;
        int rc = fv_create(&m, devices[r]);
        if (rc) return fail(rc);
        g->members.push_back(m);
        m->rank = r; m->nranks = ndev;
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(FV_ERR_DEVICE);
        g->ans_ready.push_back(e);
    }
    if (ndev > 1 && distinct) {
        std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
        ncclResult_t nr = ncclCommInitAll(comms.data(), ndev, devices);
        if (nr != ncclSuccess) { g->members[0]->detail = std::string("ncclCommInitAll: ") + ncclGetErrorString(nr); return fail(FV_ERR_COMM); }
        for (int r = 0; r < ndev; ++r) g->members[(size_t)r]->comm = comms[(size_t)r];
        g->rccl = true;
    }
    if (ndev > 1)
        for (int r = 0; r < ndev; ++r) { g->members[(size_t)r]->group = g; g->members[(size_t)r]->group_rank = r; }
    else {                                   // one device: a plain context
        fv_ctx *only = g->members[0];
        g->members.clear();
        fail(0);
        *out = only;
        return FV_OK;
    }
    (void)hipSetDevice(devices[0]);
    *out = g->members[0];
    return FV_OK;
}
