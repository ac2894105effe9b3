// fv_internal.h — what the translation units of libflashvit.so share: the context behind the opaque fv_ctx of
// include/flashvit.h, device buffers, the error macro and the few functions that cross a seam.
//   fv_context.hip  fv_create / fv_destroy / fv_set_model / options / statistics, workspace, decode epilogue
//   fv_full.hip     full-state kernels and their decode drivers (FLASH, vanilla, checkpoint)
//   fv_beam.hip     FLASH-BS kernels and their decode driver
//   fv_comm.hip     multi-GPU: partition, RCCL all-gather, merge, the single-process multi-device context
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "flashvit.h"
#include "fv_layout.h"
#include "fv_schedule.h"

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

namespace fvb { struct HNode { float v; int s; }; }         // heap node / select candidate: {value, state}

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
    bool u16_ok = false;     // the packed 16-bit kernel can (one row of 16-bit score codes fits LDS: K <= 65536; beyond: float64 slabs)
    bool logs_nonpositive = false;
    DevBuf<float> LA32, LB32T;
    DevBuf<unsigned short> LA16, LAQ16;
    DevBuf<uint4> SPdata;    // sparse CSC-Q16 table (fv_kernels.hip.inc, trellis_step_sparse)
    DevBuf<int> SPoff, SPnwb;
    double density = 1.0;    // finite fraction of log A
    float window16 = 0.0f;   // 2 * max |half(L) - L| over the finite table entries
    float windowq = 0.0f, qscale = -1.0f;   // same for the fixed-point table; value = code * qscale
    bool laq16_ready = false; // LAQ16 holds this model's codes and windowq / qscale belong to it (host-built in fv_set_model, or
                              // built on the device by the first full-state decode of a model beyond the float32 kernels' limit)
    DevBuf<double> LA64, LB64T, LPi64;

    // workspace
    DevBuf<int> d_ob, d_ans, d_bp, d_gather;
    DevBuf<float> d_rows, d_score, d_ckpt;            // d_ckpt: kept score rows of fv_decode_checkpoint
    DevBuf<unsigned long long> d_counters;
    // decode epilogue: path (or the gathered paths), score and counters are packed into one device block and come back
    // in ONE copy into pinned host memory (three small pageable copies cost ~15 us each)
    DevBuf<int> d_pack;
    int *h_pin = nullptr;
    size_t h_pin_n = 0;          // ints
    // beam workspace
    DevBuf<float> d_hval, d_scores, d_slot_val;      // [T][B] members, [T][K] scores, [T][B] exact layout
    DevBuf<int> d_hstate, d_slot_state, d_flags;
    DevBuf<double> LA64R;                            // row-gather copy of the float64 table (built on first beam decode)
    DevBuf<unsigned short> LAQ16R;                   // row-major fixed-point table of beam_step_q16 (built on the first beam decode, with d_qaux's parameters)
    bool rowq_ready = false, beam_q16_ready = false; // LAQ16R holds this model's codes / d_qaux holds its parameters
    DevBuf<unsigned long long> d_qaux;               // [0] lmax bits, [1] dmax bits, then {qscale, window} as floats (q16_params)
    DevBuf<int2> d_tie_list;
    DevBuf<float> d_cut;         // [T][CUT_W] theta, duplicate flag, predicted lower bound of the next cut (topb_select)
    DevBuf<fvb::HNode> d_cand;   // [T][cand_cap] candidate lists of the selects (beam_step epilogue)
    DevBuf<int> d_cand_count;    // [T]
    float opt_sel_margin = 0.3f; // FV_OPT_SEL_MARGIN (in 1/1000): starting margin of the predicted cut bound in beam spreads
    DevBuf<int> d_dupwin;        // [T]
    DevBuf<int> d_doubt, d_doubt_count;   // [T][DOUBT_CAP] columns of step j won by an undecided cut duplicate of step j - 1, [T] their number
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
    // single-process multi-device context (fv_create_multi): every member points to the group; member 0 is the handle
    // the caller holds and the one that owns the group
    struct fv_group *group = nullptr;
    int group_rank = 0;

    fv_stats stats{};
};

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
inline int fv_ldq(int K) { return round_up(K, 128); }         // pitch of the row-major 16-bit table

namespace fvi {

// what fv_set_model computes on the host, once per model
struct HostTables {
    int K = 0, M = 0, nrows = 0, ntiles = 0;
    size_t tab = 0;                      // entries of one tile-major table
    bool full_ok = false, u16_ok = false, any_big = false;
    std::vector<double> h64, b64, pi64;
    std::vector<float> h32, b32;
    std::vector<unsigned short> h16, hq;
    std::vector<uint4> sp;               // sparse CSC-Q16 table
    std::vector<int> sp_off, sp_nwb;
    float window16 = 0.0f, windowq = 0.0f, qscale = -1.0f;
    double density = 1.0;
};
int build_host_tables(const float *A, const float *B, const float *Pi, int K, int M, HostTables &h, std::string &detail);
int upload_tables(fv_ctx *ctx, const HostTables &h);

size_t device_bytes(const fv_ctx *c);
int ensure_workspace(fv_ctx *ctx, int T, size_t rows_needed);
// decode epilogue: (multi-rank: all-gather + merge,) path / score / counters to the host, one sync, statistics
int finish_decode(fv_ctx *ctx, const fv::Plan &plan, int T, int *path_out, float *score_out, clk::time_point t0,
                  size_t nprof, bool beam);
int drained(fv_ctx *ctx, int rc);
// decode prologue: observation sequence to the device (through the pinned block), counters and answers cleared
int begin_decode(fv_ctx *ctx, const int *ob, int T);
// big-LDS attributes of the kernels each translation unit owns
int full_setup(fv_ctx *ctx);
int beam_setup(fv_ctx *ctx);
// fvk::init_rows lives with the full-state kernels; the beam driver starts its passes from the same rows
int launch_init_rows(fv_ctx *ctx, const fvk::PassChunk &ch, float *rows);
// fv_comm.hip
// Multi-device context: one host thread per member runs the same decode on its own device (whole-sequence pass + the
// segments the member owns), the members meet in ONE gather of their answer arrays.
struct fv_group_barrier {
    std::mutex mu;
    std::condition_variable cv;
    int n = 1, waiting = 0;
    unsigned generation = 0;
    bool failed = false;
    bool arrive_and_wait();              // false: a member failed, nobody waits any longer
    void fail();
    void reset(int members);
};
inline int group_size(const fv_ctx *ctx);
inline fv_ctx *group_member(fv_ctx *ctx, int r);
// runs fn(member, path, score) for every member (member 0 on the calling thread), returns the worst return code and
// member 0's merged path / score
int group_run(fv_ctx *ctx, int T, int *path_out, float *score_out, const std::function<int(fv_ctx *, int *, float *)> &fn);
int gather_answers(fv_ctx *ctx, int T);      // ncclAllGather of d_ans (T int32 per rank) into d_gather, on ctx->stream
void merge_gathered(const fv::Plan &plan, const std::vector<int> &gathered, int T, int nranks, int *path);

}  // namespace fvi

struct fv_group {
    std::vector<fv_ctx *> members;       // members[0]: the context the caller holds
    bool rccl = false;                   // distinct devices: the members carry the communicators of one ncclCommInitAll;
                                         // else (a device listed more than once) the gather is device-to-device copies
    fvi::fv_group_barrier barrier;
    std::vector<hipEvent_t> ans_ready;   // per member: its answer array is final (copy gather)
};

namespace fvi {
inline int group_size(const fv_ctx *ctx) { return ctx->group ? (int)ctx->group->members.size() : 1; }
inline fv_ctx *group_member(fv_ctx *ctx, int r) { return ctx->group ? ctx->group->members[(size_t)r] : ctx; }
}  // namespace fvi
