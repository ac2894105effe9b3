// fv_schedule.cpp — which forward passes a decode needs, and in what order.
//
// The reference (src/FLASH_Viterbi_multithread.c) drains a FIFO of tasks [L,R]; each
// task runs a forward pass over [L,R] (nvviter, :204-262) that only yields Ans[mid],
// mid = (L+R)>>1, then enqueues [L,mid] and, if R > mid+1, [mid+1,R] (:290-302).
//
// Two facts shrink that to far fewer passes without changing a single float:
//  (1) a pass that keeps every arg row ("back-pointers") can be back-tracked from
//      Ans[R]; the chain c[L..R-1] it yields satisfies c[mid] == the T2[cur][Ans[R]]
//      the reference reads (:261), because T2 is exactly that chain propagated forward
//      (:242);
//  (2) the left child [L,mid] starts from the same init row and performs the same
//      steps as its parent did over [L,mid], and its end state Ans[mid] = c[mid] lies
//      on the parent's chain, so its own chain is the parent's chain: every task on a
//      pass's left spine is answered by that pass.  The same holds for the first
//      top-level segment [0,m0] under the whole-sequence pass (:347,350).
// Only right children [mid+1,R] (re-initialised from Ans[mid], :215-223) need a pass
// of their own.  Each pass writes c[L..R-1] into the answer array; passes of later
// generations overwrite the positions they own, and the last writer of position j is
// the pass on whose left spine the task with mid == j lies.
#include "fv_schedule.h"

#include <algorithm>

#include "flashvit.h"

namespace fv {

void split_points(int L, int R, int N, std::vector<int> &mid)
{
    mid.assign(N - 1, 0);
    int gap = (R - L) / N, extra = (R - L) % N;
    mid[0] = L + gap;
    if (extra) { --extra; ++mid[0]; }
    for (int i = 1; i + 1 < N; ++i) {
        mid[i] = mid[i - 1] + gap;
        if (extra) { --extra; ++mid[i]; }
    }
}

namespace {

struct Builder {
    std::vector<Pass> out;
    // Task [L,R] answered by a chain that already exists when generation `gen` starts.
    void expand(int L, int R, int gen, int owner)
    {
        while (R > L + 1) {
            int mid = (L + R) >> 1;
            if (R > mid + 1) new_pass(mid + 1, R, gen, owner);
            R = mid;                       // left child: same chain
        }
    }
    void new_pass(int L, int R, int gen, int owner)
    {
        out.push_back(Pass{ L, R, false, false, gen, owner });
        expand(L, R, gen + 1, owner);
    }
};

}  // namespace

int build_plan(int T, int n_split, int mode, int nranks, Plan &plan)
{
    if (T < 2 || n_split < 1 || nranks < 1) return FV_ERR_ARG;
    plan = Plan();
    Builder b;
    b.out.push_back(Pass{ 0, T - 1, true, true, 0, -1 });
    if (mode == FV_MODE_SINGLE_PASS) {
        // nothing else: the whole-sequence chain is the answer
    } else if (mode == FV_MODE_REFERENCE) {
        const int N = n_split;
        if (N > 2 && T == 2 * N) return FV_ERR_ARG;   // reference miscounts here
        if (N > 2 && T >= (N << 1)) {                  // calc :342
            split_points(0, T - 1, N, plan.midpoints);
            for (int s = 0; s < N; ++s) {              // seeds of calc :349-353
                int L = s == 0 ? 0 : plan.midpoints[s - 1] + 1;
                int R = s == N - 1 ? T - 1 : plan.midpoints[s];
                int owner = s % nranks;
                plan.seg_L.push_back(L); plan.seg_R.push_back(R); plan.seg_owner.push_back(owner);
                if (s == 0) b.expand(L, R, 1, owner);  // same init and steps as the whole pass
                else b.new_pass(L, R, 1, owner);
            }
        } else {
            b.expand(0, T - 1, 1, -1);                 // calc :358-359
        }
    } else {
        return FV_ERR_ARG;
    }
    std::stable_sort(b.out.begin(), b.out.end(),
                     [](const Pass &x, const Pass &y) { return x.generation < y.generation; });
    plan.passes.swap(b.out);
    int g = -1;
    for (int i = 0; i < (int)plan.passes.size(); ++i)
        while (g < plan.passes[i].generation) { plan.gen_begin.push_back(i); ++g; }
    plan.gen_begin.push_back((int)plan.passes.size());
    return 0;
}

}  // namespace fv

extern "C" int fv_plan_passes(int T, int n_split, int mode, int nranks, fv_pass_info *out, int cap)
{
    fv::Plan plan;
    int rc = fv::build_plan(T, n_split, mode, nranks, plan);
    if (rc) return rc;
    int n = (int)plan.passes.size();
    for (int i = 0; i < n && i < cap && out; ++i) {
        out[i].L = plan.passes[i].L; out[i].R = plan.passes[i].R;
        out[i].generation = plan.passes[i].generation; out[i].owner = plan.passes[i].owner;
    }
    return n;
}
