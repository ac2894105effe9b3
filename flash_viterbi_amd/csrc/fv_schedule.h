// fv_schedule.h — host-side task tree of the FLASH divide-and-conquer (no HIP here).
#pragma once
#include <vector>

namespace fv {

struct Pass {
    int L, R;          // time range; steps j = L+1 .. R
    bool from_pi;      // init row from Pi (L == 0) instead of A[Ans[L-1]][*]
    bool whole;        // the whole-sequence pass: end state = argmax of the last row
    int generation;    // passes of one generation are mutually independent
    int owner;         // rank that runs it (-1: every rank)
};

struct Plan {
    std::vector<Pass> passes;             // sorted by generation
    std::vector<int> gen_begin;           // passes[gen_begin[g] .. gen_begin[g+1]) is generation g
    std::vector<int> midpoints;           // top-level split points (empty when no N-way split)
    std::vector<int> seg_L, seg_R, seg_owner;   // top-level segments and their ranks
    int generations() const { return (int)gen_begin.size() - 1; }
};

// Even split of [L,R] into N parts — reference FLASH_Viterbi_multithread.c:129-136.
void split_points(int L, int R, int N, std::vector<int> &mid);

// mode 0: the reference's task tree (calc :338-368 + worker :284-302), with every task
// whose forward pass repeats a prefix of an already-run pass folded into that pass.
// mode 1: one pass over [0,T-1].
// Returns 0 or a negative FV_ERR_* code.
int build_plan(int T, int n_split, int mode, int nranks, Plan &plan);

}  // namespace fv
