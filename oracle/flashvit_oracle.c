/*
 * flashvit_oracle.c — CPU restatement of the reference decoders (see header).
 * TEST INFRASTRUCTURE ONLY.
 *
 * What is restated, by reference file:line (all under /root/reference/src/):
 *   FLASH_Viterbi_multithread.c     nvviterNdivide :126-202, nvviter :204-262,
 *                                   worker's task split :284-302, calc :338-368
 *   FLASH_BS_Viterbi_multithread.c  heap ops :58-211, nvviterNdivide :295-399,
 *                                   nvviter :401-473, calc :548-577
 *
 * Arithmetic contract (the reference's C expression types, x86-64 SSE2):
 *   init   T1[0][i] = (float)( log((double)X) + log((double)B[i][o]) )      one rounding
 *   step   tmp  = (float)log((double)B[i][o])
 *          s    = tmp + T1[cur][k]                                   float add
 *          ktmp = (float)( (double)s + log((double)A[k][i]) )        double add, one rounding
 *          if (ktmp > score) { arg = k; score = ktmp; }     from score=-FLT_MAX, arg=-1
 * The only change of form: log() of every model entry is taken once up front
 * (same libm, same argument, hence the same double) instead of once per cell,
 * and sizes are run-time arguments instead of #defines.  The pthread pool is
 * replaced by draining the same FIFO on one thread: tasks read only Ans entries
 * fixed by their ancestors, so thread interleaving is not observable.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math, no x87): see oracle/Makefile.
 */
#include "flashvit_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int fvo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

struct fvo_model {
    int K, M;
    double *logA;   /* [K][K]  logA[k*K + i]  = log((double)A[k][i])          */
    double *logAT;  /* [K][K]  logAT[i*K + k] = logA[k*K + i] (walk k contiguously); built by the first
                       full-state decode (the beam variant never reads it: at K = 65536 it is 34 GB) */
    double *logB;   /* [K][M]  log((double)B[i][o])                            */
    double *logPi;  /* [K]                                                     */
};

fvo_model *fvo_model_create(const float *A, const float *B, const float *Pi, int K, int M)
{
    if (K < 1 || M < 1 || !A || !B || !Pi) return NULL;
    fvo_model *m = (fvo_model *)calloc(1, sizeof *m);
    if (!m) return NULL;
    m->K = K; m->M = M;
    size_t kk = (size_t)K * K;
    m->logA = (double *)malloc(kk * sizeof(double));
    m->logAT = NULL;
    m->logB = (double *)malloc((size_t)K * M * sizeof(double));
    m->logPi = (double *)malloc((size_t)K * sizeof(double));
    if (!m->logA || !m->logB || !m->logPi) { fvo_model_destroy(m); return NULL; }
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < K; ++i)
            m->logA[(size_t)k * K + i] = log((double)A[(size_t)k * K + i]);
    for (size_t x = 0; x < (size_t)K * M; ++x) m->logB[x] = log((double)B[x]);
    for (int i = 0; i < K; ++i) m->logPi[i] = log((double)Pi[i]);
    return m;
}

/* Column-major copy for the full-state cell (walks k contiguously); first use only. */
/* The transposed table is built on the first full-state call of a model.  Two host threads sharing one model may get
 * here together: one mutex for all models (the table is built once per model, so contention does not matter), the
 * pointer is published only when the table is complete. */
static pthread_mutex_t logAT_lock = PTHREAD_MUTEX_INITIALIZER;

static int model_need_logAT(fvo_model *m)
{
    int rc = 0;
    pthread_mutex_lock(&logAT_lock);
    if (!m->logAT) {
        const int K = m->K;
        double *t = (double *)malloc((size_t)K * K * sizeof(double));
        if (!t) rc = FVO_ERR_NOMEM;
        else {
            const int BLK = 64;
#pragma omp parallel for schedule(static)
            for (int i0 = 0; i0 < K; i0 += BLK)
                for (int k = 0; k < K; ++k)
                    for (int i = i0; i < i0 + BLK && i < K; ++i) t[(size_t)i * K + k] = m->logA[(size_t)k * K + i];
            m->logAT = t;
        }
    }
    pthread_mutex_unlock(&logAT_lock);
    return rc;
}

void fvo_model_destroy(fvo_model *m)
{
    if (!m) return;
    free(m->logA); free(m->logAT); free(m->logB); free(m->logPi);
    free(m);
}

/* ------------------------------------------------------------------ full -- */

typedef struct { int L, R; } interval;

typedef struct {
    const fvo_model *m;
    const int *ob;
    int T;
    int *ans;
    long long cells;
    int err;
    float final_score;
} full_run;

/* Init row: FLASH:138-151 / :208-222. */
static void full_init_row(const fvo_model *m, int o, int prev_state, float *row)
{
    const int K = m->K, M = m->M;
    for (int i = 0; i < K; ++i) {
        double x = prev_state < 0 ? m->logPi[i] : m->logA[(size_t)prev_state * K + i];
        row[i] = (float)(x + m->logB[(size_t)i * M + o]);
    }
}

/* One destination state of one step: FLASH:167-173 / :233-239. */
static inline float full_cell(const fvo_model *m, const float *t1, int i, int o, int *arg_out)
{
    const int K = m->K;
    const double *col = m->logAT + (size_t)i * K;
    float score = -FLT_MAX;
    int arg = -1;
    float tmp = (float)m->logB[(size_t)i * m->M + o];
    for (int k = 0; k < K; ++k) {
        float s = tmp + t1[k];
        float ktmp = (float)((double)s + col[k]);
        if (ktmp > score) { arg = k; score = ktmp; }
    }
    *arg_out = arg;
    return score;
}

/* nvviter, FLASH:204-262. */
static void full_bisect(full_run *r, int L, int R, int mid, float *T1[2], int *T2[2])
{
    const fvo_model *m = r->m;
    const int K = m->K;
    if (L == 0) {
        full_init_row(m, r->ob[L], -1, T1[0]);
    } else {
        int st = r->ans[L - 1];
        full_init_row(m, r->ob[L], st, T1[0]);
        for (int i = 0; i < K; ++i) T2[0][i] = st;
    }
    int cur = 0;
    for (int j = L + 1; j <= R; ++j) {
        int o = r->ob[j];
        const float *t1 = T1[cur];
        float *n1 = T1[cur ^ 1];
        const int *t2 = T2[cur];
        int *n2 = T2[cur ^ 1];
        const int carry = j > mid + 1;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < K; ++i) {
            int arg;
            n1[i] = full_cell(m, t1, i, o, &arg);
            /* arg == -1: no finite predecessor.  The reference then reads T2[cur][-1]
             * (:242, out of bounds); the value is never used because a -FLT_MAX score
             * can never win a later argmax.  Here the slot is marked -1. */
            n2[i] = arg < 0 ? -1 : (carry ? t2[arg] : arg);
        }
        r->cells += (long long)K * K;
        cur ^= 1;
    }
    int arg = r->ans[R];
    if (L == 0 && R == r->T - 1) {
        float score = T1[cur][0]; arg = 0;
        for (int i = 1; i < K; ++i)
            if (T1[cur][i] > score) { arg = i; score = T1[cur][i]; }
        r->ans[R] = arg;
        r->final_score = score;
    }
    r->ans[mid] = T2[cur][arg];
    if (r->ans[mid] < 0) r->err = FVO_ERR_NO_PRED;   /* end state unreachable: reference output undefined */
}

/* Even split of [L,R] into N parts: FLASH:129-136. */
static void split_points(int L, int R, int N, int *midpoint)
{
    int gap = (R - L) / N, extra = (R - L) % N;
    midpoint[0] = L + gap;
    if (extra) { --extra; ++midpoint[0]; }
    for (int i = 1; i + 1 < N; ++i) {
        midpoint[i] = midpoint[i - 1] + gap;
        if (extra) { --extra; ++midpoint[i]; }
    }
}

/* nvviterNdivide, FLASH:126-202 (only ever called with L=0, R=T-1 by calc :347). */
static int full_ndivide(full_run *r, int L, int R, int N, int *midpoint)
{
    const fvo_model *m = r->m;
    const int K = m->K;
    split_points(L, R, N, midpoint);
    float *T1[2];
    int *T2[2];
    T1[0] = (float *)malloc(sizeof(float) * K); T1[1] = (float *)malloc(sizeof(float) * K);
    T2[0] = (int *)malloc(sizeof(int) * (size_t)(N - 1) * K);
    T2[1] = (int *)malloc(sizeof(int) * (size_t)(N - 1) * K);
    if (!T1[0] || !T1[1] || !T2[0] || !T2[1]) { free(T1[0]); free(T1[1]); free(T2[0]); free(T2[1]); return FVO_ERR_NOMEM; }
    if (L == 0) {
        full_init_row(m, r->ob[L], -1, T1[0]);
    } else {
        int st = r->ans[L - 1];
        full_init_row(m, r->ob[L], st, T1[0]);
        for (int x = 0; x + 1 < N; ++x) for (int i = 0; i < K; ++i) T2[0][(size_t)x * K + i] = st;
    }
    int cur = 0, p = -1;
    for (int j = L + 1; j <= R; ++j) {
        int o = r->ob[j];
        while (p + 2 < N && j > midpoint[p + 1] + 1) ++p;
        const float *t1 = T1[cur];
        float *n1 = T1[cur ^ 1];
        const int *t2 = T2[cur];
        int *n2 = T2[cur ^ 1];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < K; ++i) {
            int arg;
            n1[i] = full_cell(m, t1, i, o, &arg);
            if (arg < 0) { for (int x = 0; x + 1 < N; ++x) n2[(size_t)x * K + i] = -1; continue; }
            for (int x = 0; x <= p; ++x) n2[(size_t)x * K + i] = t2[(size_t)x * K + arg];
            for (int x = p + 1; x + 1 < N; ++x) n2[(size_t)x * K + i] = arg;
        }
        r->cells += (long long)K * K;
        cur ^= 1;
    }
    int arg = r->ans[R];
    if (L == 0 && R == r->T - 1) {
        float score = T1[cur][0]; arg = 0;
        for (int i = 1; i < K; ++i)
            if (T1[cur][i] > score) { arg = i; score = T1[cur][i]; }
        r->ans[R] = arg;
        r->final_score = score;
    }
    for (int x = 0; x + 1 < N; ++x) {
        r->ans[midpoint[x]] = T2[cur][(size_t)x * K + arg];
        if (r->ans[midpoint[x]] < 0) r->err = FVO_ERR_NO_PRED;
    }
    free(T1[0]); free(T1[1]); free(T2[0]); free(T2[1]);
    return 0;
}

int fvo_full_decode(const fvo_model *m, const int *ob, int T, int n_split,
                    int *path, float *score, long long *cells)
{
    if (!m || !ob || !path || T < 2 || n_split < 1) return FVO_ERR_ARG;
    /* T == 2N makes the last top-level segment a single element; the reference then
     * miscounts its tasks and prints a wrong path.  Not restated. */
    if (n_split > 2 && T == 2 * n_split) return FVO_ERR_ARG;
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= m->M) return FVO_ERR_ARG;
    if (model_need_logAT((fvo_model *)m)) return FVO_ERR_NOMEM;
    const int K = m->K;
    int N = n_split;
    full_run r = { m, ob, T, path, 0, 0, 0.0f };
    for (int j = 0; j < T; ++j) path[j] = 0;
    interval *Q = (interval *)malloc(sizeof(interval) * (size_t)(T + N + 2));
    int *midpoint = (int *)malloc(sizeof(int) * (size_t)(N > 1 ? N : 2));
    float *T1[2] = { (float *)malloc(sizeof(float) * K), (float *)malloc(sizeof(float) * K) };
    int *T2[2] = { (int *)malloc(sizeof(int) * K), (int *)malloc(sizeof(int) * K) };
    int rc = 0;
    if (!Q || !midpoint || !T1[0] || !T1[1] || !T2[0] || !T2[1]) { rc = FVO_ERR_NOMEM; goto out; }
    int head = 0, tail = 0; /* Q[head..tail) pending */
    if (N > 2 && T >= (N << 1)) {           /* calc :342 */
        rc = full_ndivide(&r, 0, T - 1, N, midpoint);
        if (rc) goto out;
        Q[tail++] = (interval){ 0, midpoint[0] };
        for (int i = 0; i + 2 < N; ++i) Q[tail++] = (interval){ midpoint[i] + 1, midpoint[i + 1] };
        Q[tail++] = (interval){ midpoint[N - 2] + 1, T - 1 };
    } else {
        Q[tail++] = (interval){ 0, T - 1 };
    }
    while (head < tail) {                    /* worker :284-302, one thread */
        int L = Q[head].L, R = Q[head].R; ++head;
        int mid = (L + R) >> 1;
        full_bisect(&r, L, R, mid, T1, T2);
        if (R <= L + 1) continue;
        Q[tail++] = (interval){ L, mid };
        if (R > mid + 1) Q[tail++] = (interval){ mid + 1, R };
        if (tail > T + N) { rc = FVO_ERR_ARG; break; }
    }
    if (!rc) rc = r.err;
    if (score) *score = r.final_score;
    if (cells) *cells = r.cells;
out:
    free(Q); free(midpoint); free(T1[0]); free(T1[1]); free(T2[0]); free(T2[1]);
    return rc;
}

int fvo_full_forward(const fvo_model *m, const int *ob, int L, int R, int init_state,
                     float *score_row, int *argtab)
{
    if (!m || !ob || L < 0 || R < L || (init_state < 0 && L != 0)) return FVO_ERR_ARG;
    if (model_need_logAT((fvo_model *)m)) return FVO_ERR_NOMEM;
    const int K = m->K;
    float *a = (float *)malloc(sizeof(float) * K), *b = (float *)malloc(sizeof(float) * K);
    if (!a || !b) { free(a); free(b); return FVO_ERR_NOMEM; }
    full_init_row(m, ob[L], init_state, a);
    for (int j = L + 1; j <= R; ++j) {
        int *args = argtab ? argtab + (size_t)(j - L - 1) * K : NULL;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < K; ++i) {
            int arg;
            b[i] = full_cell(m, a, i, ob[j], &arg);
            if (args) args[i] = arg;
        }
        float *t = a; a = b; b = t;
    }
    if (score_row) memcpy(score_row, a, sizeof(float) * K);
    free(a); free(b);
    return 0;
}

/* --------------------------------------------------------------- vanilla -- */

/* viterbi() of Base_line/C implementations/vanilla Viterbi.c:125-173.  Different rounding order from
 * FLASH: tmp2 = (float)( ((double)T1[k][j-1] + log A[k][i]) + log B[i][ob[j]] ) — two double adds, one
 * rounding (:140); end state from (-FLT_MAX, -1) with strict '>' (:153-162). */
int fvo_vanilla_decode(const fvo_model *m, const int *ob, int T, int *path, float *score)
{
    if (!m || !ob || !path || T < 1) return FVO_ERR_ARG;
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= m->M) return FVO_ERR_ARG;
    if (model_need_logAT((fvo_model *)m)) return FVO_ERR_NOMEM;
    const int K = m->K, M = m->M;
    float *a = (float *)malloc(sizeof(float) * K), *b = (float *)malloc(sizeof(float) * K);
    int *T2 = (int *)malloc(sizeof(int) * (size_t)K * T);
    if (!a || !b || !T2) { free(a); free(b); free(T2); return FVO_ERR_NOMEM; }
    full_init_row(m, ob[0], -1, a);
    int rc = 0;
    for (int j = 1; j < T; ++j) {
        const int o = ob[j];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < K; ++i) {
            const double *col = m->logAT + (size_t)i * K;
            const double lb = m->logB[(size_t)i * M + o];
            float tmp = -FLT_MAX;
            int arg = -1;
            for (int k = 0; k < K; ++k) {
                float tmp2 = (float)(((double)a[k] + col[k]) + lb);
                if (tmp2 > tmp) { tmp = tmp2; arg = k; }
            }
            b[i] = tmp;
            T2[(size_t)j * K + i] = arg;
        }
        float *t = a; a = b; b = t;
    }
    float tmp = -FLT_MAX;
    int arg = -1;
    for (int i = 0; i < K; ++i) if (a[i] > tmp) { tmp = a[i]; arg = i; }
    if (arg < 0) rc = FVO_ERR_NO_PRED;
    path[T - 1] = arg;
    for (int j = T - 1; j > 0 && !rc; --j) {
        path[j - 1] = T2[(size_t)j * K + path[j]];
        if (path[j - 1] < 0) rc = FVO_ERR_NO_PRED;
    }
    if (score) *score = tmp;
    free(a); free(b); free(T2);
    return rc;
}

/* ------------------------------------------------------------ checkpoint -- */

/* One vanilla step (checkpoint Viterbi.c:132-148 with T2, :216-226 without): out[i] = max_k
 * (float)(((double)in[k] + log A[k][i]) + log B[i][o]) from (-FLT_MAX, -1), strict '>' in ascending k.
 * The arg-free form of the first pass calls emax(float, float) (:29-32, :222): the double expression is
 * rounded to float at the call, so it holds the same values. */
static void vanilla_step(const fvo_model *m, int o, const float *in, float *out, int *arg_row)
{
    const int K = m->K, M = m->M;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < K; ++i) {
        const double *col = m->logAT + (size_t)i * K;
        const double lb = m->logB[(size_t)i * M + o];
        float tmp = -FLT_MAX;
        int arg = -1;
        for (int k = 0; k < K; ++k) {
            float tmp2 = (float)(((double)in[k] + col[k]) + lb);
            if (tmp2 > tmp) { tmp = tmp2; arg = k; }
        }
        out[i] = tmp;
        if (arg_row) arg_row[i] = arg;
    }
}

/* viterbi_checkpoint() of Base_line/C implementations/checkpoint Viterbi.c:176-251 with `step` as its
 * second argument (main passes 0 => floor(sqrt(T)), :179-180).  First pass (:183-232): the score row
 * after every time j that is a multiple of `step` is kept (checkpoints 0, step, 2*step, ... < T).
 * Second pass, last checkpoint first (:236-248): viterbi_checkpoint_subroutine (:121-174) re-runs the
 * recurrence from the kept row over its segment — up to and including the next checkpoint's time, or to
 * T-1 for the last one — this time with arg rows, picks the end state (last segment only, :152-165) and
 * back-tracks through the segment (:167-171). */
int fvo_checkpoint_decode(const fvo_model *m, const int *ob, int T, int step, int *path, float *score)
{
    if (!m || !ob || !path || T < 1) return FVO_ERR_ARG;
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= m->M) return FVO_ERR_ARG;
    if (model_need_logAT((fvo_model *)m)) return FVO_ERR_NOMEM;
    if (step <= 0) step = (int)floor(sqrt(1.0 * T));
    const int K = m->K;
    const int nck = (T + step - 1) / step;
    float *ck = (float *)malloc(sizeof(float) * (size_t)K * nck);
    float *a = (float *)malloc(sizeof(float) * K), *b = (float *)malloc(sizeof(float) * K);
    int *T2 = (int *)malloc(sizeof(int) * (size_t)K * (step + 1));
    if (!ck || !a || !b || !T2) { free(ck); free(a); free(b); free(T2); return FVO_ERR_NOMEM; }
    full_init_row(m, ob[0], -1, a);                                  /* initT1, :119 */
    memcpy(ck, a, sizeof(float) * K);
    for (int j = 1; j < T; ++j) {
        vanilla_step(m, ob[j], a, b, NULL);
        float *t = a; a = b; b = t;
        if (j % step == 0) memcpy(ck + (size_t)(j / step) * K, a, sizeof(float) * K);
    }
    int rc = 0, count = T - 1;
    for (int c = nck - 1; c >= 0 && !rc; --c) {
        const int start = c * step;
        const int tsub = (c == nck - 1) ? T - start : step + 1;       /* T_sub, :123 */
        memcpy(a, ck + (size_t)c * K, sizeof(float) * K);
        for (int j = 1; j < tsub; ++j) {
            vanilla_step(m, ob[start + j], a, b, T2 + (size_t)j * K);
            float *t = a; a = b; b = t;
        }
        if (c == nck - 1) {
            float tmp = -FLT_MAX;
            int arg = -1;
            for (int k = 0; k < K; ++k) if (a[k] > tmp) { tmp = a[k]; arg = k; }
            if (score) *score = tmp;
            if (arg < 0) { rc = FVO_ERR_NO_PRED; path[count] = arg; break; }
            path[count--] = arg;
        }
        for (int i = tsub - 1; i > 0; --i) {
            path[count] = T2[(size_t)i * K + path[count + 1]];
            if (path[count] < 0) { rc = FVO_ERR_NO_PRED; break; }
            count--;
        }
    }
    free(ck); free(a); free(b); free(T2);
    return rc;
}

/* memory_bytes of the same program (:250): T1_previous + T1[K][checkpointslen] + T1_current +
 * checkpoints[T/step+1] + the largest sizeof(T1_sub)+sizeof(T2_sub) of the subroutine (:173). */
long long fvo_checkpoint_memory_bytes(int K, int T, int step)
{
    if (step <= 0) step = (int)floor(sqrt(1.0 * T));
    const long long nck = (T + step - 1) / step;
    const long long last = T - (nck - 1) * step;
    const long long tsub = nck > 1 && step + 1 > last ? step + 1 : last;
    return 4LL * K + 4LL * K * nck + 4LL * K + 4LL * (T / step + 1) + 8LL * K * tsub;
}

/* ------------------------------------------------------------------ beam -- */

/* FLASH_BS:51-56.  Slot 0's .value carries the element count as a float. */
typedef struct { float value; int state; int t3; } hnode;

static inline void heap_reset(hnode *h) { h[0].value = 0; h[0].state = -1; h[0].t3 = -1; } /* :65-70 */

/* create_min_heap, FLASH_BS:96-123: Floyd build, smaller child preferred with
 * '>' (left wins ties), stop on '<=' (equal keys are not swapped). */
static void heap_build(hnode *h)
{
    int total = (int)h[0].value;
    for (int node = total / 2; node > 0; --node) {
        int parent = node, child = 2 * node;
        hnode temp = h[parent];
        for (; child <= total; child *= 2) {
            if (child + 1 <= total && h[child].value > h[child + 1].value) child++;
            if (temp.value <= h[child].value) break;
            h[parent] = h[child];
            parent = child;
        }
        h[parent] = temp;
    }
}

/* replace_min_heap_element, FLASH_BS:131-165. */
static void heap_replace_min(hnode *h, float v, int state, int t3)
{
    h[1].value = v; h[1].state = state; h[1].t3 = t3;
    int total = (int)h[0].value;
    int parent = 1, child = 2;
    while (child <= total) {
        if (child + 1 <= total && h[child].value > h[child + 1].value) child++;
        if (h[parent].value <= h[child].value) break;
        hnode t = h[parent]; h[parent] = h[child]; h[child] = t;
        parent = child;
        child *= 2;
    }
}

/* generate_state_heap, FLASH_BS:167-211.  States arrive in index order. */
static void heap_offer(hnode *h, int beam, float v, int i, int t3)
{
    if (i < beam - 1) {
        h[i + 1].value = v; h[i + 1].state = i; h[i + 1].t3 = t3;
        h[0].value++;
    } else if (i == beam - 1) {
        h[i + 1].value = v; h[i + 1].state = i; h[i + 1].t3 = t3;
        h[0].value++;
        heap_build(h);
    } else if (v > h[1].value) {
        heap_replace_min(h, v, i, t3);
    }
}

/* Find_T3_State, FLASH_BS:73-86. */
static int heap_find_t3(const hnode *h, int state)
{
    int total = (int)h[0].value;
    for (int i = 1; i <= total; ++i)
        if (h[i].state == state) return h[i].t3;
    return -1;
}

typedef struct {
    const fvo_model *m;
    const int *ob;
    int T, beam;
    int *ans;
    long long cells;
    int err;
    float final_score;
} beam_run;

/* Best predecessor of destination i over the heap slots, slot order, FLASH_BS:439-446. */
static inline float beam_cell(const fvo_model *m, const hnode *h, int beam, int i, int o, int *arg_out)
{
    const int K = m->K;
    float score = -FLT_MAX;
    int arg = -1;
    float tmp = (float)m->logB[(size_t)i * m->M + o];
    for (int k = 0; k < beam; ++k) {
        int pre = h[k + 1].state;
        float s = tmp + h[k + 1].value;
        float ktmp = (float)((double)s + m->logA[(size_t)pre * K + i]);
        if (ktmp > score) { arg = k; score = ktmp; }
    }
    *arg_out = arg;
    return score;
}

/* beam_cell for a block of destinations [i0, i0+n), the heap slots in the OUTER loop: every destination
 * still meets the slots in ascending order with strict '>' from (-FLT_MAX, -1) (FLASH_BS:439-446), so
 * (score, arg) are those of beam_cell bit for bit; only the memory walk differs (the row of log A of a
 * slot is read contiguously instead of one strided entry per cell), which is what makes K = 65536,
 * B = 1024 (BASELINE configs[4]) finish in about a minute.  tests/test_oracle_golden.py runs every beam
 * golden through this form and tests/test_oracle_blocked.py compares it with beam_cell cell by cell. */
#define FVO_BLK 256
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx2", "default")))
#endif
static void beam_block(const fvo_model *m, const hnode *h, int beam, int i0, int n, int o, float *scr, int *argv)
{
    const int K = m->K, M = m->M;
    float tmp[FVO_BLK], sc[FVO_BLK];
    int ar[FVO_BLK];
    for (int i = 0; i < n; ++i) { tmp[i] = (float)m->logB[(size_t)(i0 + i) * M + o]; sc[i] = -FLT_MAX; ar[i] = -1; }
    for (int k = 0; k < beam; ++k) {
        const double *row = m->logA + (size_t)h[k + 1].state * K + i0;
        const float v = h[k + 1].value;
        for (int i = 0; i < n; ++i) {
            const float s = tmp[i] + v;
            const float ktmp = (float)((double)s + row[i]);
            const int gt = ktmp > sc[i];
            sc[i] = gt ? ktmp : sc[i];
            ar[i] = gt ? k : ar[i];
        }
    }
    memcpy(scr + i0, sc, sizeof(float) * (size_t)n);
    memcpy(argv + i0, ar, sizeof(int) * (size_t)n);
}

/* All K destinations of one beam step (the parallel loop of both beam passes). */
static void beam_step_all(const fvo_model *m, const hnode *h, int beam, int o, float *scr, int *argv)
{
    const int K = m->K;
#pragma omp parallel for schedule(static)
    for (int i0 = 0; i0 < K; i0 += FVO_BLK)
        beam_block(m, h, beam, i0, K - i0 < FVO_BLK ? K - i0 : FVO_BLK, o, scr, argv);
}

/* Test hook: one step's scores and args by the cell-at-a-time form (blocked = 0) or the blocked form. */
int fvo_beam_step_probe(const fvo_model *m, const float *hval, const int *hstate, int beam, int o, int blocked,
                        float *scr, int *argv)
{
    if (!m || !hval || !hstate || beam < 1 || o < 0 || o >= m->M) return FVO_ERR_ARG;
    hnode *h = (hnode *)malloc(sizeof(hnode) * ((size_t)beam + 1));
    if (!h) return FVO_ERR_NOMEM;
    heap_reset(h);
    for (int k = 0; k < beam; ++k) { h[k + 1].value = hval[k]; h[k + 1].state = hstate[k]; h[k + 1].t3 = -1; }
    if (blocked) beam_step_all(m, h, beam, o, scr, argv);
    else for (int i = 0; i < m->K; ++i) scr[i] = beam_cell(m, h, beam, i, o, &argv[i]);
    free(h);
    return 0;
}

/* The whole-sequence end pick, FLASH_BS:456-461 / :376-381: slot 1, then slots
 * beam/2+2 .. beam; leaf slot beam/2+1 is never looked at. */
static int beam_final_slot(const hnode *h, int beam, float *score_out)
{
    float score = h[1].value;
    int arg = 0;
    for (int i = beam / 2 + 1; i < beam; ++i)
        if (h[i + 1].value > score) { arg = i; score = h[i + 1].value; }
    *score_out = score;
    return arg;
}

/* nvviter (beam), FLASH_BS:401-473. */
static void beam_bisect(beam_run *r, int L, int R, int mid, hnode *H[2], float *scr, int *argv)
{
    const fvo_model *m = r->m;
    const int K = m->K, M = m->M, beam = r->beam;
    heap_reset(H[0]);
    {
        int o = r->ob[L];
        /* After a beam miss Ans[L-1] is -1 and the reference reads A[-1][i], which in its VIT
         * struct (FLASH_BS:27-30, Pi directly in front of A) is Pi[i]: same row as L == 0. */
        int st = L == 0 ? -1 : r->ans[L - 1];
        for (int i = 0; i < K; ++i) {
            double x = st < 0 ? m->logPi[i] : m->logA[(size_t)st * K + i];
            float tmp = (float)(x + m->logB[(size_t)i * M + o]);
            heap_offer(H[0], beam, tmp, i, -1);
        }
    }
    int cur = 0;
    for (int j = L + 1; j <= R; ++j) {
        int o = r->ob[j];
        const hnode *h = H[cur];
        hnode *n = H[cur ^ 1];
        heap_reset(n);
        beam_step_all(m, h, beam, o, scr, argv);
        const int carry = j > mid + 1;
        for (int i = 0; i < K; ++i) {       /* heap pushes stay in state order */
            int a = argv[i];
            heap_offer(n, beam, scr[i], i, carry ? h[a + 1].t3 : h[a + 1].state);
        }
        r->cells += (long long)K * beam;
        cur ^= 1;
    }
    if (L == 0 && R == r->T - 1) {
        float score;
        int arg = beam_final_slot(H[cur], beam, &score);
        r->ans[R] = H[cur][arg + 1].state;
        r->ans[mid] = H[cur][arg + 1].t3;
        r->final_score = score;
    } else {
        int t3 = heap_find_t3(H[cur], r->ans[R]);
        if (t3 < 0) r->err = FVO_WARN_BEAM_MISS;
        r->ans[mid] = t3;
    }
}

/* nvviterNdivide (beam), FLASH_BS:295-399: N-1 heaps advanced in lock-step; scores
 * and predecessor states are read from heap index 1 (:352-353), T3 per heap. */
static int beam_ndivide(beam_run *r, int L, int R, int N, int *midpoint)
{
    const fvo_model *m = r->m;
    const int K = m->K, M = m->M, beam = r->beam;
    split_points(L, R, N, midpoint);
    const size_t hs = (size_t)beam + 1;
    hnode *H[2];
    H[0] = (hnode *)malloc(sizeof(hnode) * hs * (size_t)(N - 1));
    H[1] = (hnode *)malloc(sizeof(hnode) * hs * (size_t)(N - 1));
    float *scr = (float *)malloc(sizeof(float) * K);
    int *argv = (int *)malloc(sizeof(int) * K);
    if (!H[0] || !H[1] || !scr || !argv) { free(H[0]); free(H[1]); free(scr); free(argv); return FVO_ERR_NOMEM; }
    for (int x = 0; x + 1 < N; ++x) heap_reset(H[0] + x * hs);
    {
        int o = r->ob[L];
        int st = L == 0 ? -1 : r->ans[L - 1];
        for (int i = 0; i < K; ++i) {
            double v = st < 0 ? m->logPi[i] : m->logA[(size_t)st * K + i];
            float tmp = (float)(v + m->logB[(size_t)i * M + o]);
            for (int x = 0; x + 1 < N; ++x) heap_offer(H[0] + x * hs, beam, tmp, i, -1);
        }
    }
    int cur = 0, p = -1;
    for (int j = L + 1; j <= R; ++j) {
        int o = r->ob[j];
        while (p + 2 < N && j > midpoint[p + 1] + 1) ++p;
        for (int x = 0; x + 1 < N; ++x) heap_reset(H[cur ^ 1] + x * hs);
        const hnode *h1 = H[cur] + 1 * hs;          /* H[cur][1], as the reference reads it */
        beam_step_all(m, h1, beam, o, scr, argv);
        for (int i = 0; i < K; ++i) {
            int a = argv[i];
            for (int x = 0; x <= p; ++x)
                heap_offer(H[cur ^ 1] + x * hs, beam, scr[i], i, (H[cur] + x * hs)[a + 1].t3);
            for (int x = p + 1; x + 1 < N; ++x)
                heap_offer(H[cur ^ 1] + x * hs, beam, scr[i], i, (H[cur] + x * hs)[a + 1].state);
        }
        r->cells += (long long)K * beam;
        cur ^= 1;
    }
    if (L == 0 && R == r->T - 1) {
        float score;
        int arg = beam_final_slot(H[cur] + 1 * hs, beam, &score);
        r->ans[R] = (H[cur] + 1 * hs)[arg + 1].state;
        for (int x = 0; x + 1 < N; ++x) r->ans[midpoint[x]] = (H[cur] + x * hs)[arg + 1].t3;
        r->final_score = score;
    } else {
        int st = r->ans[R];
        for (int x = 0; x + 1 < N; ++x) {
            int t3 = heap_find_t3(H[cur] + x * hs, st);
            if (t3 < 0) r->err = FVO_WARN_BEAM_MISS;
            r->ans[midpoint[x]] = t3;
        }
    }
    free(H[0]); free(H[1]); free(scr); free(argv);
    return 0;
}

int fvo_beam_decode(const fvo_model *m, const int *ob, int T, int n_split, int beam,
                    int *path, float *score, long long *cells)
{
    if (!m || !ob || !path || T < 2 || n_split < 1 || beam < 2 || beam > m->K) return FVO_ERR_ARG;
    if (n_split > 2 && T == 2 * n_split) return FVO_ERR_ARG;
    for (int j = 0; j < T; ++j) if (ob[j] < 0 || ob[j] >= m->M) return FVO_ERR_ARG;
    const int K = m->K;
    int N = n_split;
    beam_run r = { m, ob, T, beam, path, 0, 0, 0.0f };
    for (int j = 0; j < T; ++j) path[j] = 0;
    interval *Q = (interval *)malloc(sizeof(interval) * (size_t)(T + N + 2));
    int *midpoint = (int *)malloc(sizeof(int) * (size_t)(N > 1 ? N : 2));
    hnode *H[2] = { (hnode *)malloc(sizeof(hnode) * ((size_t)beam + 1)),
                    (hnode *)malloc(sizeof(hnode) * ((size_t)beam + 1)) };
    float *scr = (float *)malloc(sizeof(float) * K);
    int *argv = (int *)malloc(sizeof(int) * K);
    int rc = 0;
    if (!Q || !midpoint || !H[0] || !H[1] || !scr || !argv) { rc = FVO_ERR_NOMEM; goto out; }
    int head = 0, tail = 0;
    if (N > 2 && T >= (N << 1)) {            /* calc :552 */
        rc = beam_ndivide(&r, 0, T - 1, N, midpoint);
        if (rc) goto out;
        Q[tail++] = (interval){ 0, midpoint[0] };
        for (int i = 0; i + 2 < N; ++i) Q[tail++] = (interval){ midpoint[i] + 1, midpoint[i + 1] };
        Q[tail++] = (interval){ midpoint[N - 2] + 1, T - 1 };
    } else {
        Q[tail++] = (interval){ 0, T - 1 };
    }
    while (head < tail) {
        int L = Q[head].L, R = Q[head].R; ++head;
        int mid = (L + R) >> 1;
        beam_bisect(&r, L, R, mid, H, scr, argv);
        if (R <= L + 1) continue;
        Q[tail++] = (interval){ L, mid };
        if (R > mid + 1) Q[tail++] = (interval){ mid + 1, R };
        if (tail > T + N) { rc = FVO_ERR_ARG; break; }
    }
    if (!rc) rc = r.err;
    if (score) *score = r.final_score;
    if (cells) *cells = r.cells;
out:
    free(Q); free(midpoint); free(H[0]); free(H[1]); free(scr); free(argv);
    return rc;
}

/* ---------------------------------------------------------------- memory -- */

/* sizeof(ThreadPool) on x86-64 glibc: mutex 40 + cond 48 + N*8 + 3 ints, padded to 8. */
static long long pool_bytes(int N) { return ((40 + 48 + 8LL * N + 12 + 7) / 8) * 8; }

long long fvo_full_memory_bytes(int K, int T, int N)
{
    long long mem = 0;
    if (N > 2 && T >= (N << 1))
        mem = 4LL * (N - 1) + 2LL * K * 4 + 2LL * (N - 1) * K * 4;     /* FLASH:355 */
    long long tmp = (long long)N * (2LL * K * 4 + 2LL * K * 4);          /* :364 */
    if (tmp > mem) mem = tmp;
    return mem + pool_bytes(N) + 8;                                      /* :367, sizeof(size_t) quirk */
}

long long fvo_beam_memory_bytes(int K, int T, int N, int beam)
{
    (void)K;
    long long mem = 0;
    if (N > 2 && T >= (N << 1))
        mem = 4LL * (N - 1) + 2LL * (N - 1) * (beam + 1) * 12;           /* FLASH_BS:564 */
    long long tmp = (long long)N * (2LL * (beam + 1) * 12);               /* :573 */
    if (tmp > mem) mem = tmp;
    return mem + pool_bytes(N) + 8;                                       /* :576 */
}
