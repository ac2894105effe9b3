/*
 * flashvit_oracle.h — CPU restatement of the reference FLASH / FLASH-BS Viterbi
 * decoders.  TEST INFRASTRUCTURE ONLY: nothing outside tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libflashvit.so) never links or calls it.
 *
 * Parity pinning: validated path-for-path against binaries compiled from the
 * reference's own sources (oracle/build_ref.py -> oracle/_ref/) on every
 * fixture under tests/golden/ (tests/test_oracle_golden.py).
 */
#ifndef FLASHVIT_ORACLE_H
#define FLASHVIT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FVO_OK = 0,
    FVO_ERR_ARG = -1,       /* bad sizes (K<1, T<2, beam > K, ...) */
    FVO_ERR_NOMEM = -2,
    FVO_ERR_NO_PRED = -3,   /* a decoded entry came from a state without finite predecessor: the
                               reference's output is undefined there (T2[cur][-1], FLASH:242,261) */
    FVO_WARN_BEAM_MISS = 1, /* Find_T3_State returned -1 (FLASH_BS:73-86): path holds -1 entries and
                               later tasks restart from Pi, exactly as the reference binary behaves */
};

typedef struct fvo_model fvo_model;

/* OpenMP threads the per-step loops over destination states use (results do not depend on it);
 * returns the number in effect. */
int fvo_set_threads(int n);

/* Takes log() of every model entry in double with this host's libm, exactly the
 * calls the reference makes per trellis cell (FLASH_Viterbi_multithread.c:142,150,167,170). */
fvo_model *fvo_model_create(const float *A, const float *B, const float *Pi, int K, int M);
void fvo_model_destroy(fvo_model *m);

/* calc() of FLASH_Viterbi_multithread.c:338-368 with MAX_THREADS = n_split.
 * path[T]; *score = T1[cur][Ans[T-1]] of the whole-sequence pass; *cells = number of
 * (k,i) add-compare cells evaluated over all passes. */
int fvo_full_decode(const fvo_model *m, const int *ob, int T, int n_split,
                    int *path, float *score, long long *cells);

/* calc() of FLASH_BS_Viterbi_multithread.c:548-577 with MAX_THREADS = n_split,
 * BeamSearchWidth = beam. */
int fvo_beam_decode(const fvo_model *m, const int *ob, int T, int n_split, int beam,
                    int *path, float *score, long long *cells);

/* Test hook: scores and winning slots of one beam step over the slots (hval, hstate)[beam], by the
 * cell-at-a-time restatement of FLASH_BS:437-446 (blocked = 0) or the row-streaming form the decoder uses. */
int fvo_beam_step_probe(const fvo_model *m, const float *hval, const int *hstate, int beam, int o, int blocked,
                        float *scr, int *argv);

/* viterbi() of the reference's baseline Base_line/C implementations/vanilla Viterbi.c:125-173 (also the
 * output of its checkpoint Viterbi.c: same recurrence, different memory schedule). */
int fvo_vanilla_decode(const fvo_model *m, const int *ob, int T, int *path, float *score);

/* viterbi_checkpoint(vit, step) of Base_line/C implementations/checkpoint Viterbi.c:176-251 (step <= 0:
 * floor(sqrt(T)), as its main does) and the memory_bytes it reports (:250). */
int fvo_checkpoint_decode(const fvo_model *m, const int *ob, int T, int step, int *path, float *score);
long long fvo_checkpoint_memory_bytes(int K, int T, int step);

/* One plain forward pass over [L,R] (nvviter's recurrence, :204-246) that keeps
 * every arg row: score_row[K] = T1 after the last step, argtab[(R-L)*K] = arg of
 * step j at row j-L-1 (-1 where no finite predecessor).  init_state < 0 => start
 * from Pi (L must be 0).  Used to check individual HIP kernels. */
int fvo_full_forward(const fvo_model *m, const int *ob, int L, int R, int init_state,
                     float *score_row, int *argtab);

/* The reference's memory_bytes formulas (FLASH:355,364-367 / FLASH_BS:564,573-576),
 * evaluated for the x86-64 glibc struct sizes the reference binary has. */
long long fvo_full_memory_bytes(int K, int T, int n_split);
long long fvo_beam_memory_bytes(int K, int T, int n_split, int beam);

#ifdef __cplusplus
}
#endif
#endif
