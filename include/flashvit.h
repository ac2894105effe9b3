/*
 * flashvit.h — C ABI of libflashvit.so, the MI355X (gfx950) implementation of the
 * FLASH / FLASH-BS Viterbi hot path.  extern "C", plain pointers and sizes only.
 *
 * The reference has no library API: its seam is `void calc(void)` over the globals
 * `VIT *vit; ThreadPool pool;` (reference src/FLASH_Viterbi_multithread.c:45-46,338-368;
 * src/FLASH_BS_Viterbi_multithread.c:47-48,548-577), called once from main() inside
 * the clock_gettime bracket (:375-377).  Each entry point below names the piece of
 * that seam it replaces.  Caller owns every host buffer passed in or out; the
 * library never frees caller memory; one ctx is not thread-safe; functions return
 * 0 on success, >0 for warnings that still deliver the reference's result, <0 for
 * errors (never abort, never perror-and-continue as the reference's loader does).
 */
#ifndef FLASHVIT_H
#define FLASHVIT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fv_ctx fv_ctx;

enum {
    FV_OK = 0,
    /* Find_T3_State returned -1 in some task (FLASH_BS:73-86,466-471): the path holds -1
     * entries and later tasks restart from Pi, exactly what the reference binary prints. */
    FV_WARN_BEAM_MISS = 1,
    FV_ERR_ARG = -1,         /* bad pointer / size; T == 2*n_split with n_split > 2 */
    FV_ERR_NOMEM = -2,       /* host or device allocation failed */
    FV_ERR_NO_PRED = -3,     /* a decoded entry has no finite predecessor (reference: T2[cur][-1], UB) */
    FV_ERR_DEVICE = -4,      /* a HIP call failed; fv_last_error_detail() has the text */
    FV_ERR_STATE = -5,       /* decode before fv_set_model, comm calls out of order */
    FV_ERR_UNSUPPORTED = -6, /* size or kernel choice outside what the kernels are built for: a filter kernel forced for a model it
                                cannot take (entries above 1; a score row beyond LDS), a beam width whose heap does not fit LDS.
                                K itself is bounded by device memory only */
    FV_ERR_COMM = -7,        /* RCCL failure */
};

/* `mode` argument of the decode calls. */
enum {
    /* Replays the reference's divide-and-conquer task tree pass for pass (top-level N-way
     * pass, then every bisection task whose rounding history differs from its parent's),
     * so every float the reference compares is reproduced: bit-exact paths by construction. */
    FV_MODE_REFERENCE = 0,
    /* One forward pass over [0,T-1] with full back-pointers + one backtrack.  Equal to the
     * reference in exact arithmetic; float rounding histories of right-hand sub-tasks differ,
     * so equality with the reference binary is empirical (it held on every fixture).
     * fv_decode_beam: ordinary one-pass beam search — NOT the reference's result, whose right-hand
     * tasks re-run the beam conditioned on Ans[mid] and can leave the first pass's beam. */
    FV_MODE_SINGLE_PASS = 1,
};

/* fv_set_option keys. */
enum {
    FV_OPT_KERNEL = 1,      /* FV_KERNEL_* : which trellis-step kernel streams the transition table */
    FV_OPT_MAX_BATCH = 2,   /* 1..8: most independent tasks advanced by one step launch */
    FV_OPT_PROFILE = 3,     /* 0/1: bracket every step launch with HIP events (fills step_kernel_ms) */
    FV_OPT_SEL_MARGIN = 4,  /* FLASH-BS: margin, in 1/1000 of the beam spread (max - cut value), below the extrapolated cut value
                               from which the step kernels collect the next select's candidates (default 300; the margin then follows the
                               length of the lists it produces); speed only */
    FV_OPT_DEBUG = 100,     /* UNSTABLE: kernel-tuning switches (a bit mask) that select alternative forms of a kernel or of the
                               launch schedule.  Every bit the library accepts is speed-only — the parity tests run each
                               alternative against the same goldens (tests/test_boundary.py checks that no accepted value
                               changes a result).  The switches that leave a part of a kernel out to time the rest (bits 0, 4, 5,
                               11, 12 = FV_DEBUG_TIMING_ONLY) change results; they exist only in the separate timing build
                               (libflashvit_timing.so, used by tools/), and this library answers FV_ERR_ARG to them.
                               Full-state: 1 no reverse sweep, 2 alternate load schedule, 3 full last step instead of one
                               column, 6 hipGraph replay of a generation, 13 packed kernel in 16-wave workgroups, 14 packed
                               kernel for every batched launch, 18 right-hand generations on one stream, 21 F64 / F32 / F16 / Q16
                               kernels in three slabs of source rows (the route of K > 65536) at any size.  FLASH-BS: 7 the cut predictor uses the pass's own
                               cuts only, 8 / 9
                               float64 / 16-bit step kernel always, 10 no candidate lists, 15 whole-workgroup select for short
                               lists too, 16 / 17 pass groups on one stream / on four streams whatever the size, 19 every heap
                               layout rebuilt and every tie re-decided whether or not the path needs it, 20 every duplicate
                               step replays its heap at once (no speculative member lists), 22 every selection in the memory-resident
                               form (the route of K > 65536), 23 runs of undecided steps always decided in full, 24 four-wave select also on steps that may have to be resolved,
                               25 / 26 the 16-bit step kernel in 16-wave / 8-wave workgroups whatever the launch size (25 also: the float64
                               step kernel in 16-wave workgroups for small beams too) */
};
#define FV_DEBUG_TIMING_ONLY ((1 << 0) | (1 << 4) | (1 << 5) | (1 << 11) | (1 << 12))
enum {
    FV_KERNEL_AUTO = 0,        /* every model entry in [0,1]: SPARSE_Q16 if <= 35 % of A is non-zero, else U16_REFINE;
                                  otherwise F64_STREAM */
    FV_KERNEL_F64_STREAM = 1,  /* streams log A as float64 (8 B/cell): the reference expression verbatim; any K (score rows
                                  beyond LDS are swept in slabs of source rows, one launch per slab) */
    FV_KERNEL_F32_REFINE = 2,  /* streams (float)log A (4 B/cell), brackets the winner within 2 ulp,
                                  then re-evaluates the few candidates in float64: same bits out */
    FV_KERNEL_F16_REFINE = 3,  /* same scheme with log A rounded to binary16 (2 B/cell) and a window of
                                  2*max|half(L)-L| + 3 ulp: same bits out, a quarter of the float64 bytes; the wider
                                  window costs more refines than the bytes save at K=3965 */
    FV_KERNEL_Q16_REFINE = 4,  /* same scheme with 16-bit fixed point (step = max|log A|/65534): 2 B/cell and
                                  a window ~ step: same bits out; any K (slabs of source rows, as F64_STREAM) */
    FV_KERNEL_U16_REFINE = 6,  /* the Q16 table again, but the filter itself runs in 16-bit fixed point, two cells per packed
                                  instruction (the score row is quantised with the table's step while it is staged into LDS);
                                  candidates inside the window are re-evaluated in float64 as above: same bits out.  Used
                                  for single-task launches (the whole-sequence pass) and for models whose float32 rows do not
                                  fit LDS (K up to 65536; beyond that AUTO takes Q16_REFINE in slabs); batched launches take the
                                  Q16_REFINE filter (same table, same bits) */
    FV_KERNEL_SPARSE_Q16 = 5,  /* the Q16 codes of the NON-ZERO transitions only (per destination column, ascending
                                  source state): log 0 = -inf can never win (FLASH:171), so skipping those cells
                                  changes no bit; 7.6 MB instead of 31.5 MB at K=3965, p=0.112 */
};

typedef struct {
    double set_model_ms;      /* host log() tables + H2D of the last fv_set_model */
    double decode_ms;         /* host wall time of the last decode call (enqueue .. final sync) */
    double gpu_ms;            /* HIP-event time from first to last kernel of the last decode */
    double top_pass_ms;       /* HIP-event time of generation 0 (the whole-sequence pass incl. init/argmax/backtrack) */
    double top_steps_ms;      /* HIP-event time around generation 0's T-1 back-to-back step launches only */
    double step_kernel_ms;    /* sum of per-launch event times of the step kernel (FV_OPT_PROFILE=1) */
    long long step_launches;  /* trellis-step kernel launches in the last decode */
    long long task_steps;     /* sum over launches of tasks advanced by a full K*K (or K*beam) step */
    long long column_steps;   /* last steps of passes evaluated for the one destination column the back-track reads */
    long long cells;          /* add-compare cells: task_steps * K * K + column_steps * K (full) or task_steps * K * beam */
    long long alg_bytes;      /* 4 bytes per cell */
    long long table_bytes_per_step; /* bytes of transition table one step launch streams */
    long long device_bytes;   /* device working set (tables + workspace) */
    long long refine_near;    /* filter kernels: candidates inside the window besides a column's best */
    long long refine_rescan;  /* filter kernels: lanes that had to rescan their rows */
    long long beam_exact_sets;/* FLASH-BS: exact heap replays run for member sets (duplicate scores at the cut whose survivors mattered) */
    long long beam_ties;      /* FLASH-BS: (step, state) cells re-decided by slot order; 0 unless a back-tracked path met a cell
                                 whose maximum two beam entries attained (only then are the heap layouts rebuilt) */
    long long beam_dup_cols;  /* FLASH-BS statistics: columns won by an entry whose value equals a duplicated cut value */
    long long beam_dup_steps; /* ... and the number of steps in which that happened at least once */
    long long beam_cand_selects; /* FLASH-BS: top-B selections that ran on a step's candidate list instead of all K scores */
    double density;           /* non-zero fraction of the transition matrix */
    int passes;               /* forward passes run (reference mode: one per right-hand task) */
    int generations;          /* dependent batches of passes */
    int kernel;               /* FV_KERNEL_* actually used */
    int ranks;                /* ranks sharing the decode (1 without fv_comm_init) */
    long long refine_saturated; /* packed 16-bit filter: (step, column) pairs whose window reached the end of the code range, so that
                                   every source row was re-evaluated exactly (score rows spread wider than max|log A|) */
    long long beam_spec_steps;  /* FLASH-BS: steps whose cut fell on duplicated scores and that were carried speculatively (all duplicates
                                   in the member list, survivors undecided) instead of replaying the heap at once */
    long long beam_reach_events;/* FLASH-BS: selects at which an undecided duplicate's column reached the beam, so that the steps before
                                   it had to be decided (exact replays, counted in beam_exact_sets) and the selection repeated */
    long long beam_list_short;  /* FLASH-BS: selects (third step of a pass on) whose candidate list held fewer than B entries ... */
    long long beam_list_long;   /* ... or more than its capacity: both re-read all K scores */
    long long beam_list_entries;/* FLASH-BS: total length of the candidate lists the selects ran on (beam_cand_selects of them) */
    long long beam_chain_cuts;  /* FLASH-BS: runs of undecided steps that were decided only back to a step whose replay provably does not
                                   depend on the undecided columns (instead of back to the last step with exact scores) */
} fv_stats;

/* Device + stream + workspace owner.  Replaces `vit = create_vit()`'s allocation role
 * (FLASH:97-107) and the ThreadPool globals (:36-46). */
int fv_create(fv_ctx **out, int device);
void fv_destroy(fv_ctx *ctx);

/* One host process, several GPUs: the reference is one process whose MAX_THREADS workers share one task queue
 * (FLASH:316-335, calc :338-368); here the workers are devices.  The returned context is used like any other
 * (fv_set_model uploads to every device, fv_set_option applies to all): a decode runs the whole-sequence pass on
 * every device (the same deterministic computation, no communication), deals the n_split top-level segments
 * (:349-353) round-robin to the devices — one host thread and one stream set per device — and merges the answer
 * arrays with ONE grouped ncclAllGather (ncclCommInitAll) of T int32 over xGMI.  A device may be listed more than
 * once (a 1-GPU lease): the control flow is the same, with device-to-device copies in place of RCCL (two RCCL ranks
 * cannot share a device).  ndev = 1 gives a plain fv_create context.  fv_last_stats reports the first device's
 * decode and `ranks` = ndev.  fv_device_count: GPUs visible to the process (0 if none). */
int fv_create_multi(fv_ctx **out, const int *devices, int ndev);
int fv_device_count(void);

/* Model upload.  A is K*K row-major A[from][to], B is K*M row-major B[state][symbol],
 * Pi is K — the VIT fields of FLASH:26-28 as InitElement filled them (:82-93).  Takes
 * log() of every entry in double with the host libm (the calls the reference makes per
 * cell, :142,150,167,170) and ships the tables; the caller keeps ownership of A/B/Pi. */
int fv_set_model(fv_ctx *ctx, const float *A, const float *B, const float *Pi, int K, int M);

int fv_set_option(fv_ctx *ctx, int key, long long value);

/* calc() of FLASH_Viterbi_multithread.c:338-368 with MAX_THREADS = n_split:
 * ob[T] = vit->Obroute, path_out[T] = vit->Ans, *score_out = T1[cur][Ans[T-1]] of the
 * whole-sequence pass (the reference never prints it). */
int fv_decode_full(fv_ctx *ctx, const int *ob, int T, int n_split, int mode,
                   int *path_out, float *score_out);

/* calc() of FLASH_BS_Viterbi_multithread.c:548-577 with MAX_THREADS = n_split and
 * BeamSearchWidth = beam_width (2 <= beam_width <= K). */
int fv_decode_beam(fv_ctx *ctx, const int *ob, int T, int n_split, int beam_width, int mode,
                   int *path_out, float *score_out);

/* GPU form of the reference's baseline programs Base_line/C implementations/vanilla Viterbi.c:125-173 and
 * checkpoint Viterbi.c (same recurrence, hence the same output): one forward pass with the BASELINE's
 * rounding order, tmp2 = (float)(((double)T1[k] + log A[k][i]) + log B[i][o]) (:140), lowest-index
 * end state, plain back-track.  An independent cross-check of the FLASH paths (its scores may differ from
 * FLASH's in the last ulp, its path equals FLASH's on every fixture) and the measure of what the
 * divide-and-conquer schedule costs on a device where T*K back-pointers fit trivially. */
int fv_decode_vanilla(fv_ctx *ctx, const int *ob, int T, int *path_out, float *score_out);

/* GPU form of Base_line/C implementations/checkpoint Viterbi.c:176-251 (viterbi_checkpoint(vit, step); step
 * <= 0 selects floor(sqrt(T)), what its main passes): a first pass that keeps the score row of every
 * step-th time only, then every segment re-run from its kept row with arg rows.  The arithmetic is
 * vanilla's, so path and score equal fv_decode_vanilla's; what differs is the work (about 2x) and the
 * memory the CPU program needs, reported by fv_checkpoint_memory_bytes as that program computes it (:250). */
int fv_decode_checkpoint(fv_ctx *ctx, const int *ob, int T, int step, int *path_out, float *score_out);
long long fv_checkpoint_memory_bytes(int K, int T, int step);

int fv_last_stats(const fv_ctx *ctx, fv_stats *out);
const char *fv_strerror(int rc);
const char *fv_last_error_detail(const fv_ctx *ctx);

/* vit->memory_bytes as the reference computes it (FLASH:355,364-367; FLASH_BS:564,573-576)
 * — a sizeof formula, not a measurement; beam_width = 0 selects the full variant. */
long long fv_reference_memory_bytes(int K, int T, int n_split, int beam_width);

/* Multi-GPU: one process per GPU.  Rank 0 calls fv_comm_unique_id and hands the 128 bytes
 * to every rank by any means (bench.py uses torch.distributed); every rank then calls
 * fv_comm_init.  After that each decode call is collective: the whole-sequence pass runs
 * on every rank, the n_split top-level segments (FLASH:349-353) are dealt round-robin to
 * ranks, and one RCCL all-gather over xGMI merges the per-rank path slices.  Replaces the
 * shared-memory work queue of worker() (FLASH:264-308). */
#define FV_UNIQUE_ID_BYTES 128
int fv_comm_unique_id(void *id_out);
int fv_comm_init(fv_ctx *ctx, int rank, int nranks, const void *id);

/* Sharding without a communicator: the decode calls run the whole-sequence pass plus only the
 * segments rank `rank` of `nranks` owns and return that rank's answer array (positions owned by other
 * ranks hold the whole-pass chain); the caller gathers the nranks arrays by its own means and applies
 * fv_merge_paths.  For hosts that already have a collective layer (bench.py falls back to this with
 * torch.distributed if fv_comm_init fails). */
int fv_set_partition(fv_ctx *ctx, int rank, int nranks);

/* The merge applied after the all-gather, exposed for CPU tests: gathered = nranks arrays of T
 * answers (rank-major); position j is taken from the rank owning the top-level segment that
 * contains it, segment end points from rank 0. */
int fv_merge_paths(int T, int n_split, int nranks, const int *gathered, int *path_out);

/* Host-side schedule, exposed so it can be tested without a GPU.  Fills at most `cap`
 * entries of (L, R, generation, owner_rank) per forward pass, in launch order, and returns
 * the number of passes (or <0).  Pass 0 is always the whole-sequence pass. */
typedef struct { int L, R, generation, owner; } fv_pass_info;
int fv_plan_passes(int T, int n_split, int mode, int nranks, fv_pass_info *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
