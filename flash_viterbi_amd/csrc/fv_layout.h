// fv_layout.h — table layout, workgroup geometry and LDS budgets of the full-state kernels: the part of
// fv_kernels.hip.inc the host-only translation units (model upload, workspace sizing) need as well.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace fvk {

constexpr int TILE_W = 16;                 // destination states (columns) per workgroup: 64 B of f32 per row
constexpr int ROWG = 64 / TILE_W;          // source rows covered by one wave-wide load (4)
constexpr int NWAVES = 16;
constexpr int BLOCK = NWAVES * 64;         // 1024 threads: one workgroup per CU keeps 64+ KB of loads in flight
constexpr int ROW_ALIGN = 32;              // nrows = roundup(K, 32): whole row blocks for every table type
constexpr int MAX_BATCH = 8;               // independent tasks sharing one sweep of the table

// Tile-major layout with an R-row interleave: R = 4 for the f32 / f64 tables, 8 for the f16 table.
template <int R>
__host__ __device__ inline size_t tab_index(int k, int col, int nrows)
{
    return (size_t)(col >> 4) * nrows * TILE_W + ((size_t)(k / R) * TILE_W + (col & 15)) * R + (k % R);
}
// back-tracks of at least BT_MIN_PARALLEL steps are walked by 64 lanes, each starting BT_WARMUP steps above its chunk (fvk::backtrack)
constexpr int BT_WARMUP = 32, BT_MIN_PARALLEL = 128;
struct PassDesc { int L, R, from_pi, whole; long long row_off; };   // row_off: float offset of the pass's 2 score rows
constexpr int PASS_CHUNK = 64;
struct PassChunk { int n; PassDesc p[PASS_CHUNK]; };

// LDS of one trellis_step / trellis_step_u16 workgroup (score rows + reduction scratch)
template <int NB>
static inline size_t step_lds_bytes(int nrows)
{
    return (size_t)nrows * NB * 4 + (size_t)NB * NWAVES * TILE_W * 12 + (size_t)NB * TILE_W * 4;
}

template <int NB, int NWV>
static inline size_t u16_lds_bytes(int nrows)
{
    return (size_t)nrows * NB * 2 + (size_t)NB * NWV * TILE_W * 12 + (size_t)NB * TILE_W * 4;
}

}  // namespace fvk
