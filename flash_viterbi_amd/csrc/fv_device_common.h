// fv_device_common.h — device helpers shared by the full-state and the FLASH-BS kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstdint>

#include "fv_layout.h"

#define FV_NEG_INF (-__builtin_huge_valf())

namespace fvk {

// Monotone map float -> int (and back): adjacent floats are adjacent integers.
__device__ __forceinline__ int f2ord(float f)
{
    int i = __float_as_int(f);
    return i >= 0 ? i : (int)(0x80000000u - (unsigned)i);
}
__device__ __forceinline__ float ord2f(int o)
{
    if (o < -0x7F800000) o = -0x7F800000;          // clamp at -inf
    return __int_as_float(o >= 0 ? o : (int)(0x80000000u - (unsigned)o));
}

// (value desc, index asc): the order in which the reference's ascending strict-'>' scan ranks candidates.
__device__ __forceinline__ bool better(float v1, int k1, float v2, int k2)
{
    return v1 > v2 || (v1 == v2 && k1 < k2);
}

// Threshold of the filter kernels that keep tmp inside the per-cell value (NB <= 2, the sparse walk, beam_step_q16):
// y = fl(s + Lq) and ktmp = fl(s + L) share s = fl(tmp + T1[k]), so for the true winner a and the cell b holding
// the filter maximum M:  y_b - y_a <= 2*dmax + (ulp(y_a) + ulp(ktmp_a) + ulp(y_b) + ulp(ktmp_b)) / 2.  The four
// spacings are NOT all the spacing at M: when the candidates straddle a power of two (every long decode crosses
// -1024, -2048, ...) the ones on the far side have twice the spacing.  U is therefore the float spacing at an
// upper bound of every magnitude in play (|M| + window + 1): the bound is 2*dmax + 2U; half a U more covers the
// rounding of the subtraction below and one float step the rounding of window + 2.5 U.  (A wider margin is not
// free: at T = 4096, where U = 2^-8 dwarfs the table's 2*dmax, 4U + 2 steps made the whole-sequence pass 20 % slower.)
__device__ __forceinline__ float refine_threshold(float M, float window)
{
    const float z = fabsf(M) + window + 1.0f;
    const unsigned int eb = __float_as_uint(z) & 0x7f800000u;
    const float U = eb > (24u << 23) ? __uint_as_float(eb - (23u << 23)) : FLT_MIN;
    return ord2f(f2ord(M - (window + 2.5f * U)) - 1);
}

__device__ __forceinline__ float exact_cell(float tmp, float t1, double logA)
{
    float s = tmp + t1;                       // float add
    return (float)((double)s + logA);         // double add, one rounding to float
}

}  // namespace fvk
