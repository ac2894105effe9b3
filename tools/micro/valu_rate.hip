// Issue rate of the packed 16-bit integer VALU ops against plain f32 VALU ops on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define PK(op, a, b) __builtin_bit_cast(unsigned int, op(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)))

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned int *out, int iters, unsigned int seed)
{
    unsigned int a[8], m1[8], m2[8];
    float f[8], g1[8], g2[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i + 1); m1[i] = ~0u; m2[i] = ~0u; f[i] = (float)a[i]; g1[i] = -1e30f; g2[i] = -1e30f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {            // 4 packed u16 ops
                unsigned int z = PK(__builtin_elementwise_add_sat, a[i], m1[(i + 1) & 7] | 1u);
                m2[i] = PK(__builtin_elementwise_min, m2[i], PK(__builtin_elementwise_max, z, m1[i]));
                m1[i] = PK(__builtin_elementwise_min, m1[i], z);
                a[i] = z;
            } else if (MODE == 1) {     // 4 f32 ops: fma, med3, max (+1 add)
                float y = __builtin_fmaf(f[i], 1.0001f, g1[(i + 1) & 7]);
                g2[i] = __builtin_amdgcn_fmed3f(y, g1[i], g2[i]);
                g1[i] = fmaxf(g1[i], y);
                f[i] = y + 1.0f;
            } else {                    // 4 u32 ops
                unsigned int z = a[i] + (m1[(i + 1) & 7] | 1u);
                m2[i] = min(m2[i], max(z, m1[i]));
                m1[i] = min(m1[i], z);
                a[i] = z;
            }
        }
    }
    unsigned int r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + m1[i] + m2[i] + __float_as_uint(f[i] + g1[i] + g2[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char *name, unsigned int *d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 1024>>>(d, 16, 3);
    hipEventRecord(e0);
    k<MODE><<<256, 1024>>>(d, iters, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 8 * 4;             // VALU instructions per wave
    // 4 waves per SIMD: cycles per instruction per SIMD = time * clock / (insts * 4)
    printf("%-10s %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz (4 waves/SIMD)\n", name, ms, ms * 1e-3 * 2.4e9 / (insts * 4));
}

int main()
{
    unsigned int *d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("pk_u16", d, 20000); run<1>("f32", d, 20000); run<2>("u32", d, 20000);
    run<0>("pk_u16", d, 20000); run<1>("f32", d, 20000); run<2>("u32", d, 20000);
    return 0;
}
