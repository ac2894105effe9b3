// Why does beam_step's gather (B random rows of a K x K float64 table, 512-byte runs) take 2-3x as long as a stream of the
// same bytes at cfg4's size (33.5 MB per launch)?  Same kernel body, different placements of the rows:
//   random      B rows drawn from the whole table (2.1 GB at K = 16384): what beam_step does
//   consecutive B neighbouring rows (one contiguous 33.5 MB block)
//   window      B random rows out of a window of 2 B rows (67 MB)
//   panel       random rows, table stored panel-major ([K/64][K][64]): a workgroup's 512-byte runs all lie in one 8 MB panel
//   random2x    random rows, grid of 2 x K/64 workgroups of 512 threads (8 waves: entries w, w + 8, ...)
// usage: gather_tlb.bin [K=16384] [B=256]
//   hipcc --offload-arch=gfx950 -O3 -o gather_tlb.bin gather_tlb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void rows8(const double *tab, const int *rows, int B, size_t row_stride, size_t panel_stride, unsigned int *sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double *Lc = tab + (size_t)blockIdx.x * panel_stride + lane;
    double acc = 0.0;
    for (int s0 = w; s0 < B; s0 += WAVES * 16) {
        double L[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int s = s0 + WAVES * u; L[u] = s < B ? Lc[(size_t)rows[s] * row_stride] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += L[u];
    }
    if (acc == 1.2345) sink[0] = 1;
}

__global__ __launch_bounds__(1024) void stream16(const uint4 *p, size_t n16, unsigned int *sink)
{
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 16384, B = argc > 2 ? atoi(argv[2]) : 256;
    const int REPS = 8;
    double *t8; int *rows; unsigned int *sink;
    hipMalloc(&t8, (size_t)K * K * 8); hipMalloc(&rows, (size_t)B * 4 * REPS * 4); hipMalloc(&sink, 64);
    hipMemset(t8, 0, (size_t)K * K * 8);
    std::mt19937 rng(7);
    std::vector<int> perm(K); std::iota(perm.begin(), perm.end(), 0); std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int> h((size_t)B * REPS * 4);
    for (int r = 0; r < REPS; ++r) {
        for (int s = 0; s < B; ++s) h[(size_t)(0 * REPS + r) * B + s] = perm[(size_t)r * B + s];                   // random
        for (int s = 0; s < B; ++s) h[(size_t)(1 * REPS + r) * B + s] = (r * 2 * B + s) % K;                       // consecutive
        std::vector<int> win(2 * B); std::iota(win.begin(), win.end(), 0); std::shuffle(win.begin(), win.end(), rng);
        for (int s = 0; s < B; ++s) h[(size_t)(2 * REPS + r) * B + s] = (r * 2 * B + win[s]) % K;                  // window
        for (int s = 0; s < B; ++s) h[(size_t)(3 * REPS + r) * B + s] = perm[(size_t)((r + REPS) * B + s) % K];    // random, other rows
    }
    hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const double bytes = (double)B * K * 8;
    printf("K=%d B=%d: %.1f MB per launch\n", K, B, bytes / 1e6);
    auto report = [&](const char *name, int rep) { hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                                                   printf("%-12s rep %d: %6.1f us = %.2f TB/s\n", name, rep, 1e3 * ms, bytes / (ms * 1e9)); };
    for (int rep = 0; rep < 4; ++rep) { hipEventRecord(e0); stream16<<<1024, 1024>>>(reinterpret_cast<const uint4 *>(t8) + (size_t)rep * (size_t)(bytes / 16), (size_t)(bytes / 16), sink); report("stream16", rep); }
    for (int rep = 0; rep < REPS; ++rep) { hipEventRecord(e0); rows8<16><<<K / 64, 1024>>>(t8, rows + (size_t)(0 * REPS + rep) * B, B, (size_t)K, 64, sink); report("random", rep); }
    for (int rep = 0; rep < REPS; ++rep) { hipEventRecord(e0); rows8<16><<<K / 64, 1024>>>(t8, rows + (size_t)(1 * REPS + rep) * B, B, (size_t)K, 64, sink); report("consecutive", rep); }
    for (int rep = 0; rep < REPS; ++rep) { hipEventRecord(e0); rows8<16><<<K / 64, 1024>>>(t8, rows + (size_t)(2 * REPS + rep) * B, B, (size_t)K, 64, sink); report("window", rep); }
    for (int rep = 0; rep < REPS; ++rep) { hipEventRecord(e0); rows8<16><<<K / 64, 1024>>>(t8, rows + (size_t)(3 * REPS + rep) * B, B, 64, (size_t)K * 64, sink); report("panel", rep); }
    for (int rep = 0; rep < REPS; ++rep) { hipEventRecord(e0); rows8<8><<<K / 64, 512>>>(t8, rows + (size_t)(0 * REPS + rep) * B, B, (size_t)K, 64, sink); report("random 8w", rep); }
    for (int rep = 0; rep < REPS; ++rep) { hipEventRecord(e0); rows8<4><<<K / 64, 256>>>(t8, rows + (size_t)(3 * REPS + rep) * B, B, (size_t)K, 64, sink); report("random 4w", rep); }
    return 0;
}
