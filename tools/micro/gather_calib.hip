// Calibration of rocprofv3's FETCH_SIZE and TCC_MISS_sum for the access patterns of the FLASH-BS step kernels (VERDICT r2
// item 5; MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Each kernel reads an exactly known number of bytes, once, from a table far larger than L2 + the
// Infinity Cache, in the pattern named:
//   stream16   16 B per lane, fully coalesced (the guide's calibrated case: FETCH_SIZE x 2)
//   rows8      beam_step: a workgroup = 64 columns; wave w reads rows w, w+16, ... of B randomly chosen rows, each wave-wide
//              load one contiguous 512-byte run (64 lanes x 8 B), 16 loads in flight per lane
//   rows4      beam_step_q16: a workgroup = 128 columns of a 2-byte table; each wave-wide load one contiguous 256-byte run
//              (64 lanes x 4 B), 16 in flight
// usage: gather_calib.bin [K=16384] [B=256]      (run once per counter under rocprofv3 --pmc ...; tools/gather_calib.sh)
//   hipcc --offload-arch=gfx950 -O3 -o gather_calib.bin gather_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ __launch_bounds__(1024) void stream16(const uint4 *p, size_t n16, unsigned int *sink)
{
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void rows8(const double *tab, const int *rows, int B, size_t ld, unsigned int *sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double *Lc = tab + (size_t)blockIdx.x * 64 + lane;
    double acc = 0.0;
    for (int s0 = w; s0 < B; s0 += 16 * 16) {
        double L[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int s = s0 + 16 * u; L[u] = s < B ? Lc[(size_t)rows[s] * ld] : 0.0; }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += L[u];
    }
    if (acc == 1.2345) sink[0] = 1;
}

__global__ __launch_bounds__(1024) void rows4(const unsigned int *tab, const int *rows, int B, size_t pitch, unsigned int *sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned int *Lq = tab + (size_t)blockIdx.x * 64 + lane;
    unsigned int acc = 0;
    for (int s0 = w; s0 < B; s0 += 16 * 16) {
        unsigned int L[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int s = s0 + 16 * u; L[u] = s < B ? Lq[(size_t)rows[s] * pitch] : 0u; }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= L[u];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// rows16x4: a wave-wide load of 16 B per lane, its four 16-lane groups reading four DIFFERENT rows (256-byte run each): the
// same 256-byte row pieces as rows4, a quarter of the load instructions
__global__ __launch_bounds__(1024) void rows16x4(const uint4 *tab, const int *rows, int B, size_t pitch16, unsigned int *sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    const uint4 *Lq = tab + (size_t)blockIdx.x * 16 + c;          // 128 columns of 2 bytes = 16 uint4 per row and panel
    unsigned int acc = 0;
    for (int i0 = 0; w + 16 * (4 * i0 + g) < B + 16 * 4 * 8; i0 += 8) {      // 8 loads in flight; entry = w + 16 * (4 i + g)
        uint4 L[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int s = w + 16 * (4 * (i0 + u) + g); L[u] = s < B ? Lq[(size_t)rows[s] * pitch16] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= L[u].x ^ L[u].y ^ L[u].z ^ L[u].w;
        if (w + 16 * 4 * (i0 + 8) >= B) break;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 16384, B = argc > 2 ? atoi(argv[2]) : 256;
    const size_t ld = (size_t)K;                                   // doubles per row of the 8-byte table; u16 per row of the 2-byte one
    double *t8; unsigned short *t2; uint4 *st; int *rows; unsigned int *sink;
    const size_t stream_bytes = (size_t)B * K * 8;                 // the same byte count as one rows8 launch
    hipMalloc(&t8, ld * K * 8); hipMalloc(&t2, ld * K * 2); hipMalloc(&st, stream_bytes); hipMalloc(&rows, B * 4 * 8); hipMalloc(&sink, 64);
    hipMemset(t8, 0, ld * K * 8); hipMemset(t2, 0, ld * K * 2); hipMemset(st, 0, stream_bytes);
    std::vector<int> perm(K); std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(7); std::shuffle(perm.begin(), perm.end(), rng);
    hipMemcpy(rows, perm.data(), (size_t)B * 4 * 8, hipMemcpyHostToDevice);       // 8 disjoint sets of B rows: every launch reads cold rows
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    printf("K=%d B=%d: bytes per launch: stream16 %zu, rows8 %zu (B*K*8), rows4 %zu (B*K*2)\n", K, B, stream_bytes, (size_t)B * K * 8, (size_t)B * K * 2);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0); stream16<<<1024, 1024>>>(st, stream_bytes / 16, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("stream16 rep %d: %.1f us = %.2f TB/s\n", rep, 1e3 * ms, stream_bytes / (ms * 1e9));
    }
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0); rows8<<<K / 64, 1024>>>(t8, rows + rep * B, B, ld, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("rows8 rep %d: %.1f us = %.2f TB/s\n", rep, 1e3 * ms, (double)B * K * 8 / (ms * 1e9));
    }
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0); rows4<<<K / 128, 1024>>>(reinterpret_cast<const unsigned int *>(t2), rows + rep * B, B, ld / 2, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("rows4 rep %d: %.1f us = %.2f TB/s\n", rep, 1e3 * ms, (double)B * K * 2 / (ms * 1e9));
    }
    for (int rep = 0; rep < 8; ++rep) {
        hipEventRecord(e0); rows16x4<<<K / 128, 1024>>>(reinterpret_cast<const uint4 *>(t2), rows + rep * B, B, ld / 8, sink); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("rows16x4 rep %d: %.1f us = %.2f TB/s\n", rep, 1e3 * ms, (double)B * K * 2 / (ms * 1e9));
    }
    return 0;
}
