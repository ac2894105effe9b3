// Does a second (third, ...) stream that has carried work make every dispatch on the first stream slower?
// Chain of small same-stream launches on stream A, timed: alone; after stream B ran one kernel; after B .. D did;
// after those streams were destroyed.   hipcc --offload-arch=gfx950 -O3 -o queue_penalty.bin queue_penalty.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct Small { int *p; int n; };
__global__ void k(const Small a) { if (a.n == 12345) a.p[blockIdx.x] = 1; }
__global__ void busy(int *p, int iters) { int x = threadIdx.x; for (int i = 0; i < iters; ++i) x = x * 1664525 + 1013904223; if (x == 42) p[0] = x; }

static float chain(hipStream_t s, int *d, int grid, int block)
{
    Small a{ d, 1 };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) k<<<grid, block, 0, s>>>(a);
    hipStreamSynchronize(s);
    const int N = 2000;
    hipEventRecord(e0, s);
    for (int i = 0; i < N; ++i) k<<<grid, block, 0, s>>>(a);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 1e3f * ms / N;
}

int main()
{
    int *d; hipMalloc(&d, 1 << 20);
    hipStream_t A; hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
    printf("A alone:                         %.2f us per launch (grid 248 x 1024), %.2f (grid 1 x 1024)\n", chain(A, d, 248, 1024), chain(A, d, 1, 1024));
    hipStream_t B[3];
    for (int q = 0; q < 3; ++q) hipStreamCreateWithFlags(&B[q], hipStreamNonBlocking);
    printf("3 more streams created, unused:  %.2f, %.2f\n", chain(A, d, 248, 1024), chain(A, d, 1, 1024));
    busy<<<1, 64, 0, B[0]>>>(d, 1000); hipStreamSynchronize(B[0]);
    printf("after stream B ran one kernel:   %.2f, %.2f\n", chain(A, d, 248, 1024), chain(A, d, 1, 1024));
    for (int q = 1; q < 3; ++q) { busy<<<1, 64, 0, B[q]>>>(d, 1000); hipStreamSynchronize(B[q]); }
    printf("after B, C, D ran one kernel:    %.2f, %.2f\n", chain(A, d, 248, 1024), chain(A, d, 1, 1024));
    for (int q = 0; q < 3; ++q) hipStreamDestroy(B[q]);
    printf("after B, C, D were destroyed:    %.2f, %.2f\n", chain(A, d, 248, 1024), chain(A, d, 1, 1024));
    hipStream_t A2; hipStreamCreateWithFlags(&A2, hipStreamNonBlocking);
    printf("a fresh stream:                  %.2f, %.2f\n", chain(A2, d, 248, 1024), chain(A2, d, 1, 1024));
    return 0;
}
