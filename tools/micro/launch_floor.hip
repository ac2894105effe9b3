// Per-launch cost of a chain of dependent (same-stream) empty-ish launches, by workgroup size, dynamic LDS size and
// kernel-argument size.   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
struct Small { int *p; int n; };
struct Big { int *p; int n; long long pad[70]; };        // ~580 bytes, like StepArgs<8>

template <typename A> __global__ void k(const A a)
{
    extern __shared__ int sm[];
    if (a.n == 12345) { sm[threadIdx.x] = a.n; a.p[blockIdx.x] = sm[(threadIdx.x + 1) & 63]; }   // never taken: keeps args and LDS alive
}

template <typename A> void run(const char *name, int grid, int block, size_t lds, int *d)
{
    A a{}; a.p = d; a.n = 1;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<A>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) k<A><<<grid, block, lds>>>(a);
    hipDeviceSynchronize();
    const int N = 2000;
    hipEventRecord(e0);
    for (int i = 0; i < N; ++i) k<A><<<grid, block, lds>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s grid %4d block %4d lds %6zu B: %.2f us per launch\n", name, grid, block, lds, 1e3 * ms / N);
}

int main()
{
    int *d; hipMalloc(&d, 1 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        run<Small>("small args", 248, 256, 0, d);
        run<Small>("small args", 248, 512, 0, d);
        run<Small>("small args", 248, 1024, 0, d);
        run<Small>("small args", 248, 1024, 20 * 1024, d);
        run<Small>("small args", 248, 1024, 70 * 1024, d);
        run<Small>("small args", 248, 1024, 150 * 1024, d);
        run<Small>("small args", 248, 512, 70 * 1024, d);
        run<Small>("small args", 248, 512, 150 * 1024, d);
        run<Big>("580-byte args", 248, 1024, 150 * 1024, d);
        run<Big>("580-byte args", 248, 512, 70 * 1024, d);
        run<Small>("small args, 124 WGs", 124, 1024, 150 * 1024, d);
        run<Small>("small args, 496 WGs", 496, 512, 70 * 1024, d);
    }
    return 0;
}
