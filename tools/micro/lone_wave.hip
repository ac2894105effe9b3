// What ONE wave64 pays per instruction on gfx950 when nothing else runs on its SIMD: dependent and independent VALU
// chains, compare -> SGPR mask -> select chains, LDS pointer chases (b32 / b64 / b128), write-then-read, lane
// broadcasts.  These are the costs that bound the exact heap replay of FLASH-BS (one consumer wave).
// hipcc --offload-arch=gfx950 -O3 -o lone_wave.bin lone_wave.hip && ./lone_wave.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned int *out, int iters, unsigned long long *cyc)
{
    __shared__ unsigned int lds[4096];
    const int lane = threadIdx.x;
    // pointer-chase ring: entry i (16 bytes) holds the byte address of entry (i*5+3) % 256 in all four words
    for (int i = lane; i < 256; i += 64) {
        const unsigned int nxt = (unsigned int)(((i * 5 + 3) & 255) * 16);
        lds[i * 4 + 0] = nxt; lds[i * 4 + 1] = nxt; lds[i * 4 + 2] = nxt; lds[i * 4 + 3] = nxt;
    }
    __syncthreads();
    float a = (float)lane, b = 1.0f, c = 2.0f, d = 3.0f;
    unsigned int p = (unsigned int)lane * 16u, q = (unsigned int)(lane + 64) * 16u;
    const unsigned int base = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) unsigned int *)lds;
    p += base; q += base;
    // make the ring hold absolute LDS addresses
    for (int i = lane; i < 1024; i += 64) lds[i] += base;
    __syncthreads();
    unsigned int sl = 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            __asm__ volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        } else if (MODE == 1) {
            __asm__ volatile(REP64("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t") : "+v"(a), "+v"(c) : "v"(b));
        } else if (MODE == 2) {
            __asm__ volatile(REP64("v_cmp_gt_f32_e64 s[10:11], %0, %1\n\tv_cndmask_b32_e64 %0, %2, %0, s[10:11]\n\t")
                             : "+v"(a) : "v"(b), "v"(c) : "s10", "s11");
        } else if (MODE == 3) {
            __asm__ volatile(REP64("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(p) : : "memory");
        } else if (MODE == 4) {
            __asm__ volatile(REP64("ds_read_b64 v[20:21], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v20\n\t") : "+v"(p) : : "memory", "v20", "v21");
        } else if (MODE == 5) {
            __asm__ volatile(REP64("ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v20\n\t") : "+v"(p) : : "memory", "v20", "v21", "v22", "v23");
        } else if (MODE == 6) {
            // write-then-read: the read waits for nothing but its own data (same-wave DS ops are in order)
            __asm__ volatile(REP64("ds_write_b32 %1, %0\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(p) : "v"(q) : "memory");
        } else if (MODE == 7) {
            // lane broadcast chain: readlane -> SGPR -> writelane -> readlane ...
            __asm__ volatile(REP64("v_readlane_b32 s10, %0, 5\n\ts_nop 3\n\tv_writelane_b32 %0, s10, 7\n\t") : "+v"(a) : "s"(sl) : "s10");
        } else if (MODE == 8) {
            // ballot -> ff1 -> readlane (the candidate pick of the replay)
            __asm__ volatile(REP16("v_cmp_gt_f32_e64 s[10:11], %0, %1\n\ts_ff1_i32_b64 s12, s[10:11]\n\ts_nop 3\n\tv_readlane_b32 s13, %0, s12\n\tv_add_f32 %0, s13, %0\n\t")
                             : "+v"(a) : "v"(b) : "s10", "s11", "s12", "s13");
        } else if (MODE == 9) {
            // two independent LDS chases in flight
            __asm__ volatile(REP64("ds_read_b32 %0, %0\n\tds_read_b32 %1, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(p), "+v"(q) : : "memory");
        } else if (MODE == 10) {
            // SALU dependent chain
            __asm__ volatile(REP64("s_add_u32 %0, %0, 3\n\t") : "+s"(sl) : : "scc");
        } else if (MODE == 11) {
            // DPP row shift chain
            __asm__ volatile(REP64("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a));
        } else if (MODE == 12) {
            // ds_bpermute chain
            unsigned int idx = (unsigned int)((lane + 1) & 63) * 4u;
            __asm__ volatile(REP64("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(a) : "v"(idx) : "memory");
        } else if (MODE == 13) {
            // v_cmp -> VCC -> v_cndmask (VOP2 form)
            __asm__ volatile(REP64("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %2, %0, vcc\n\t") : "+v"(a) : "v"(b), "v"(c) : "vcc");
        } else if (MODE == 15) {
            // 58 lanes write the SAME 8 bytes (the replay's idle lanes on slot 0), then a dependent read
            unsigned int w = lane < 6 ? q : base;
            __asm__ volatile(REP64("ds_write2_b32 %1, %2, %2 offset1:1\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(p) : "v"(w), "v"(q) : "memory");
        } else if (MODE == 16) {
            // every lane writes its own 8 bytes, then a dependent read
            __asm__ volatile(REP64("ds_write2_b32 %1, %2, %2 offset1:1\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(p) : "v"(q), "v"(q) : "memory");
        } else if (MODE == 17) {
            // 58 lanes read the same 16 bytes
            unsigned int r = lane < 6 ? p : base;
            __asm__ volatile(REP64("ds_read_b128 v[20:23], %0\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v20\n\t") : "+v"(r) : : "memory", "v20", "v21", "v22", "v23");
            p += r;
        } else if (MODE == 14) {
            // v_min3 / v_med3 dependent
            __asm__ volatile(REP64("v_min3_f32 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[0] = t1 - t0;
    out[lane] = __float_as_uint(a + c + d) + p + q + sl;
}

template <int MODE> void run(const char *name, unsigned int *d, unsigned long long *c, int per_iter)
{
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<1, 64>>>(d, 10, c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<1, 64>>>(d, iters, c);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("%-34s %8.2f ns per group  (%6.1f s_memtime ticks; %6.1f cycles at 2.4 GHz)\n", name, ms * 1e6 / n, (double)cy / n, ms * 1e6 / n * 2.4);
    fflush(stdout);
}

int main()
{
    unsigned int *d; unsigned long long *c;
    hipMalloc(&d, 4096); hipMalloc(&c, 64);
    printf("start\n"); fflush(stdout);
    run<0>("v_add dependent", d, c, 64);
    run<1>("2 independent v_add (pair)", d, c, 64);
    run<14>("v_min3 dependent", d, c, 64);
    run<2>("v_cmp_e64->sgpr + v_cndmask (pair)", d, c, 64);
    run<13>("v_cmp->vcc + v_cndmask (pair)", d, c, 64);
    run<3>("ds_read_b32 chase", d, c, 64);
    run<4>("ds_read_b64 chase (+v_mov)", d, c, 64);
    run<5>("ds_read_b128 chase (+v_mov)", d, c, 64);
    run<6>("ds_write_b32 + ds_read_b32 chase", d, c, 64);
    run<9>("2 LDS chases in flight (pair)", d, c, 64);
    run<7>("readlane + nop3 + writelane", d, c, 64);
    run<8>("cmp+ff1+nop3+readlane+add", d, c, 16);
    run<15>("write2 (58 lanes same addr) + read", d, c, 64);
    run<16>("write2 (distinct) + read", d, c, 64);
    run<17>("ds_read_b128, 58 lanes same addr", d, c, 64);
    run<10>("s_add dependent", d, c, 64);
    run<11>("dpp row_shr dependent", d, c, 64);
    run<12>("ds_bpermute chase", d, c, 64);
    return 0;
}
