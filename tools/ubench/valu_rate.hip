// Runs ON THE GPU BOX: issue rate of a few VALU instructions the scan loops use, relative to v_add_u32 (dependent chains of 8
// independent accumulators per lane, 1 024 workgroups x 256 threads, 4 096 iterations).   hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8], b = seed + threadIdx.x, c = seed * 3 + 1;
    for (int i = 0; i < 8; i++) a[i] = seed + i * 7 + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#define STEP(i)                                                                                              \
        if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                           \
        if (OP == 1) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                \
        if (OP == 2) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));              \
        if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));              \
        if (OP == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c)); \
        if (OP == 5) asm volatile("v_pk_mad_i16 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(b), "v"(c));      \
        if (OP == 6) asm volatile("v_pk_mad_u16 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(b), "v"(c));      \
        if (OP == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                        \
        if (OP == 8) asm volatile("v_pk_lshrrev_b16 %0, 3, %0" : "+v"(a[i]));                              \
        if (OP == 9) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));               \
        if (OP == 10) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));             \
        if (OP == 11) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        REP8(STEP) REP8(STEP) REP8(STEP) REP8(STEP)
#undef STEP
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> float run(uint32_t *d, const char *name, float base)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<2048, 256>>>(d, 1, 64);
    hipEventRecord(e0);
    k<OP><<<2048, 256>>>(d, 1, 4096);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s %8.3f ms  x%.2f of v_add_u32\n", name, ms, base > 0 ? ms / base : 1.0f);
    return ms;
}
int main()
{
    uint32_t *d; hipMalloc(&d, 2048 * 256 * 4);
    float b = run<0>(d, "v_add_u32", 0);
    run<1>(d, "v_sad_u8", b); run<2>(d, "v_dot4c_i32_i8", b); run<3>(d, "v_perm_b32", b); run<4>(d, "v_bitop3_b32", b);
    run<5>(d, "v_pk_mad_i16 clamp", b); run<6>(d, "v_pk_mad_u16 clamp", b); run<7>(d, "v_mul_lo_u32", b); run<8>(d, "v_pk_lshrrev_b16", b);
    run<9>(d, "v_bfi_b32", b); run<10>(d, "v_add3_u32", b); run<11>(d, "v_and_or_b32", b);
    return 0;
}
