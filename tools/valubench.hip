// valubench.hip -- development microbenchmark: issue rate of the packed-16-bit integer ops the key scan uses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define ITER 4096
#define BODY(OPS) \
    unsigned a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
    unsigned b = seed; \
    for (int i = 0; i < ITER; i++) { OPS(a0) OPS(a1) OPS(a2) OPS(a3) OPS(a4) OPS(a5) OPS(a6) OPS(a7) } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
#define OP_PKMUL(a) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_PKMAD(a) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define OP_PKASHR(a) asm volatile("v_pk_ashrrev_i16 %0, 3, %0 op_sel_hi:[0,1]" : "+v"(a));
#define OP_PKADD(a) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_PKMAX(a) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_AND(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_ADD(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MUL24(a) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a) : "v"(b));
#define OP_MAD24(a) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define OP_DOT4(a) asm volatile("v_dot4_i32_i8 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define OP_FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define OP_CVTUB(a) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a));
#define OP_LSHL(a) asm volatile("v_lshlrev_b32 %0, 8, %0" : "+v"(a));
#define OP_BFE(a) asm volatile("v_bfe_i32 %0, %0, 8, 8" : "+v"(a));
#define OP_PERM(a) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define OP_SAD(a) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define OP_DPP(a) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));
#define OP_TRUNC(a) asm volatile("v_trunc_f32 %0, %0" : "+v"(a));
#define OP_PKFMA32(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define K(name, OP) __global__ void __launch_bounds__(256) name(unsigned *out, unsigned seed) { BODY(OP) }
K(k_pkmul, OP_PKMUL) K(k_pkmad, OP_PKMAD) K(k_pkashr, OP_PKASHR) K(k_pkadd, OP_PKADD) K(k_pkmax, OP_PKMAX) K(k_and, OP_AND)
K(k_add, OP_ADD) K(k_mul24, OP_MUL24) K(k_mad24, OP_MAD24) K(k_dot4, OP_DOT4) K(k_fma, OP_FMA) K(k_cvtub, OP_CVTUB)
K(k_lshl, OP_LSHL) K(k_bfe, OP_BFE) K(k_perm, OP_PERM) K(k_sad, OP_SAD) K(k_dpp, OP_DPP) K(k_trunc, OP_TRUNC)
int main()
{
    unsigned *out; const int blocks = 256 * 8, threads = 256;   // 8 blocks/CU = 8 waves/SIMD
    CK(hipMalloc(&out, blocks * threads * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
#define RUN(kern) { kern<<<blocks, threads>>>(out, 3); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); for (int r = 0; r < 5; r++) kern<<<blocks, threads>>>(out, 3); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5; \
        double winst = (double)blocks * (threads / 64) * ITER * 8; /* wave-instructions */ \
        printf("%-10s %.3f ms  %.2f Tlane-op/s  -> %.2f cycles per wave-instr per SIMD @2.4GHz\n", #kern, ms, winst * 64 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / winst); }
    RUN(k_add) RUN(k_and) RUN(k_lshl) RUN(k_pkadd) RUN(k_pkashr) RUN(k_pkmax) RUN(k_pkmul) RUN(k_pkmad) RUN(k_mul24) RUN(k_mad24)
    RUN(k_dot4) RUN(k_fma) RUN(k_cvtub) RUN(k_bfe) RUN(k_perm) RUN(k_sad) RUN(k_dpp) RUN(k_trunc)
    return 0;
}
