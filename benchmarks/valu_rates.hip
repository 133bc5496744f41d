// valu_rates.hip — issue cost (cycles per wave64 instruction per SIMD) of the integer VALU ops the
// filter could be built from, on gfx950.  One wave per SIMD (256 threads per CU), 8 independent
// accumulators per op so that latency does not matter; cycles from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define ITERS 4096

#define BODY8(ASM) \
    ASM(a0) ASM(a1) ASM(a2) ASM(a3) ASM(a4) ASM(a5) ASM(a6) ASM(a7)

#define KERNEL(NAME, ASMSTR)                                                                      \
__global__ void NAME(uint32_t *out, unsigned long long *cyc, uint32_t s, uint32_t t) {             \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    uint32_t b = s ^ threadIdx.x, c = t + threadIdx.x;                                              \
    unsigned long long t0 = __builtin_readcyclecounter();                                          \
    for (int i = 0; i < ITERS; ++i) {                                                              \
        asm volatile(ASMSTR(0) ASMSTR(1) ASMSTR(2) ASMSTR(3) ASMSTR(4) ASMSTR(5) ASMSTR(6) ASMSTR(7)  \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s)); \
    }                                                                                              \
    unsigned long long t1 = __builtin_readcyclecounter();                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;             \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                        \
}

#define S_XOR(n) "v_xor_b32 %" #n ", %8, %" #n "\n"
#define S_XORS(n) "v_xor_b32 %" #n ", %10, %" #n "\n"
#define S_MIN(n) "v_min_u32 %" #n ", %8, %" #n "\n"
#define S_MIN3(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n"
#define S_ALIGN(n) "v_alignbyte_b32 %" #n ", %8, %" #n ", 1\n"
#define S_MAD24(n) "v_mad_u32_u24 %" #n ", %8, %10, %" #n "\n"
#define S_PERM(n) "v_perm_b32 %" #n ", %8, %" #n ", %9\n"
#define S_PKMIN(n) "v_pk_min_u16 %" #n ", %8, %" #n "\n"
#define S_PKSUB(n) "v_pk_sub_u16 %" #n ", %8, %" #n "\n"
#define S_ANDOR(n) "v_and_or_b32 %" #n ", %8, %9, %" #n "\n"
#define S_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 12\n"
#define S_LSHLADD(n) "v_lshl_add_u32 %" #n ", %8, 5, %" #n "\n"
#define S_XAD(n) "v_xad_u32 %" #n ", %8, %9, %" #n "\n"
#define S_SAD(n) "v_sad_u32 %" #n ", %8, %9, %" #n "\n"
#define S_MSAD(n) "v_msad_u8 %" #n ", %8, %9, %" #n "\n"
#define S_SADU8(n) "v_sad_u8 %" #n ", %8, %9, %" #n "\n"
#define S_DOT4(n) "v_dot4_u32_u8 %" #n ", %8, %9, %" #n "\n"
#define S_ADD(n) "v_add_u32 %" #n ", %8, %" #n "\n"
#define S_SUB(n) "v_sub_u32 %" #n ", %8, %" #n "\n"
#define S_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define S_CMP(n) "v_cmp_eq_u32 vcc, %8, %" #n "\n"
#define S_CMPS(n) "v_cmp_eq_u32 s[20:21], %10, %" #n "\n"
#define S_FMA(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define S_PKFMA(n) "v_mul_u32_u24 %" #n ", %8, %" #n "\n"
#define S_MULLO(n) "v_mul_lo_u32 %" #n ", %8, %" #n "\n"
#define S_CNDMASK(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define S_LSHR(n) "v_lshrrev_b32 %" #n ", %8, %" #n "\n"
#define S_MAX3(n) "v_max3_u32 %" #n ", %" #n ", %8, %9\n"
#define S_PKMAD(n) "v_pk_mad_u16 %" #n ", %8, %9, %" #n "\n"
#define S_MADU16(n) "v_mad_u32_u16 %" #n ", %8, %9, %" #n "\n"
#define S_XOR64(n) "v_xor_b32_e64 %" #n ", %8, %" #n "\n"
#define S_XORSDWA(n) "v_xor_b32_sdwa %" #n ", %8, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define S_MINU16(n) "v_min_u16 %" #n ", %8, %" #n "\n"
#define S_XORLIT(n) "v_xor_b32 %" #n ", 0x12345678, %" #n "\n"
#define S_PKADD(n) "v_pk_add_u16 %" #n ", %8, %" #n "\n"
#define S_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %8, %" #n "\n"
#define S_MED3(n) "v_med3_u32 %" #n ", %" #n ", %8, %9\n"
#define S_AND(n) "v_and_b32 %" #n ", %8, %" #n "\n"
#define S_ANDLIT(n) "v_and_b32 %" #n ", 0x7c, %" #n "\n"
#define S_OR(n) "v_or_b32 %" #n ", %8, %" #n "\n"
#define S_LSHL(n) "v_lshlrev_b32 %" #n ", %8, %" #n "\n"
#define S_LSHRI(n) "v_lshrrev_b32 %" #n ", 25, %" #n "\n"
#define S_ADDU16(n) "v_add_u16 %" #n ", %8, %" #n "\n"
#define S_MAXU16(n) "v_max_u16 %" #n ", %8, %" #n "\n"
#define S_MULLOU16(n) "v_mul_lo_u16 %" #n ", %8, %" #n "\n"
#define S_MINF32(n) "v_min_f32 %" #n ", %8, %" #n "\n"
#define S_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define S_BFI(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n"
#define S_MINI16(n) "v_min_i16 %" #n ", %8, %" #n "\n"
#define S_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define S_LSHLOR(n) "v_lshl_or_b32 %" #n ", %8, 4, %" #n "\n"
#define S_MUL24V(n) "v_mul_u32_u24 %" #n ", %8, %" #n "\n"
#define S_MAD24V(n) "v_mad_u32_u24 %" #n ", %8, %9, %" #n "\n"
#define S_DOT4V(n) "v_dot4_u32_u8 %" #n ", %8, %9, %" #n "\n"
#define S_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define S_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %10, %" #n "\n"

KERNEL(k_xor, S_XOR) KERNEL(k_xor_sgpr, S_XORS) KERNEL(k_min, S_MIN) KERNEL(k_min3, S_MIN3) KERNEL(k_align, S_ALIGN)
KERNEL(k_mad24, S_MAD24) KERNEL(k_perm, S_PERM) KERNEL(k_pkmin, S_PKMIN) KERNEL(k_pksub, S_PKSUB) KERNEL(k_andor, S_ANDOR)
KERNEL(k_bfe, S_BFE) KERNEL(k_lshladd, S_LSHLADD) KERNEL(k_xad, S_XAD) KERNEL(k_sad, S_SAD) KERNEL(k_msad, S_MSAD)
KERNEL(k_sadu8, S_SADU8) KERNEL(k_dot4, S_DOT4) KERNEL(k_add, S_ADD) KERNEL(k_sub, S_SUB) KERNEL(k_or3, S_OR3)
KERNEL(k_fma, S_FMA) KERNEL(k_mul24, S_PKFMA) KERNEL(k_mullo, S_MULLO) KERNEL(k_lshr, S_LSHR) KERNEL(k_max3, S_MAX3) KERNEL(k_med3, S_MED3)
KERNEL(k_pkmad, S_PKMAD) KERNEL(k_madu16, S_MADU16) KERNEL(k_xor64, S_XOR64) KERNEL(k_xorsdwa, S_XORSDWA) KERNEL(k_minu16, S_MINU16)
KERNEL(k_xorlit, S_XORLIT) KERNEL(k_pkadd, S_PKADD) KERNEL(k_pkmul, S_PKMUL)
KERNEL(k_and, S_AND) KERNEL(k_andlit, S_ANDLIT) KERNEL(k_or, S_OR) KERNEL(k_lshl, S_LSHL) KERNEL(k_lshri, S_LSHRI) KERNEL(k_addu16, S_ADDU16)
KERNEL(k_maxu16, S_MAXU16) KERNEL(k_mullou16, S_MULLOU16) KERNEL(k_minf32, S_MINF32) KERNEL(k_mov, S_MOV) KERNEL(k_bfi, S_BFI)
KERNEL(k_mini16, S_MINI16) KERNEL(k_add3, S_ADD3) KERNEL(k_lshlor, S_LSHLOR) KERNEL(k_mad24v, S_MAD24V) KERNEL(k_fmac, S_FMAC) KERNEL(k_mbcnt, S_MBCNT)

// ds_read_b32 at aligned / unaligned byte addresses inside a 160-byte table: correctness of the unaligned
// form (does the hardware return bytes a .. a+3?) and its issue cost.
template <int UNALIGNED>
__global__ void k_ldsread(uint32_t *out, unsigned long long *cyc, uint32_t s, uint32_t t) {
    __shared__ uint8_t tab[256];
    tab[threadIdx.x & 255u] = (uint8_t)((threadIdx.x * 7u + 3u) & 255u);
    __syncthreads();
    uint32_t a0 = (threadIdx.x * 13u + s) & 127u, acc = 0, bad = 0;
    if (!UNALIGNED) a0 &= ~3u;
    // correctness: every address 0..131
    for (uint32_t a = threadIdx.x & 63u; a < 132u; a += 64u) {
        uint32_t got;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(a));
        uint32_t exp = 0;
        for (int b = 0; b < 4; ++b) exp |= (uint32_t)(uint8_t)(((a + b) * 7u + 3u) & 255u) << (8 * b);
        if (UNALIGNED || (a & 3u) == 0) bad += got != exp;
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
        asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:12\n\t"
                     "ds_read_b32 %4, %8 offset:16\n\tds_read_b32 %5, %8 offset:20\n\tds_read_b32 %6, %8 offset:24\n\tds_read_b32 %7, %8 offset:28\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a0));
        acc ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
        a0 = (a0 + (UNALIGNED ? 5u : 4u)) & 127u;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (bad << 24);
    if (bad) atomicAdd(&cyc[1], (unsigned long long)bad);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// the filter's per-offset mix: 3 xor + min3 + min (+ mad + alignbyte), all independent chains
__global__ void k_mix(uint32_t *out, unsigned long long *cyc, uint32_t s, uint32_t t) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = s ^ threadIdx.x, c = t + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(S_XORS(0) S_XORS(1) S_XORS(2) S_MIN3(3) S_MIN(4) S_MAD24(5) S_ALIGN(6) S_XORS(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// the packed-16 mix: perm + pk_mad + xor + 3 xor + 3 pk_min (two offsets)
__global__ void k_mixpk(uint32_t *out, unsigned long long *cyc, uint32_t s, uint32_t t) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = s ^ threadIdx.x, c = t + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(S_PERM(0) S_PKMAD(1) S_XORS(2) S_XORS(3) S_PKMIN(4) S_XORS(5) S_PKMIN(6) S_XORS(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void k_cmp(uint32_t *out, unsigned long long *cyc, uint32_t s, uint32_t t) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = s ^ threadIdx.x, c = t + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(S_CMP(0) S_CMP(1) S_CMP(2) S_CMP(3) S_CMP(4) S_CMP(5) S_CMP(6) S_CMP(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s) : "vcc");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_cmp_sgpr(uint32_t *out, unsigned long long *cyc, uint32_t s, uint32_t t) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = s ^ threadIdx.x, c = t + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile(S_CMPS(0) S_CMPS(1) S_CMPS(2) S_CMPS(3) S_CMPS(4) S_CMPS(5) S_CMPS(6) S_CMPS(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(s) : "s20", "s21");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    uint32_t *out; unsigned long long *cyc;
    CHECK(hipMalloc((void **)&out, 256 * 1024 * 4)); CHECK(hipMalloc((void **)&cyc, 16));
#define RUN(K, WAVES) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); \
    hipLaunchKernelGGL(K, dim3(256), dim3(64 * (WAVES)), 0, 0, out, cyc, 12345u, 678u); \
    hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(256), dim3(64 * (WAVES)), 0, 0, out, cyc, 12345u, 678u); hipEventRecord(e1); \
    CHECK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long h; CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); \
    printf("%-14s waves/SIMD=%d: %6.2f ticks/instr/wave; kernel %.1f us -> %.2f ns per instr per wave -> %.2f cycles@2.4GHz per instr per SIMD\n", #K, (WAVES) / 4, (double)h / (ITERS * 8.0), ms * 1e3, ms * 1e6 / (ITERS * 8.0), ms * 1e6 / (ITERS * 8.0) * 2.4 / ((WAVES) / 4)); }
    RUN(k_xor, 16)
    RUN(k_xor_sgpr, 16)
    RUN(k_min, 16)
    RUN(k_min3, 16)
    RUN(k_max3, 16)
    RUN(k_med3, 16)
    RUN(k_align, 16)
    RUN(k_mad24, 16)
    RUN(k_mul24, 16)
    RUN(k_mullo, 16)
    RUN(k_perm, 16)
    RUN(k_pkmin, 16)
    RUN(k_pksub, 16)
    RUN(k_andor, 16)
    RUN(k_bfe, 16)
    RUN(k_lshladd, 16)
    RUN(k_lshr, 16)
    RUN(k_xad, 16)
    RUN(k_sad, 16)
    RUN(k_msad, 16)
    RUN(k_sadu8, 16)
    RUN(k_dot4, 16)
    RUN(k_add, 16)
    RUN(k_sub, 16)
    RUN(k_or3, 16)
    RUN(k_cmp, 16)
    RUN(k_cmp_sgpr, 16)
    RUN(k_fma, 16)
    RUN(k_pkmad, 16)
    RUN(k_madu16, 16)
    RUN(k_xor64, 16)
    RUN(k_xorsdwa, 16)
    RUN(k_minu16, 16)
    RUN(k_xorlit, 16)
    RUN(k_pkadd, 16)
    RUN(k_pkmul, 16)
    RUN(k_and, 16) RUN(k_andlit, 16) RUN(k_or, 16) RUN(k_lshl, 16) RUN(k_lshri, 16) RUN(k_addu16, 16) RUN(k_maxu16, 16) RUN(k_mullou16, 16)
    RUN(k_minf32, 16) RUN(k_mov, 16) RUN(k_bfi, 16) RUN(k_mini16, 16) RUN(k_add3, 16) RUN(k_lshlor, 16) RUN(k_mad24v, 16) RUN(k_fmac, 16) RUN(k_mbcnt, 16)
    CHECK(hipMemset(cyc, 0, 16));
    RUN(k_ldsread<0>, 16)
    { unsigned long long b; CHECK(hipMemcpy(&b, cyc + 1, 8, hipMemcpyDeviceToHost)); printf("aligned ds_read_b32: %llu wrong values\n", b); }
    CHECK(hipMemset(cyc, 0, 16));
    RUN(k_ldsread<1>, 16)
    { unsigned long long b; CHECK(hipMemcpy(&b, cyc + 1, 8, hipMemcpyDeviceToHost)); printf("UNALIGNED ds_read_b32: %llu wrong values\n", b); }
    RUN(k_mix, 16)
    RUN(k_mixpk, 16)
    RUN(k_xor, 8) RUN(k_xor, 4) RUN(k_add, 8) RUN(k_min, 8) RUN(k_mix, 8)
    return 0;
}
