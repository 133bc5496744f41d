// valu_rates64.hip — issue cost (cycles per wave64 instruction per SIMD) of the 64-bit / carry VALU ops the bit-vector
// column (fz_device.h: fz_bits_column) is built from, on gfx950.  4 waves per SIMD, 4 independent accumulators per op.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define ITERS 4096
#define KERNEL(NAME, ASM1)                                                                                   \
__global__ void NAME(unsigned long long *out, unsigned long long *cyc, unsigned long long s) {                \
    unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = s ^ threadIdx.x;          \
    unsigned long long t0 = __builtin_readcyclecounter();                                                     \
    for (int i = 0; i < ITERS; ++i) {                                                                         \
        asm volatile(ASM1(0) ASM1(1) ASM1(2) ASM1(3) ASM1(0) ASM1(1) ASM1(2) ASM1(3)                           \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");                               \
    }                                                                                                         \
    unsigned long long t1 = __builtin_readcyclecounter();                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;                                           \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                                   \
}
#define KERNEL32(NAME, ASM1)                                                                                 \
__global__ void NAME(unsigned long long *out, unsigned long long *cyc, unsigned long long s) {                \
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = (uint32_t)s ^ threadIdx.x, c = b * 3u; \
    unsigned long long t0 = __builtin_readcyclecounter();                                                     \
    for (int i = 0; i < ITERS; ++i) {                                                                         \
        asm volatile(ASM1(0) ASM1(1) ASM1(2) ASM1(3) ASM1(0) ASM1(1) ASM1(2) ASM1(3)                           \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");                       \
    }                                                                                                         \
    unsigned long long t1 = __builtin_readcyclecounter();                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;                                           \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                                   \
}
#define S_SHL64(n) "v_lshlrev_b64 %" #n ", 1, %" #n "\n"
#define S_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 1, %4\n"
#define S_ADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %4\n"
#define S_MOV64(n) "v_mov_b64 %" #n ", %4\n"
#define S_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %4, %" #n "\n"
#define S_ADDC(n) "v_addc_co_u32 %" #n ", vcc, %4, %" #n ", vcc\n"
#define S_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %4, %" #n ", 31\n"
#define S_BFI(n) "v_bfi_b32 %" #n ", %4, %" #n ", -1\n"
#define S_NOT(n) "v_not_b32 %" #n ", %" #n "\n"
#define S_ASHR(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define S_ADDX2(n) "v_add_u32 %" #n ", %" #n ", %" #n "\n"
#define S_SDWA(n) "v_lshlrev_b32_sdwa %" #n ", 3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define S_CND(n) "v_cndmask_b32 %" #n ", %4, %" #n ", vcc\n"
#define S_MINU(n) "v_min_u32 %" #n ", %4, %" #n "\n"
#define S_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 1, 1\n"
#define S_MULLO(n) "v_mul_lo_u32 %" #n ", %4, %" #n "\n"
#define S_MAD24(n) "v_mad_u32_u24 %" #n ", %4, %5, %" #n "\n"
KERNEL(k_shl64, S_SHL64) KERNEL(k_lshladd64, S_LSHLADD64) KERNEL(k_add64, S_ADD64) KERNEL(k_mov64, S_MOV64)
KERNEL32(k_addco, S_ADDCO) KERNEL32(k_addc, S_ADDC) KERNEL32(k_alignbit, S_ALIGNBIT) KERNEL32(k_bfi, S_BFI) KERNEL32(k_not, S_NOT)
KERNEL32(k_ashr, S_ASHR) KERNEL32(k_addx2, S_ADDX2) KERNEL32(k_sdwa, S_SDWA) KERNEL32(k_cnd, S_CND) KERNEL32(k_minu, S_MINU)
KERNEL32(k_lshlor, S_LSHLOR) KERNEL32(k_mullo, S_MULLO) KERNEL32(k_mad24, S_MAD24)
int main() {
    unsigned long long *out, *cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * 8)); CHECK(hipMalloc(&cyc, 8));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
#define RUN(K) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); K<<<cus, 1024>>>(out, cyc, 5); hipDeviceSynchronize(); \
      hipEventRecord(e0); K<<<cus, 1024>>>(out, cyc, 5); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
      printf("%-12s kernel %7.1f us -> %5.2f cycles@2.4GHz per instr per SIMD (4 waves/SIMD)\n", #K, ms * 1e3, ms * 1e-3 * 2.4e9 / (ITERS * 8.0 * 4)); }
    RUN(k_shl64) RUN(k_lshladd64) RUN(k_add64) RUN(k_mov64) RUN(k_addco) RUN(k_addc) RUN(k_alignbit) RUN(k_bfi) RUN(k_not) RUN(k_ashr) RUN(k_addx2) RUN(k_sdwa) RUN(k_cnd) RUN(k_minu) RUN(k_lshlor) RUN(k_mullo) RUN(k_mad24)
    return 0;
}
