// valu_rates_src.hip — what decides the issue cost of a 32-bit VALU instruction on gfx950: its encoding (VOP1/VOP2 = 4 bytes,
// VOP3 / literal / SDWA = 8 bytes), or the number of DISTINCT vector registers it reads?  Hard registers, 4 independent
// accumulators (v1..v4), the second / third sources v5..v8 / v9, scalars s20 / s21; 4 waves per SIMD unless WAVES says else.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define ITERS 2048
// one body = the instruction on each of the four accumulators; four bodies per loop trip
#define KERNEL(NAME, I1, I2, I3, I4)                                                                          \
__global__ void NAME(uint32_t *out, uint32_t s) {                                                             \
    uint32_t r;                                                                                               \
    asm volatile(                                                                                             \
        "v_mov_b32 v1, %1\n v_add_u32 v2, 1, v1\n v_add_u32 v3, 2, v1\n v_add_u32 v4, 3, v1\n"                  \
        "v_add_u32 v5, 4, v1\n v_add_u32 v6, 5, v1\n v_add_u32 v7, 6, v1\n v_add_u32 v8, 7, v1\n v_add_u32 v9, 9, v1\n" \
        "v_add_u32 v10, 11, v1\n v_add_u32 v11, 12, v1\n v_add_u32 v12, 13, v1\n v_add_u32 v13, 14, v1\n"        \
        "s_mov_b32 s20, 0x9e3779b1\n s_mov_b32 s21, 0x7f4a7c15\n s_movk_i32 s22, %2\n"                           \
        "L_loop_%=:\n"                                                                                         \
        I1 I2 I3 I4 I1 I2 I3 I4 I1 I2 I3 I4 I1 I2 I3 I4                                                       \
        "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 L_loop_%=\n"                              \
        "v_xor_b32 v1, v1, v2\n v_xor_b32 v3, v3, v4\n v_xor_b32 %0, v1, v3\n"                                  \
        : "=v"(r) : "v"(s ^ threadIdx.x), "n"(ITERS)                                                          \
        : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "s20", "s21", "s22", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "vcc", "scc"); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                           \
}
#define K4(NAME, A, B, C, D) KERNEL(NAME, A "\n", B "\n", C "\n", D "\n")
// --- encoding against sources -------------------------------------------------------------------------------------
K4(xor_v_v,      "v_xor_b32 v1, v5, v1", "v_xor_b32 v2, v6, v2", "v_xor_b32 v3, v7, v3", "v_xor_b32 v4, v8, v4")          // VOP2, two VGPRs of the same bank (index mod 4)
K4(xor_v_v_nb,   "v_xor_b32 v1, v6, v1", "v_xor_b32 v2, v7, v2", "v_xor_b32 v3, v8, v3", "v_xor_b32 v4, v5, v4")          // different banks
K4(xor_s_v,      "v_xor_b32 v1, s20, v1", "v_xor_b32 v2, s20, v2", "v_xor_b32 v3, s20, v3", "v_xor_b32 v4, s20, v4")      // VOP2, SGPR + VGPR
K4(xor_c_v,      "v_xor_b32 v1, 31, v1", "v_xor_b32 v2, 31, v2", "v_xor_b32 v3, 31, v3", "v_xor_b32 v4, 31, v4")          // VOP2, inline constant
K4(xor_lit_v,    "v_xor_b32 v1, 0x12345, v1", "v_xor_b32 v2, 0x12345, v2", "v_xor_b32 v3, 0x12345, v3", "v_xor_b32 v4, 0x12345, v4")  // VOP2 + literal: 8 bytes, one VGPR
K4(xor_e64_c_v,  "v_xor_b32_e64 v1, 31, v1", "v_xor_b32_e64 v2, 31, v2", "v_xor_b32_e64 v3, 31, v3", "v_xor_b32_e64 v4, 31, v4")      // VOP3 encoding of the same, one VGPR
K4(xor_e64_v_v,  "v_xor_b32_e64 v1, v6, v1", "v_xor_b32_e64 v2, v7, v2", "v_xor_b32_e64 v3, v8, v3", "v_xor_b32_e64 v4, v5, v4")
K4(xor_same,     "v_xor_b32 v1, v1, v1", "v_xor_b32 v2, v2, v2", "v_xor_b32 v3, v3, v3", "v_xor_b32 v4, v4, v4")
K4(mov_v,        "v_mov_b32 v1, v6", "v_mov_b32 v2, v7", "v_mov_b32 v3, v8", "v_mov_b32 v4, v5")
K4(add_v_v,      "v_add_u32 v1, v6, v1", "v_add_u32 v2, v7, v2", "v_add_u32 v3, v8, v3", "v_add_u32 v4, v5, v4")
K4(add_s_v,      "v_add_u32 v1, s20, v1", "v_add_u32 v2, s20, v2", "v_add_u32 v3, s20, v3", "v_add_u32 v4, s20, v4")
K4(min_v_v,      "v_min_u32 v1, v6, v1", "v_min_u32 v2, v7, v2", "v_min_u32 v3, v8, v3", "v_min_u32 v4, v5, v4")
K4(min_s_v,      "v_min_u32 v1, s20, v1", "v_min_u32 v2, s20, v2", "v_min_u32 v3, s20, v3", "v_min_u32 v4, s20, v4")
K4(and_s_v,      "v_and_b32 v1, s20, v1", "v_and_b32 v2, s20, v2", "v_and_b32 v3, s20, v3", "v_and_b32 v4, s20, v4")
K4(shl_c_v,      "v_lshlrev_b32 v1, 1, v1", "v_lshlrev_b32 v2, 1, v2", "v_lshlrev_b32 v3, 1, v3", "v_lshlrev_b32 v4, 1, v4")
K4(shl_v_v,      "v_lshlrev_b32 v1, v6, v1", "v_lshlrev_b32 v2, v7, v2", "v_lshlrev_b32 v3, v8, v3", "v_lshlrev_b32 v4, v5, v4")
K4(mul24_s_v,    "v_mul_u32_u24 v1, s20, v1", "v_mul_u32_u24 v2, s20, v2", "v_mul_u32_u24 v3, s20, v3", "v_mul_u32_u24 v4, s20, v4")  // VOP2 multiply, one VGPR
K4(mul24_v_v,    "v_mul_u32_u24 v1, v6, v1", "v_mul_u32_u24 v2, v7, v2", "v_mul_u32_u24 v3, v8, v3", "v_mul_u32_u24 v4, v5, v4")
// --- VOP3-only ops by number of vector sources --------------------------------------------------------------------
K4(mad24_v_s_s,  "v_mad_u32_u24 v1, v1, s20, 17", "v_mad_u32_u24 v2, v2, s20, 17", "v_mad_u32_u24 v3, v3, s20, 17", "v_mad_u32_u24 v4, v4, s20, 17")
K4(mad24_v_s_v,  "v_mad_u32_u24 v1, v1, s20, v6", "v_mad_u32_u24 v2, v2, s20, v7", "v_mad_u32_u24 v3, v3, s20, v8", "v_mad_u32_u24 v4, v4, s20, v5")
K4(mad24_v_v_v,  "v_mad_u32_u24 v1, v1, v6, v9", "v_mad_u32_u24 v2, v2, v7, v9", "v_mad_u32_u24 v3, v3, v8, v9", "v_mad_u32_u24 v4, v4, v5, v9")
K4(alignbyte_vv, "v_alignbyte_b32 v1, v6, v1, 1", "v_alignbyte_b32 v2, v7, v2, 1", "v_alignbyte_b32 v3, v8, v3, 1", "v_alignbyte_b32 v4, v5, v4, 1")
K4(alignbyte_1v, "v_alignbyte_b32 v1, v1, v1, 1", "v_alignbyte_b32 v2, v2, v2, 1", "v_alignbyte_b32 v3, v3, v3, 1", "v_alignbyte_b32 v4, v4, v4, 1")
K4(alignbyte_ro, "v_alignbyte_b32 v1, v10, v11, 1", "v_alignbyte_b32 v2, v11, v12, 1", "v_alignbyte_b32 v3, v12, v13, 1", "v_alignbyte_b32 v4, v13, v10, 1")  // sources not the destination
K4(perm_vv,      "v_perm_b32 v1, v6, v1, s20", "v_perm_b32 v2, v7, v2, s20", "v_perm_b32 v3, v8, v3, s20", "v_perm_b32 v4, v5, v4, s20")
K4(bfe_1v,       "v_bfe_u32 v1, v1, 8, 8", "v_bfe_u32 v2, v2, 8, 8", "v_bfe_u32 v3, v3, 8, 8", "v_bfe_u32 v4, v4, 8, 8")
K4(lshlor_1v,    "v_lshl_or_b32 v1, v1, 1, 1", "v_lshl_or_b32 v2, v2, 1, 1", "v_lshl_or_b32 v3, v3, 1, 1", "v_lshl_or_b32 v4, v4, 1, 1")
K4(min3_vvv,     "v_min3_u32 v1, v1, v6, v9", "v_min3_u32 v2, v2, v7, v9", "v_min3_u32 v3, v3, v8, v9", "v_min3_u32 v4, v4, v5, v9")
K4(min3_ro,      "v_min3_u32 v1, v10, v11, v12", "v_min3_u32 v2, v11, v12, v13", "v_min3_u32 v3, v12, v13, v10", "v_min3_u32 v4, v13, v10, v11")
K4(andor_vvs,    "v_and_or_b32 v1, v1, s20, v6", "v_and_or_b32 v2, v2, s20, v7", "v_and_or_b32 v3, v3, s20, v8", "v_and_or_b32 v4, v4, s20, v5")
K4(xad_vvs,      "v_xad_u32 v1, v1, s20, v6", "v_xad_u32 v2, v2, s20, v7", "v_xad_u32 v3, v3, s20, v8", "v_xad_u32 v4, v4, s20, v5")
K4(bitop3_vvv,   "v_bitop3_b32 v1, v1, v6, v9 bitop3:0x96", "v_bitop3_b32 v2, v2, v7, v9 bitop3:0x96", "v_bitop3_b32 v3, v3, v8, v9 bitop3:0x96", "v_bitop3_b32 v4, v4, v5, v9 bitop3:0x96")
K4(sdwa_b1,      "v_lshlrev_b32_sdwa v1, 2, v6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa v2, 2, v7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1",
                 "v_lshlrev_b32_sdwa v3, 2, v8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa v4, 2, v5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
K4(cmp_v_v,      "v_cmp_eq_u32 vcc, v6, v1", "v_cmp_eq_u32 vcc, v7, v2", "v_cmp_eq_u32 vcc, v8, v3", "v_cmp_eq_u32 vcc, v5, v4")
K4(cmp_s_v,      "v_cmp_eq_u32 vcc, s20, v1", "v_cmp_eq_u32 vcc, s20, v2", "v_cmp_eq_u32 vcc, s20, v3", "v_cmp_eq_u32 vcc, s20, v4")
K4(cmp_e64_s_v,  "v_cmp_eq_u32 s[24:25], s20, v1", "v_cmp_eq_u32 s[26:27], s20, v2", "v_cmp_eq_u32 s[28:29], s20, v3", "v_cmp_eq_u32 s[30:31], s20, v4")
K4(pk_add_u16,   "v_pk_add_u16 v1, v1, v6", "v_pk_add_u16 v2, v2, v7", "v_pk_add_u16 v3, v3, v8", "v_pk_add_u16 v4, v4, v5")
K4(pk_mul_lo16,  "v_pk_mul_lo_u16 v1, v1, s20", "v_pk_mul_lo_u16 v2, v2, s20", "v_pk_mul_lo_u16 v3, v3, s20", "v_pk_mul_lo_u16 v4, v4, s20")
K4(pk_mad_u16,   "v_pk_mad_u16 v1, v1, s20, v6", "v_pk_mad_u16 v2, v2, s20, v7", "v_pk_mad_u16 v3, v3, s20, v8", "v_pk_mad_u16 v4, v4, s20, v5")
K4(pk_min_u16,   "v_pk_min_u16 v1, v1, v6", "v_pk_min_u16 v2, v2, v7", "v_pk_min_u16 v3, v3, v8", "v_pk_min_u16 v4, v4, v5")
// mixes: does a slow instruction between fast ones cost its own time only?
K4(mix_fast_slow, "v_xor_b32 v1, 31, v1", "v_alignbyte_b32 v2, v7, v2, 1", "v_xor_b32 v3, 31, v3", "v_alignbyte_b32 v4, v5, v4, 1")
K4(salu_only,    "s_add_u32 s24, s24, s20", "s_xor_b32 s25, s25, s20", "s_add_u32 s26, s26, s20", "s_xor_b32 s27, s27, s20")
K4(mix_valu_salu, "v_xor_b32 v1, v6, v1", "s_add_u32 s24, s24, s20", "v_xor_b32 v3, v8, v3", "s_xor_b32 s25, s25, s20")

int main(int argc, char **argv) {
    uint32_t *out;
    CHECK(hipMalloc(&out, 256 * 2048 * 4));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const int waves = argc > 1 ? atoi(argv[1]) : 4;           // per SIMD
    printf("%d CUs, %d waves per SIMD; cycles at 2.4 GHz per instruction and SIMD\n", cus, waves);
#define RUN(K) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); K<<<cus, 256 * waves>>>(out, 5); hipDeviceSynchronize(); \
      hipEventRecord(e0); K<<<cus, 256 * waves>>>(out, 5); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
      printf("%-14s %7.1f us  %5.2f\n", #K, ms * 1e3, ms * 1e-3 * 2.4e9 / (ITERS * 16.0 * waves)); }
    RUN(xor_v_v) RUN(xor_v_v_nb) RUN(xor_s_v) RUN(xor_c_v) RUN(xor_lit_v) RUN(xor_e64_c_v) RUN(xor_e64_v_v) RUN(xor_same) RUN(mov_v)
    RUN(add_v_v) RUN(add_s_v) RUN(min_v_v) RUN(min_s_v) RUN(and_s_v) RUN(shl_c_v) RUN(shl_v_v) RUN(mul24_s_v) RUN(mul24_v_v)
    RUN(mad24_v_s_s) RUN(mad24_v_s_v) RUN(mad24_v_v_v) RUN(alignbyte_vv) RUN(alignbyte_1v) RUN(alignbyte_ro) RUN(perm_vv) RUN(bfe_1v) RUN(lshlor_1v)
    RUN(min3_vvv) RUN(min3_ro) RUN(andor_vvs) RUN(xad_vvs) RUN(bitop3_vvv) RUN(sdwa_b1) RUN(cmp_v_v) RUN(cmp_s_v) RUN(cmp_e64_s_v)
    RUN(pk_add_u16) RUN(pk_mul_lo16) RUN(pk_mad_u16) RUN(pk_min_u16) RUN(mix_fast_slow) RUN(salu_only) RUN(mix_valu_salu)
    return 0;
}
