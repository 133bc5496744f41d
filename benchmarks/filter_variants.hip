// filter_variants.hip — micro-benchmark of candidate inner loops for the n-gram filter kernel.
// Standalone: hipcc --offload-arch=gfx950 -O3 -std=c++17 filter_variants.hip -o filter_variants
// Each variant streams the same N bytes of pseudo-random DNA and counts fast hits of G=3 6-byte
// n-grams (exactness is not the point here — relative cost of the compare/reduce strategy is).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <cstring>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#define THREADS 256
#define ROWB (THREADS * 16)

__global__ void gen_dna(uint8_t *buf, uint64_t n, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n / 8; i += stride) {
        uint64_t z = (i + seed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        uint64_t out = 0;
        for (int b = 0; b < 8; ++b) out |= (uint64_t)("ACGT"[(z >> (2 * b + 7)) & 3]) << (8 * b);
        reinterpret_cast<uint64_t *>(buf)[i] = out;
    }
}

struct Args {
    uint32_t A[4], B[4], H[4], HP[4];
    uint32_t K;       // hash multiplier (24 bit)
    uint32_t d2;      // second window offset (exact variants)
    uint32_t dh;      // hash window offset
};

__device__ __forceinline__ uint32_t win(uint32_t lo, uint32_t hi, int b) {
    return b == 0 ? lo : __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)b);
}
#define WIN(w, o) win((w)[(o) >> 2], (w)[((o) >> 2) + 1], (o) & 3)

// ---- V4: streaming only -------------------------------------------------------------------
template <int ROWS>
__global__ __launch_bounds__(THREADS) void k_stream(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint32_t acc = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc += v[r].x ^ v[r].y ^ v[r].z ^ v[r].w ^ h[r].x ^ h[r].y;
    }
    if (acc == 0x12345678u) atomicAdd(cnt, 1ull);
    if (threadIdx.x == 0) { if (blockIdx.x == 0) { cnt[1] = clock64() - c0; cnt[2] = wall_clock64() - w0; } if (blockIdx.x < 8192) { cnt[8 + 2 * blockIdx.x] = w0; cnt[9 + 2 * blockIdx.x] = wall_clock64(); cnt[8 + 2 * 8192 + 512 + blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); } }
}

// shared rare path: count the lanes (stand-in for queue push + drain)
__device__ __forceinline__ void rare(unsigned long long mask, uint32_t &qn) { qn += (uint32_t)__popcll(mask); }

// ---- V0: exact two-window compare, s_and + s_or, branch per offset -------------------------
template <int ROWS, int D2>
__global__ __launch_bounds__(THREADS) void k_v0(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t A[3] = {a.A[0], a.A[1], a.A[2]}, B[3] = {a.B[0], a.B[1], a.B[2]};
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                const uint32_t x = WIN(w, o), y = WIN(w, o + D2);
                unsigned long long mk[3], any = 0;
#pragma unroll
                for (int g = 0; g < 3; ++g) { mk[g] = __ballot(x == A[g]) & __ballot(y == B[g]); any |= mk[g]; }
                if (any) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) if (mk[g]) rare(mk[g], qn);
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V1: mad_u24 hash, cmp + s_or, branch per offset --------------------------------------
template <int ROWS, int DH>
__global__ __launch_bounds__(THREADS) void k_v1(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t H[3] = {a.H[0], a.H[1], a.H[2]};
    const uint32_t K = a.K;
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                const uint32_t x = WIN(w, o), y = WIN(w, o + DH);
                const uint32_t hv = __umul24(y, K) + x;   // v_mad_u32_u24
                unsigned long long mk[3], any = 0;
#pragma unroll
                for (int g = 0; g < 3; ++g) { mk[g] = __ballot(hv == H[g]); any |= mk[g]; }
                if (any) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) if (mk[g]) rare(mk[g], qn);
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V2: hash, cmp + s_or, branch per 4 offsets --------------------------------------------
template <int ROWS, int DH>
__global__ __launch_bounds__(THREADS) void k_v2(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t H[3] = {a.H[0], a.H[1], a.H[2]};
    const uint32_t K = a.K;
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned long long mk[4][3], any = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    const uint32_t x = WIN(w, o), y = WIN(w, o + DH);
                    const uint32_t hv = __umul24(y, K) + x;
#pragma unroll
                    for (int g = 0; g < 3; ++g) { mk[i][g] = __ballot(hv == H[g]); any |= mk[i][g]; }
                }
                if (any) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) if (mk[i][g]) rare(mk[i][g], qn);
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V3: hash, xor + min3 accumulate (no SALU), one ballot per 4 offsets --------------------
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(a, min(b, c)); }
template <int ROWS, int DH, int GROUP, int WRAPBITS = 63, bool HV = false, int NB = 3, int ASSIGN = 0>
__global__ __launch_bounds__(THREADS) void k_v3(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint32_t H[3] = {a.H[0], a.H[1], a.H[2]};
    if (HV) {   // block hashes in VGPRs: v_xor with an SGPR operand issues at ~4.5 cycles, VGPR-only at ~2.6
#pragma unroll
        for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(H[g]) : "s"(a.H[g]));
    }
    const uint32_t K = a.K;
    uint32_t qn = 0;
    const uint64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint64_t t_begin = ASSIGN == 1 ? blockIdx.x * per : blockIdx.x;
    const uint64_t t_end = ASSIGN == 1 ? (t_begin + per < ntiles ? t_begin + per : ntiles) : ntiles;
    const uint64_t t_step = ASSIGN == 1 ? 1 : gridDim.x;
    // ASSIGN 2: each wave reads ROWS consecutive 1 KiB segments (4 KiB contiguous per wave)
    const uint32_t lane_off = ASSIGN == 2 ? (threadIdx.x >> 6) * (ROWS * 1024u) + (threadIdx.x & 63u) * 16u : threadIdx.x * 16u;
    constexpr uint32_t ROWSTEP = ASSIGN == 2 ? 1024u : ROWB;
    const uint32_t my_tiles = (uint32_t)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    uint32_t it = 0;
    for (uint64_t tile = t_begin; tile < t_end; tile += t_step, ++it) {
        if (ASSIGN == 3) {      // priority falls with progress: the least advanced wave issues first
            const uint32_t q = it * 4 / my_tiles;
            if (q == 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        if (ASSIGN == 4) {      // the opposite: the most advanced wave first
            const uint32_t q = it * 4 / my_tiles;
            if (q == 0) __builtin_amdgcn_s_setprio(0); else if (q == 1) __builtin_amdgcn_s_setprio(1);
            else if (q == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
        }
        // WRAPBITS < 63: all tiles alias a small L2-resident region -> compute-only time
        const uint8_t *src = buf + (tile & ((1ull << WRAPBITS) - 1)) * (uint64_t)(ROWB * ROWS) + lane_off;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWSTEP); h[r] = *(const uint2 *)(src + r * ROWSTEP + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 16 / GROUP; ++j) {
                uint32_t acc = 0xffffffffu;
                uint32_t hv[GROUP];
#pragma unroll
                for (int i = 0; i < GROUP; ++i) {
                    const int o = GROUP * j + i;
                    const uint32_t x = WIN(w, o), y = WIN(w, o + DH);
                    hv[i] = __umul24(y, K) + x;
                    if (NB == 3) { acc = min3u(acc, hv[i] ^ H[0], hv[i] ^ H[1]); acc = min(acc, hv[i] ^ H[2]); }
                    else if (NB == 2) acc = min3u(acc, hv[i] ^ H[0], hv[i] ^ H[1]);
                    else acc = min(acc, hv[i] ^ H[0]);
                }
                const unsigned long long any = __ballot(acc == 0);
                if (any) {
#pragma unroll
                    for (int i = 0; i < GROUP; ++i)
#pragma unroll
                        for (int g = 0; g < NB; ++g) { const unsigned long long mm = __ballot(hv[i] == H[g]); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
    if (threadIdx.x == 0) { if (blockIdx.x == 0) { cnt[1] = clock64() - c0; cnt[2] = wall_clock64() - w0; } if (blockIdx.x < 8192) { cnt[8 + 2 * blockIdx.x] = w0; cnt[9 + 2 * blockIdx.x] = wall_clock64(); cnt[8 + 2 * 8192 + 512 + blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); } }
}

// ---- V5: hash, v_cmp + v_addc-style per-lane counter (no SALU), ballot per GROUP offsets ------
template <int ROWS, int DH, int GROUP>
__global__ __launch_bounds__(THREADS) void k_v5(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t H[3] = {a.H[0], a.H[1], a.H[2]};
    const uint32_t K = a.K;
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 16 / GROUP; ++j) {
                uint32_t acc = 0;
                uint32_t hv[GROUP];
#pragma unroll
                for (int i = 0; i < GROUP; ++i) {
                    const int o = GROUP * j + i;
                    const uint32_t x = WIN(w, o), y = WIN(w, o + DH);
                    hv[i] = __umul24(y, K) + x;
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc += (hv[i] == H[g]) ? 1u : 0u;
                }
                const unsigned long long any = __ballot(acc != 0);
                if (any) {
#pragma unroll
                    for (int i = 0; i < GROUP; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) { const unsigned long long mm = __ballot(hv[i] == H[g]); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}


// ---- V6: v_qsad_pk_u16_u8 / v_mqsad_pk_u16_u8: 4 byte offsets per instruction, exact 8-byte compare ----
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b) {
    us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b);
    us2 r = __builtin_elementwise_min(x, y);
    return __builtin_bit_cast(uint32_t, r);
}
template <int ROWS>
__global__ __launch_bounds__(THREADS) void k_v6(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t A[3] = {a.A[0], a.A[1], a.A[2]}, B[3] = {a.B[0], a.B[1], a.B[2]};   // B = bytes 4..7 (masked)
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t p0 = ((uint64_t)w[j + 1] << 32) | w[j];
                const uint64_t p1 = ((uint64_t)w[j + 2] << 32) | w[j + 1];
                uint64_t R[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    R[g] = __builtin_amdgcn_qsad_pk_u16_u8(p0, A[g], 0ull);
                    R[g] = __builtin_amdgcn_mqsad_pk_u16_u8(p1, B[g], R[g]);
                }
                uint32_t lo = pkmin(pkmin((uint32_t)R[0], (uint32_t)R[1]), (uint32_t)R[2]);
                uint32_t hi = pkmin(pkmin((uint32_t)(R[0] >> 32), (uint32_t)(R[1] >> 32)), (uint32_t)(R[2] >> 32));
                uint32_t mm = pkmin(lo, hi);
                mm = min(mm & 0xffffu, mm >> 16);
                const unsigned long long any = __ballot(mm == 0);
                if (any) {
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned long long mk = __ballot(((R[g] >> (16 * i)) & 0xffffu) == 0);
                            if (mk) rare(mk, qn);
                        }
                    }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V7/V8: per-workgroup LDS byte table indexed by the top TB bits of the hash; FRAC_LDS of the
// 4-offset groups of a row use the table (VALU: alignbyte + mad + shift + or), the rest xor/min3.
template <int ROWS, int DH, int TB, int LDS_GROUPS>
__global__ __launch_bounds__(THREADS) void k_v7(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    __shared__ uint8_t table[1 << TB];
    for (int i = threadIdx.x * 16; i < (1 << TB); i += THREADS * 16) *(uint4 *)(table + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (threadIdx.x < 3) table[a.H[threadIdx.x] >> (32 - TB)] = 1;
    __syncthreads();
    uint32_t H[3] = {a.H[0], a.H[1], a.H[2]};
    const uint32_t K = a.K;
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int o = 4 * j + i; hv[i] = __umul24(WIN(w, o + DH), K) + WIN(w, o); }
                unsigned long long any;
                if (j < LDS_GROUPS) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc |= table[hv[i] >> (32 - TB)];
                    any = __ballot(acc != 0);
                } else {
                    uint32_t acc = 0xffffffffu;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc = min3u(acc, hv[i] ^ H[0], hv[i] ^ H[1]); acc = min(acc, hv[i] ^ H[2]); }
                    any = __ballot(acc == 0);
                }
                if (any) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) { const unsigned long long mm = __ballot(hv[i] == H[g]); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V8: four byte-shifted (unaligned) dwordx4 loads per row instead of v_alignbyte: every 32-bit
// window of the row is a loaded register (VOP3 ops such as v_alignbyte cost 2x a VOP2 op) ----------
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };
template <int ROWS, int DH>
__global__ __launch_bounds__(THREADS) void k_v8(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t H[3] = {a.H[0], a.H[1], a.H[2]};
    const uint32_t K = a.K;
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        U4 p[ROWS][4]; uint32_t e[ROWS][4];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                p[r][b] = *(const U4 *)(src + r * ROWB + b);
                e[r][b] = *(const u32_unaligned *)(src + r * ROWB + 16 + b);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            // win(o) for o in 0..19
            uint32_t wv[20];
#pragma unroll
            for (int b = 0; b < 4; ++b) { wv[b] = p[r][b].x; wv[4 + b] = p[r][b].y; wv[8 + b] = p[r][b].z; wv[12 + b] = p[r][b].w; wv[16 + b] = e[r][b]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t acc = 0xffffffffu;
                uint32_t hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    hv[i] = __umul24(wv[o + DH], K) + wv[o];
                    acc = min3u(acc, hv[i] ^ H[0], hv[i] ^ H[1]);
                    acc = min(acc, hv[i] ^ H[2]);
                }
                const unsigned long long any = __ballot(acc == 0);
                if (any) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) { const unsigned long long mm = __ballot(hv[i] == H[g]); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V9: v3 + software pipelining: the next tile's rows are requested before this tile is processed ----
template <int ROWS, int DH>
__global__ __launch_bounds__(THREADS) void k_v9(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t H[3] = {a.H[0], a.H[1], a.H[2]};
    const uint32_t K = a.K;
    uint32_t qn = 0;
    uint4 nv[ROWS]; uint2 nh[ROWS];
    uint64_t tile = blockIdx.x;
    if (tile < ntiles) {
        const uint8_t *src = buf + tile * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { nv[r] = *(const uint4 *)(src + r * ROWB); nh[r] = *(const uint2 *)(src + r * ROWB + 16); }
    }
    for (; tile < ntiles; tile += gridDim.x) {
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = nv[r]; h[r] = nh[r]; }
        const uint64_t nt = tile + gridDim.x;
        if (nt < ntiles) {
            const uint8_t *src = buf + nt * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { nv[r] = *(const uint4 *)(src + r * ROWB); nh[r] = *(const uint2 *)(src + r * ROWB + 16); }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t acc = 0xffffffffu;
                uint32_t hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    hv[i] = __umul24(WIN(w, o + DH), K) + WIN(w, o);
                    acc = min3u(acc, hv[i] ^ H[0], hv[i] ^ H[1]);
                    acc = min(acc, hv[i] ^ H[2]);
                }
                const unsigned long long any = __ballot(acc == 0);
                if (any) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) { const unsigned long long mm = __ballot(hv[i] == H[g]); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V10: 16-bit hashes of two adjacent offsets packed in one dword (v_perm / v_pk_mad_u16 /
//      v_xor / v_pk_min_u16), block hashes in VGPRs, one ballot per GROUP offsets ----------------
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c)); return d; }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t min_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
template <int ROWS, int GROUP, int WRAPBITS = 63>
__global__ __launch_bounds__(THREADS) void k_v10(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t HP[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(HP[g]) : "s"(a.HP[g]));
    const uint32_t K2 = (a.K & 0xffffu) * 0x10001u;
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + (tile & ((1ull << WRAPBITS) - 1)) * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
            uint32_t P[10];     // P[j] = 16-bit windows at byte offsets 2j and 2j+1
#pragma unroll
            for (int j = 0; j < 10; ++j)
                P[j] = (j & 1) ? __builtin_amdgcn_perm(w[j / 2 + 1], w[j / 2], 0x04030302u)
                               : __builtin_amdgcn_perm(w[j / 2], w[j / 2], 0x02010100u);
            uint32_t hp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) hp[j] = pk_mad(P[j + 1], K2, P[j]) ^ P[j + 2];
#pragma unroll
            for (int q = 0; q < 16 / GROUP; ++q) {
                uint32_t acc = 0xffffffffu;
#pragma unroll
                for (int j = q * GROUP / 2; j < (q + 1) * GROUP / 2; ++j)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc = pk_min(acc, hp[j] ^ HP[g]);
                const uint32_t t = min_u16(acc >> 16, acc);
                if (__ballot((t & 0xffffu) == 0)) {
#pragma unroll
                    for (int j = q * GROUP / 2; j < (q + 1) * GROUP / 2; ++j)
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const uint32_t x = hp[j] ^ HP[g];
                            const unsigned long long m0 = __ballot((x & 0xffffu) == 0), m1 = __ballot((x >> 16) == 0);
                            if (m0) rare(m0, qn);
                            if (m1) rare(m1, qn);
                        }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
}

// ---- V11: v3 (block hashes in VGPRs) with DYNAMIC work distribution: every wave grabs chunks of
//      C consecutive 4 KiB wave-tiles from one of P atomic counters (next grab prefetched) ----------
template <int NB, int WRAPBITS = 63>
__global__ __launch_bounds__(THREADS) void k_v11(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    const unsigned long long w0 = wall_clock64();
    uint32_t H[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(H[g]) : "s"(a.H[g]));
    const uint32_t K = a.K;
    const uint32_t C = a.d2, P = a.dh;                 // chunk size in wave-tiles, number of counters (reused fields)
    const uint64_t nwt = ntiles * 4;                   // 4 KiB wave-tiles
    const uint64_t nchunks = (nwt + C - 1) / C;
    unsigned int *sched = reinterpret_cast<unsigned int *>(cnt + 8 + 2 * 8192);
    const uint32_t part = blockIdx.x % P;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t qn = 0;
    uint32_t nextv = 0;
    if (lane == 0) nextv = atomicAdd(&sched[part * 16], 1u);
    for (;;) {
        const uint32_t c = __builtin_amdgcn_readfirstlane(nextv);
        const uint64_t gc = (uint64_t)c * P + part;
        if (gc >= nchunks) break;
        if (lane == 0) nextv = atomicAdd(&sched[part * 16], 1u);
        const uint64_t t0 = gc * C, t1 = t0 + C < nwt ? t0 + C : nwt;
        for (uint64_t wt = t0; wt < t1; ++wt) {
            const uint8_t *src = buf + (wt & ((1ull << WRAPBITS) - 1)) * 4096ull + lane * 16u;
            uint4 v[4]; uint2 h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = *(const uint4 *)(src + r * 1024); h[r] = *(const uint2 *)(src + r * 1024 + 16); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t acc = 0xffffffffu;
                    uint32_t hv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int o = 4 * j + i;
                        hv[i] = __umul24(WIN(w, o + 3), K) + WIN(w, o);
                        if (NB == 3) { acc = min3u(acc, hv[i] ^ H[0], hv[i] ^ H[1]); acc = min(acc, hv[i] ^ H[2]); }
                        else acc = min(acc, hv[i] ^ H[0]);
                    }
                    if (__ballot(acc == 0)) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int g = 0; g < NB; ++g) { const unsigned long long mm = __ballot(hv[i] == H[g]); if (mm) rare(mm, qn); }
                    }
                }
            }
        }
    }
    if (lane == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
    if (threadIdx.x == 0 && blockIdx.x < 8192) { cnt[8 + 2 * blockIdx.x] = w0; cnt[9 + 2 * blockIdx.x] = wall_clock64(); if (blockIdx.x == 0) { cnt[1] = 1; cnt[2] = 1; } }
}

// ---- V12: v3 (1 or 3 blocks, hashes in VGPRs) with a per-tile timeline of ONE wave per CU-slot class:
//      core-clock stamps after issuing the loads, when row 0 has arrived, and at the end of the tile.
template <int NB, int WRAPBITS = 63>
__global__ __launch_bounds__(THREADS) void k_v12(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    uint32_t H[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(H[g]) : "s"(a.H[g]));
    const uint32_t K = a.K;
    uint32_t qn = 0;
    unsigned long long t_wait = 0, t_comp = 0, t_issue = 0, n_t = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + (tile & ((1ull << WRAPBITS) - 1)) * (uint64_t)(ROWB * 4) + threadIdx.x * 16u;
        uint4 v[4]; uint2 h[4];
        const unsigned long long c0 = clock64();
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
        const unsigned long long c1 = clock64();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        const unsigned long long c2 = clock64();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t acc = 0xffffffffu;
                uint32_t hv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    hv[i] = __umul24(WIN(w, o + 3), K) + WIN(w, o);
                    if (NB == 3) { acc = min3u(acc, hv[i] ^ H[0], hv[i] ^ H[1]); acc = min(acc, hv[i] ^ H[2]); }
                    else acc = min(acc, hv[i] ^ H[0]);
                }
                if (__ballot(acc == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < NB; ++g) { const unsigned long long mm = __ballot(hv[i] == H[g]); if (mm) rare(mm, qn); }
                }
            }
        }
        const unsigned long long c3 = clock64();
        t_issue += c1 - c0; t_wait += c2 - c1; t_comp += c3 - c2; ++n_t;
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
    if (threadIdx.x == 0) {      // per-CU-slot class = blockIdx / (gridDim/8): early vs late workgroups
        const uint32_t cls = blockIdx.x * 8 / gridDim.x;
        atomicAdd(&cnt[8 + cls * 4 + 0], t_issue); atomicAdd(&cnt[8 + cls * 4 + 1], t_wait);
        atomicAdd(&cnt[8 + cls * 4 + 2], t_comp); atomicAdd(&cnt[8 + cls * 4 + 3], n_t);
    }
}

// ---- V13: block hashes in a 64-slot table held in ONE VGPR (lane s = slot s); every offset fetches
//      its slot (top 6 hash bits) with ds_bpermute_b32 and compares once: cost independent of #blocks ----
template <int ROWS, int WRAPBITS = 63, int NLUT = 4>
__global__ __launch_bounds__(THREADS) void k_v13(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    const unsigned long long w0 = wall_clock64();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t table = ((lane + 1u) & 63u) << 26;            // sentinel: a value that lives in another slot
#pragma unroll
    for (int g = 0; g < 3; ++g) if ((a.H[g] >> 26) == lane) table = a.H[g];
    const uint32_t K = a.K;
    uint32_t H[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(H[g]) : "s"(a.H[g]));
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + (tile & ((1ull << WRAPBITS) - 1)) * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t hv[4], t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    hv[i] = __umul24(WIN(w, o + 3), K) + WIN(w, o);
                    // NLUT of every 4 offsets go through the LDS crossbar, the rest through VALU compares
                    if (i < NLUT) t[i] = hv[i] ^ (uint32_t)__builtin_amdgcn_ds_bpermute((int)(hv[i] >> 24), (int)table);
                    else t[i] = min3u(hv[i] ^ H[0], hv[i] ^ H[1], hv[i] ^ H[2]);
                }
                const uint32_t acc = min(min3u(t[0], t[1], t[2]), t[3]);
                if (__ballot(acc == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const unsigned long long mm = __ballot(t[i] == 0); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
    if (threadIdx.x == 0) { if (blockIdx.x == 0) { cnt[1] = 1; cnt[2] = 1; } if (blockIdx.x < 8192) { cnt[8 + 2 * blockIdx.x] = w0; cnt[9 + 2 * blockIdx.x] = wall_clock64(); } }
}

// ---- V14: like V13 but the 64-slot table sits in LDS and is read with ds_read_b32 (2 VALU ops for
//      the address instead of 1, but a plain LDS read instead of the crossbar) ----------------------
template <int ROWS, int WRAPBITS = 63, int NLUT = 4>
__global__ __launch_bounds__(THREADS) void k_v14(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    __shared__ uint32_t tab[64];
    const unsigned long long w0 = wall_clock64();
    if (threadIdx.x < 64) {
        uint32_t t = ((threadIdx.x + 1u) & 63u) << 26;
        for (int g = 0; g < 3; ++g) if ((a.H[g] >> 26) == threadIdx.x) t = a.H[g];
        tab[threadIdx.x] = t;
    }
    __syncthreads();
    const uint32_t K = a.K;
    uint32_t H[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(H[g]) : "s"(a.H[g]));
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + (tile & ((1ull << WRAPBITS) - 1)) * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t hv[4], t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    hv[i] = __umul24(WIN(w, o + 3), K) + WIN(w, o);
                    if (i < NLUT) t[i] = hv[i] ^ *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + ((hv[i] >> 24) & 0xfcu));
                    else t[i] = min3u(hv[i] ^ H[0], hv[i] ^ H[1], hv[i] ^ H[2]);
                }
                const uint32_t acc = min(min3u(t[0], t[1], t[2]), t[3]);
                if (__ballot(acc == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const unsigned long long mm = __ballot(t[i] == 0); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
    if (threadIdx.x == 0) { if (blockIdx.x == 0) { cnt[1] = 1; cnt[2] = 1; } if (blockIdx.x < 8192) { cnt[8 + 2 * blockIdx.x] = w0; cnt[9 + 2 * blockIdx.x] = wall_clock64(); } }
}

// ---- V16: V14 with a 32-slot table (one slot per LDS bank: no bank conflicts).
// ---- (V14 text:) like V13 but the 64-slot table sits in LDS and is read with ds_read_b32 (2 VALU ops for
//      the address instead of 1, but a plain LDS read instead of the crossbar) ----------------------
template <int ROWS, int WRAPBITS = 63, int NLUT = 4>
__global__ __launch_bounds__(THREADS) void k_v16(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    __shared__ uint32_t tab[32];
    const unsigned long long w0 = wall_clock64();
    if (threadIdx.x < 32) {
        uint32_t t = ((threadIdx.x + 1u) & 31u) << 27;
        for (int g = 0; g < 3; ++g) if ((a.H[g] >> 27) == threadIdx.x) t = a.H[g];
        tab[threadIdx.x] = t;
    }
    __syncthreads();
    const uint32_t K = a.K;
    uint32_t H[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(H[g]) : "s"(a.H[g]));
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + (tile & ((1ull << WRAPBITS) - 1)) * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t hv[4], t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    hv[i] = __umul24(WIN(w, o + 3), K) + WIN(w, o);
                    if (i < NLUT) t[i] = hv[i] ^ *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + ((hv[i] >> 25) & 0x7cu));
                    else t[i] = min3u(hv[i] ^ H[0], hv[i] ^ H[1], hv[i] ^ H[2]);
                }
                const uint32_t acc = min(min3u(t[0], t[1], t[2]), t[3]);
                if (__ballot(acc == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const unsigned long long mm = __ballot(t[i] == 0); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
    if (threadIdx.x == 0) { if (blockIdx.x == 0) { cnt[1] = 1; cnt[2] = 1; } if (blockIdx.x < 8192) { cnt[8 + 2 * blockIdx.x] = w0; cnt[9 + 2 * blockIdx.x] = wall_clock64(); } }
}


// ---- V15: V14 with the 8-byte halo taken from the next lane (DPP wave_shl:1) instead of a second load; lane 63 loads it.
// ---- (V14 text:) like V13 but the 64-slot table sits in LDS and is read with ds_read_b32 (2 VALU ops for
//      the address instead of 1, but a plain LDS read instead of the crossbar) ----------------------
template <int ROWS, int WRAPBITS = 63, int NLUT = 4>
__global__ __launch_bounds__(THREADS) void k_v15(const uint8_t *__restrict__ buf, Args a, uint64_t ntiles, unsigned long long *cnt) {
    __shared__ uint32_t tab[64];
    const unsigned long long w0 = wall_clock64();
    if (threadIdx.x < 64) {
        uint32_t t = ((threadIdx.x + 1u) & 63u) << 26;
        for (int g = 0; g < 3; ++g) if ((a.H[g] >> 26) == threadIdx.x) t = a.H[g];
        tab[threadIdx.x] = t;
    }
    __syncthreads();
    const uint32_t K = a.K;
    uint32_t H[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("v_mov_b32 %0, %1" : "=v"(H[g]) : "s"(a.H[g]));
    uint32_t qn = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint8_t *src = buf + (tile & ((1ull << WRAPBITS) - 1)) * (uint64_t)(ROWB * ROWS) + threadIdx.x * 16u;
        uint4 v[ROWS]; uint2 h[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { v[r] = *(const uint4 *)(src + r * ROWB); h[r] = uint2{0u, 0u}; if ((threadIdx.x & 63u) == 63u) h[r] = *(const uint2 *)(src + r * ROWB + 16); }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            // lane l takes v.x, v.y of lane l + 1 (lane 63 keeps its loaded halo: bound_ctrl off, old value kept)
            h[r].x = (uint32_t)__builtin_amdgcn_update_dpp((int)h[r].x, (int)v[r].x, 0x130, 0xf, 0xf, false);
            h[r].y = (uint32_t)__builtin_amdgcn_update_dpp((int)h[r].y, (int)v[r].y, 0x130, 0xf, 0xf, false);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const uint32_t w[6] = {v[r].x, v[r].y, v[r].z, v[r].w, h[r].x, h[r].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t hv[4], t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * j + i;
                    hv[i] = __umul24(WIN(w, o + 3), K) + WIN(w, o);
                    if (i < NLUT) t[i] = hv[i] ^ *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + ((hv[i] >> 24) & 0xfcu));
                    else t[i] = min3u(hv[i] ^ H[0], hv[i] ^ H[1], hv[i] ^ H[2]);
                }
                const uint32_t acc = min(min3u(t[0], t[1], t[2]), t[3]);
                if (__ballot(acc == 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const unsigned long long mm = __ballot(t[i] == 0); if (mm) rare(mm, qn); }
                }
            }
        }
    }
    if ((threadIdx.x & 63) == 0 && qn) atomicAdd(cnt, (unsigned long long)qn);
    if (threadIdx.x == 0) { if (blockIdx.x == 0) { cnt[1] = 1; cnt[2] = 1; } if (blockIdx.x < 8192) { cnt[8 + 2 * blockIdx.x] = w0; cnt[9 + 2 * blockIdx.x] = wall_clock64(); } }
}


// semantic probe of the (m)qsad instructions
__global__ void k_probe(const uint64_t *s0, const uint32_t *s1, const uint64_t *s2, uint64_t *out_q, uint64_t *out_m) {
    const int i = threadIdx.x;
    out_q[i] = __builtin_amdgcn_qsad_pk_u16_u8(s0[i], s1[i], s2[i]);
    out_m[i] = __builtin_amdgcn_mqsad_pk_u16_u8(s0[i], s1[i], s2[i]);
}

static uint32_t le32(const uint8_t *p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }

template <class F>
void run(const char *name, F launch, unsigned long long *d_cnt, uint64_t n, int rows) {
    if (const char *only = getenv("FV_ONLY")) {          // ';'-separated substrings
        bool hit = false; std::string o(only); size_t p0 = 0;
        while (p0 <= o.size()) { size_t p1 = o.find(';', p0); if (p1 == std::string::npos) p1 = o.size();
            if (p1 > p0 && strstr(name, o.substr(p0, p1 - p0).c_str())) hit = true; p0 = p1 + 1; }
        if (!hit) return;
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    unsigned long long h_cnt = 0, h_clk[3] = {0, 0, 0};
    static std::vector<unsigned long long> h_t(8 + 2 * 8192), h_id(8192);
    for (int it = 0; it < 7; ++it) {
        CHECK(hipMemset(d_cnt, 0, 64 + 8192 * 16 + 64 * 64));
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float t; CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
        CHECK(hipMemcpy(h_clk, d_cnt, 24, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h_t.data(), d_cnt, (8 + 2 * 8192) * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h_id.data(), d_cnt + 8 + 2 * 8192 + 512, 8192 * 8, hipMemcpyDeviceToHost));
        h_cnt = h_clk[0];
    }
    std::sort(ms.begin(), ms.end());
    printf("%-28s rows=%d  min %.4f ms  med %.4f ms  -> %.0f GB/s (min)  hits=%llu\n", name, rows, ms[0], ms[ms.size() / 2],
           n / (ms[0] * 1e-3) / 1e9, h_cnt);
    if (h_clk[2]) {
        unsigned long long t0 = ~0ull, t1 = 0; int nwg = 0;
        for (int b = 0; b < 8192; ++b) if (h_t[9 + 2 * b]) { ++nwg; t0 = std::min(t0, h_t[8 + 2 * b]); t1 = std::max(t1, h_t[9 + 2 * b]); }
        std::vector<unsigned long long> dur, st;
        for (int b = 0; b < 8192; ++b) if (h_t[9 + 2 * b]) { dur.push_back(h_t[9 + 2 * b] - h_t[8 + 2 * b]); st.push_back(h_t[8 + 2 * b] - t0); }
        std::sort(dur.begin(), dur.end()); std::sort(st.begin(), st.end());
        printf("    %d WGs, span %.1f us; WG duration min/med/max %.1f/%.1f/%.1f us; start time med/p90/max %.1f/%.1f/%.1f us; active WGs at 10%%..90%% of span:", nwg, (t1 - t0) / 100.0,
               dur[0] / 100.0, dur[dur.size() / 2] / 100.0, dur.back() / 100.0, st[st.size() / 2] / 100.0, st[st.size() * 9 / 10] / 100.0, st.back() / 100.0);
        for (int q = 1; q < 10; q += 2) { const unsigned long long t = t0 + (t1 - t0) * q / 10; int act = 0;
            for (int b = 0; b < 8192; ++b) if (h_t[9 + 2 * b] && h_t[8 + 2 * b] <= t && h_t[9 + 2 * b] > t) ++act; printf(" %d", act); }
        printf("\n");
        if (getenv("FV_XCD")) {
            double sum[8] = {0}, endmax[8] = {0}; int nx[8] = {0};
            for (int b = 0; b < 8192; ++b) if (h_t[9 + 2 * b]) { const int x = (int)((h_id[b] >> 32) & 7);
                sum[x] += (h_t[9 + 2 * b] - h_t[8 + 2 * b]) / 100.0; endmax[x] = std::max(endmax[x], (h_t[9 + 2 * b] - t0) / 100.0); ++nx[x]; }
            printf("    per XCD (WGs, mean WG duration us, last end us):");
            for (int x = 0; x < 8; ++x) printf("  %d: %d %.0f %.0f |", x, nx[x], nx[x] ? sum[x] / nx[x] : 0.0, endmax[x]);
            printf("\n    blockIdx 0..15 -> xcc:"); for (int b = 0; b < 16; ++b) printf(" %d", (int)((h_id[b] >> 32) & 7));
            printf("  hw_id[0..3]: %08x %08x %08x %08x\n", (unsigned)h_id[0], (unsigned)h_id[1], (unsigned)h_id[8], (unsigned)h_id[16]);
        }
    }
    if (h_clk[2]) printf("    WG0: %llu core cycles / %llu ticks@100MHz -> %.0f MHz\n", h_clk[1], h_clk[2], (double)h_clk[1] / h_clk[2] * 100.0);
}

int main(int argc, char **argv) {
    const uint64_t n = (argc > 1 ? strtoull(argv[1], 0, 10) : 1024ull) << 20;
    uint8_t *buf;
    CHECK(hipMalloc((void **)&buf, n + (1 << 20)));
    CHECK(hipMemset(buf, 0, n + (1 << 20)));
    hipLaunchKernelGGL(gen_dna, dim3(4096), dim3(256), 0, 0, buf, n, 12345ull);
    CHECK(hipDeviceSynchronize());
    unsigned long long *d_cnt;
    CHECK(hipMalloc((void **)&d_cnt, 64 + 8192 * 16 + 64 * 64 + 8192 * 8));
    const uint8_t pat[21] = "GATTACAGATTACACCGTTA";
    Args a{};
    a.K = 0x9E3779u; a.d2 = 2; a.dh = 3;
    for (int g = 0; g < 3; ++g) {
        a.A[g] = le32(pat + 6 * g);
        a.B[g] = le32(pat + 6 * g + 2);
        a.H[g] = (le32(pat + 6 * g + 3) & 0xffffffu) * a.K + a.A[g];
        { const uint32_t p0 = le32(pat + 6 * g) & 0xffffu, p1 = le32(pat + 6 * g + 2) & 0xffffu, p2 = le32(pat + 6 * g + 4) & 0xffffu;
          const uint32_t h16 = ((p0 + (a.K & 0xffffu) * p1) & 0xffffu) ^ p2; a.HP[g] = h16 * 0x10001u; }
    }
    int cus = 256;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, n = %llu MiB\n", prop.gcnArchName, cus, (unsigned long long)(n >> 20));
#define RUN(NAME, KERN, ROWS, GRIDMUL) { const uint64_t nt = n / (ROWB * ROWS); dim3 grid((unsigned)std::min<uint64_t>(nt, (uint64_t)cus * GRIDMUL)); \
    run(NAME, [&]() { hipLaunchKernelGGL(KERN, grid, dim3(THREADS), 0, 0, buf, a, nt, d_cnt); }, d_cnt, n, ROWS); }
    RUN("stream r4 g8", (k_stream<4>), 4, 8);
    RUN("stream r8 g8", (k_stream<8>), 8, 8);
    RUN("stream r2 g8", (k_stream<2>), 2, 8);
    RUN("stream r4 g16", (k_stream<4>), 4, 16);
    RUN("v0 exact s_and/s_or", (k_v0<4, 2>), 4, 8);
    RUN("v1 hash cmp s_or br/1", (k_v1<4, 3>), 4, 8);
    RUN("v2 hash cmp s_or br/4", (k_v2<4, 3>), 4, 8);
    RUN("v3 hash xor min3 grp4", (k_v3<4, 3, 4>), 4, 8);
    RUN("v3 hash xor min3 grp8", (k_v3<4, 3, 8>), 4, 8);
    RUN("v3 hash xor min3 grp16", (k_v3<4, 3, 16>), 4, 8);
    RUN("v5 hash cmp addc grp4", (k_v5<4, 3, 4>), 4, 8);
    RUN("v5 hash cmp addc grp8", (k_v5<4, 3, 8>), 4, 8);
    {   // B for V6: n-gram bytes 4..5, zero (masked) beyond L = 6
        Args a6 = a;
        for (int g = 0; g < 3; ++g) a6.B[g] = pat[6 * g + 4] | pat[6 * g + 5] << 8;
        Args keep = a; a = a6;
        RUN("v6 qsad+mqsad", (k_v6<4>), 4, 8);
        RUN("v6 qsad+mqsad r8", (k_v6<8>), 8, 8);
        a = keep;
    }
    {   // probe semantics
        const int N = 64;
        uint64_t hs0[N], hs2[N], hq[N], hm[N]; uint32_t hs1[N];
        uint64_t z = 88172645463325252ull;
        for (int i = 0; i < N; ++i) {
            z ^= z << 13; z ^= z >> 7; z ^= z << 17; hs0[i] = z;
            z ^= z << 13; z ^= z >> 7; z ^= z << 17; hs1[i] = (uint32_t)z;
            z ^= z << 13; z ^= z >> 7; z ^= z << 17; hs2[i] = z & 0x0fff0fff0fff0fffull;
            if (i % 4 == 1) hs1[i] &= 0xffff00ffu;          // zero byte in reference
            if (i % 4 == 2) hs0[i] &= 0xffffffff00ffff00ull; // zero bytes in data
            if (i % 8 == 3) hs1[i] &= 0x0000ffffu;
        }
        uint64_t *d0, *d2, *dq, *dm; uint32_t *d1;
        CHECK(hipMalloc((void **)&d0, N * 8)); CHECK(hipMalloc((void **)&d2, N * 8)); CHECK(hipMalloc((void **)&dq, N * 8));
        CHECK(hipMalloc((void **)&dm, N * 8)); CHECK(hipMalloc((void **)&d1, N * 4));
        CHECK(hipMemcpy(d0, hs0, N * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d1, hs1, N * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d2, hs2, N * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(N), 0, 0, d0, d1, d2, dq, dm);
        CHECK(hipMemcpy(hq, dq, N * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hm, dm, N * 8, hipMemcpyDeviceToHost));
        int bad_q = 0, bad_m_ref = 0, bad_m_data = 0;
        for (int i = 0; i < N; ++i) {
            uint64_t eq = 0, em_ref = 0, em_data = 0;
            for (int f = 0; f < 4; ++f) {
                uint32_t sq = 0, smr = 0, smd = 0;
                for (int b = 0; b < 4; ++b) {
                    const int db = (hs0[i] >> (8 * (f + b))) & 0xff, rb = (hs1[i] >> (8 * b)) & 0xff;
                    const int ad = db > rb ? db - rb : rb - db;
                    sq += ad; if (rb != 0) smr += ad; if (db != 0) smd += ad;
                }
                const uint32_t acc = (hs2[i] >> (16 * f)) & 0xffff;
                eq |= (uint64_t)((sq + acc) & 0xffff) << (16 * f);
                em_ref |= (uint64_t)((smr + acc) & 0xffff) << (16 * f);
                em_data |= (uint64_t)((smd + acc) & 0xffff) << (16 * f);
            }
            bad_q += eq != hq[i]; bad_m_ref += em_ref != hm[i]; bad_m_data += em_data != hm[i];
        }
        printf("probe: qsad mismatches %d/64; mqsad vs mask-on-REFERENCE-zero %d/64; vs mask-on-DATA-zero %d/64\n", bad_q, bad_m_ref, bad_m_data);
        for (int i = 0; i < 4; ++i) printf("  s0=%016llx s1=%08x s2=%016llx q=%016llx m=%016llx\n", (unsigned long long)hs0[i], hs1[i], (unsigned long long)hs2[i], (unsigned long long)hq[i], (unsigned long long)hm[i]);
    }
    RUN("v9 prefetch r4 g8", (k_v9<4, 3>), 4, 8);
    RUN("v9 prefetch r4 g6", (k_v9<4, 3>), 4, 6);
    RUN("v9 prefetch r4 g4", (k_v9<4, 3>), 4, 4);
    RUN("v9 prefetch r2 g8", (k_v9<2, 3>), 2, 8);
    RUN("v9 prefetch r2 g16", (k_v9<2, 3>), 2, 16);
    RUN("v9 prefetch r8 g4", (k_v9<8, 3>), 8, 4);
#define RUNL(NAME, KERN, ROWS, GRIDMUL, LDS) { const uint64_t nt = n / (ROWB * ROWS); dim3 grid((unsigned)std::min<uint64_t>(nt, (uint64_t)cus * GRIDMUL)); \
    hipFuncSetAttribute(reinterpret_cast<const void *>(&KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    run(NAME, [&]() { hipLaunchKernelGGL(KERN, grid, dim3(THREADS), LDS, 0, buf, a, nt, d_cnt); }, d_cnt, n, ROWS); }
    RUNL("stream r4, 4 WG/CU (40K LDS)", (k_stream<4>), 4, 4, 40 * 1024);
    RUNL("stream r4, 2 WG/CU (80K LDS)", (k_stream<4>), 4, 2, 80 * 1024);
    RUNL("stream r4, 1 WG/CU (159K LDS)", (k_stream<4>), 4, 1, 159 * 1024);
    RUNL("stream r8, 2 WG/CU (80K LDS)", (k_stream<8>), 8, 2, 80 * 1024);
    RUNL("stream r8, 1 WG/CU (159K LDS)", (k_stream<8>), 8, 1, 159 * 1024);
    RUNL("stream r4, 6 WG/CU (26K LDS)", (k_stream<4>), 4, 6, 26 * 1024);
    { Args keep = a;
      for (int C : {1, 2, 4, 8, 16}) for (int P : {64, 8}) { a.d2 = C; a.dh = P; char nm[64]; snprintf(nm, sizeof nm, "v11 dynamic 3blk C=%d P=%d g8", C, P); RUN(nm, (k_v11<3>), 4, 8); }
      a.d2 = 4; a.dh = 64; RUN("v11 dynamic 3blk C=4 P=64 g6", (k_v11<3>), 4, 6);
      a.d2 = 4; a.dh = 64; RUN("v11 dynamic 1blk C=4 P=64 g8", (k_v11<1>), 4, 8);
      a.d2 = 4; a.dh = 64; RUN("v11 dynamic 3blk C=4 P=64 g8 L2res", (k_v11<3, 8>), 4, 8);
      a = keep; }
#define RUNT(NAME, KERN) { const uint64_t nt = n / (ROWB * 4); dim3 grid((unsigned)cus * 8); CHECK(hipMemset(d_cnt, 0, 64 + 8192 * 16)); \
      hipLaunchKernelGGL(KERN, grid, dim3(THREADS), 0, 0, buf, a, nt, d_cnt); CHECK(hipDeviceSynchronize()); CHECK(hipMemset(d_cnt, 0, 64 + 8192 * 16)); \
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); hipLaunchKernelGGL(KERN, grid, dim3(THREADS), 0, 0, buf, a, nt, d_cnt); hipEventRecord(e1); CHECK(hipDeviceSynchronize()); \
      float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long hc[8 + 32]; CHECK(hipMemcpy(hc, d_cnt, sizeof hc, hipMemcpyDeviceToHost)); \
      printf("%-34s %.4f ms; per tile per wave-0 of a WG, by dispatch-order class (core cycles): issue / wait-row0 / compute\n", NAME, ms); \
      for (int c = 0; c < 8; ++c) { const double nn = (double)hc[8 + c * 4 + 3]; if (nn > 0) printf("   class %d: %6.0f / %6.0f / %6.0f   (tiles %.0f)\n", c, hc[8 + c * 4] / nn, hc[8 + c * 4 + 1] / nn, hc[8 + c * 4 + 2] / nn, nn); } }
    if (!getenv("FV_ONLY")) {
    RUNT("v12 timeline 1 block HBM", (k_v12<1>));
    RUNT("v12 timeline 1 block L2-res", (k_v12<1, 6>));
    RUNT("v12 timeline 3 blocks HBM", (k_v12<3>));
    RUNT("v12 timeline 3 blocks L2-res", (k_v12<3, 6>));
    }
    printf("slots of the 3 block hashes: %u %u %u\n", a.H[0] >> 26, a.H[1] >> 26, a.H[2] >> 26);
    RUN("v13 bpermute LUT r4 g8", (k_v13<4>), 4, 8);
    RUN("v13 bpermute LUT r4 g8 L2res", (k_v13<4, 6>), 4, 8);
    RUN("v14 ds_read LUT 4/4 r4 g8", (k_v14<4, 63, 4>), 4, 8);
    printf("32-slot indices of the 3 block hashes: %u %u %u\n", a.H[0] >> 27, a.H[1] >> 27, a.H[2] >> 27);
    RUN("v16 32-slot LUT r4 g8", (k_v16<4, 63, 4>), 4, 8);
    RUN("v16 32-slot LUT r4 g8 L2res", (k_v16<4, 6, 4>), 4, 8);
    RUN("v15 LUT + DPP halo r4 g8", (k_v15<4, 63, 4>), 4, 8);
    RUN("v15 LUT + DPP halo r4 g8 L2res", (k_v15<4, 6, 4>), 4, 8);
    RUN("v15 LUT + DPP halo r8 g8", (k_v15<8, 63, 4>), 8, 8);
    RUN("v14 ds_read LUT 4/4 r4 g8 L2res", (k_v14<4, 6, 4>), 4, 8);
    RUN("v14 ds_read LUT 3/4 r4 g8", (k_v14<4, 63, 3>), 4, 8);
    RUN("v14 ds_read LUT 3/4 r4 g8 L2res", (k_v14<4, 6, 3>), 4, 8);
    RUN("v14 ds_read LUT 2/4 r4 g8", (k_v14<4, 63, 2>), 4, 8);
    RUN("v14 ds_read LUT 2/4 r4 g8 L2res", (k_v14<4, 6, 2>), 4, 8);
    RUN("v14 ds_read LUT 4/4 r8 g8", (k_v14<8, 63, 4>), 8, 8);
    RUN("v13 hybrid 3/4 LUT r4 g8", (k_v13<4, 63, 3>), 4, 8);
    RUN("v13 hybrid 2/4 LUT r4 g8", (k_v13<4, 63, 2>), 4, 8);
    RUN("v13 hybrid 1/4 LUT r4 g8", (k_v13<4, 63, 1>), 4, 8);
    RUN("v13 hybrid 2/4 LUT r4 g8 L2res", (k_v13<4, 6, 2>), 4, 8);
    RUN("v13 hybrid 1/4 LUT r4 g8 L2res", (k_v13<4, 6, 1>), 4, 8);
    RUN("v13 hybrid 2/4 LUT r8 g8", (k_v13<8, 63, 2>), 8, 8);
    RUN("v13 bpermute LUT r8 g8", (k_v13<8>), 8, 8);
    RUN("v13 bpermute LUT r4 g6", (k_v13<4>), 4, 6);
    RUN("v3 r4 g8 H-in-VGPR", (k_v3<4, 3, 4, 63, true>), 4, 8);
    RUN("v3 r4 g8 HV 1 block contiguous-per-WG", (k_v3<4, 3, 4, 63, true, 1, 1>), 4, 8);
    RUN("v3 r4 g8 HV 1 block wave-contig rows", (k_v3<4, 3, 4, 63, true, 1, 2>), 4, 8);
    RUN("v3 r4 g8 HV 3 block contiguous-per-WG", (k_v3<4, 3, 4, 63, true, 3, 1>), 4, 8);
    RUN("v3 r4 g8 HV 3 block wave-contig rows", (k_v3<4, 3, 4, 63, true, 3, 2>), 4, 8);
    RUN("v3 r4 g8 HV 1 block prio-by-progress", (k_v3<4, 3, 4, 63, true, 1, 3>), 4, 8);
    RUN("v3 r4 g8 HV 3 block prio-by-progress", (k_v3<4, 3, 4, 63, true, 3, 3>), 4, 8);
    RUN("v3 r4 g8 HV 1 block prio-reverse", (k_v3<4, 3, 4, 63, true, 1, 4>), 4, 8);
    RUN("v3 r4 g8 HV 3 block prio-reverse", (k_v3<4, 3, 4, 63, true, 3, 4>), 4, 8);
    RUN("v3 r4 g16 HV 3 block prio-by-progress", (k_v3<4, 3, 4, 63, true, 3, 3>), 4, 16);
    RUN("v3 r4 g32 HV 3 block prio-by-progress", (k_v3<4, 3, 4, 63, true, 3, 3>), 4, 32);
    RUN("v3 r4 g32 HV 3 block", (k_v3<4, 3, 4, 63, true, 3, 0>), 4, 32);
    RUN("v3 r4 g8 HV 2 blocks", (k_v3<4, 3, 4, 63, true, 2>), 4, 8);
    RUN("v3 r4 g8 HV 1 block", (k_v3<4, 3, 4, 63, true, 1>), 4, 8);
    RUN("v3 r4 g8 HV 1 block L2res", (k_v3<4, 3, 4, 6, true, 1>), 4, 8);
    RUN("v3 r8 g8 HV 1 block", (k_v3<8, 3, 4, 63, true, 1>), 8, 8);
    RUN("v3 r4 g8 H-in-VGPR L2-res", (k_v3<4, 3, 4, 6, true>), 4, 8);
    RUN("v3 r8 g8 H-in-VGPR", (k_v3<8, 3, 4, 63, true>), 8, 8);
    RUN("v10 packed16 r4 grp4", (k_v10<4, 4>), 4, 8);
    RUN("v10 packed16 r4 grp8", (k_v10<4, 8>), 4, 8);
    RUN("v10 packed16 r4 grp4 L2-res", (k_v10<4, 4, 6>), 4, 8);
    RUN("v10 packed16 r8 grp4", (k_v10<8, 4>), 8, 8);
    RUN("v10 packed16 r4 grp4 g6", (k_v10<4, 4>), 4, 6);
    RUN("v3 r4 g8 L2-resident (1MB)", (k_v3<4, 3, 4, 6>), 4, 8);
    RUN("v3 r4 g8 L2-resident (256K)", (k_v3<4, 3, 4, 4>), 4, 8);
    RUN("v3 r4 g8 MALL-resident (64MB)", (k_v3<4, 3, 4, 12>), 4, 8);
    RUN("stream r4 g8 again", (k_stream<4>), 4, 8);
    RUN("v3 r4 g6", (k_v3<4, 3, 4>), 4, 6);
    RUN("v3 r4 g4", (k_v3<4, 3, 4>), 4, 4);
    RUN("v8 shifted loads r2", (k_v8<2, 3>), 2, 8);
    RUN("v8 shifted loads r1", (k_v8<1, 3>), 1, 8);
    RUN("v8 shifted loads r2 g16", (k_v8<2, 3>), 2, 16);
    RUN("v7 lds15 4/4 groups", (k_v7<4, 3, 15, 4>), 4, 8);
    RUN("v7 lds15 3/4 groups", (k_v7<4, 3, 15, 3>), 4, 8);
    RUN("v7 lds15 2/4 groups", (k_v7<4, 3, 15, 2>), 4, 8);
    RUN("v7 lds15 1/4 groups", (k_v7<4, 3, 15, 1>), 4, 8);
    RUN("v7 lds14 3/4 groups", (k_v7<4, 3, 14, 3>), 4, 8);
    RUN("v7 lds14 2/4 groups", (k_v7<4, 3, 14, 2>), 4, 8);
    RUN("v7 lds13 2/4 groups", (k_v7<4, 3, 13, 2>), 4, 8);
    RUN("v7 lds15 2/4 g16", (k_v7<4, 3, 15, 2>), 4, 16);
    RUN("v1 r2", (k_v1<2, 3>), 2, 8);
    RUN("v1 r8", (k_v1<8, 3>), 8, 8);
    RUN("v3 grp4 r2", (k_v3<2, 3, 4>), 2, 8);
    RUN("v3 grp4 r8", (k_v3<8, 3, 4>), 8, 8);
    RUN("v1 r4 g16", (k_v1<4, 3>), 4, 16);
    RUN("v3 grp4 r4 g16", (k_v3<4, 3, 4>), 4, 16);
    return 0;
}
