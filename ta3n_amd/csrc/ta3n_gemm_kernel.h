// Device code of the tile-list GEMM (included by ta3n_gemm.hip, which launches it, and by ta3n_gemm_i*.hip, which instantiate it -
// the instantiations are split over several translation units so that they compile in parallel).
#pragma once
// Tile-list fp32 GEMM for gfx950 (CDNA4) on the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32: 64 cycles/SIMD, an fmaf chain per element, 157 TF peak).
//
// One launch = one dependency level of the TA3N train step.  Each workgroup
// (NW = WM*WN*WK wave64, 4 or 8) takes one Task: a (32*WM x 32*WN) output tile
// whose K loop runs over a list of Segs.  A Seg is an affine view
//     A(r,k) = base_a[a_off + (kmajor ? k*a_ld + r : r*a_ld + k)]
// so the same kernel does  X W^T (forward), G W (input gradients), G^T X (weight
// gradients), the TRN frame-tuple gather+concat (one Seg per tuple position,
// reference TRNmodule.py:60-63/75-77 - never materialised), the scatter-free
// TRN input gradient (one Seg per (tuple,position) that contains the frame) and
// GradReverse (reference models.py:20-29) as a "scale the accumulator by -beta
// after this Seg" flag.  WK > 1 splits every 64-deep K chunk across the
// workgroup's waves so small outputs still occupy all SIMDs of a CU.
//
// Data path per 64-deep chunk ("stage"):
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4: each wave instruction moves
//   64 lanes x 16 B to 1 KiB of consecutive LDS bytes; no staging VGPRs, no
//   ds_write pass).  Two stages are resident: the DMA of chunk c+1 is in flight
//   while the MFMAs of chunk c run; one s_barrier per chunk.
//   Stage image of a K-contiguous operand: [row][16 slots of 16 B], slot s of row r
//   holds k-group s ^ (r & 15) (the swizzle is applied to the per-lane SOURCE
//   address, the LDS side stays lane-linear) -> one conflict-free ds_read_b128 per
//   operand feeds 4 MFMAs.
//   Stage image of a k-major operand: [k][R] linear -> conflict-free ds_read_b32.
//   Within an 8-deep k group MFMA j pairs k = j (lanes 0-31) with k = 4 + j (lanes
//   32-63) for BOTH operands, so a b128 read per half-wave supplies four MFMAs.
//   Operands that cannot be moved 16 bytes at a time (odd leading dimension, K = 2,
//   num_class not a multiple of 4, ...) use the 4-byte LDS-DMA form with the same
//   stage image; out-of-range elements are read from a block of zeros (the validity
//   test selects the ADDRESS), so K tails and ragged row counts need no branches.
//   The LDS-DMA is issued from inline asm: a compiler-visible
//   __builtin_amdgcn_global_load_lds makes hipcc (ROCm 7.2) place s_waitcnt vmcnt(0)
//   in front of every later ds_read, which serialises load and compute.
// The epilogue goes through LDS once more so the K-split partials are reduced
// and the stores / bias / mask operands are row-contiguous float4s.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "ta3n_kernels.h"

using namespace ta3n;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int BKC = 64;        // K chunk per stage

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// LDS-DMA, 16 or 4 bytes per lane.  lds_byte_addr must be wave-uniform (it goes to M0);
// M0 is compiler-reserved, so it is saved and restored inside the statement.
__device__ __forceinline__ void glds16(const float *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// NP 16-byte LDS-DMAs of one wave, destinations lds_byte_addr + i * stride: one M0 save/restore for the batch.
template <int NP>
__device__ __forceinline__ void glds16_batch(const float *const (&src)[NP], unsigned lds_byte_addr, unsigned stride) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);   // wave-uniform by construction; makes it provably so
    if constexpr (NP == 1) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src[0]), "s"(lds_byte_addr) : "memory");
    } else if constexpr (NP == 2) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "s"(lds_byte_addr), "s"(stride) : "memory", "scc");
    } else if constexpr (NP == 8) {
        const float *const lo[4] = {src[0], src[1], src[2], src[3]}, *const hi[4] = {src[4], src[5], src[6], src[7]};
        glds16_batch<4>(lo, lds_byte_addr, stride);
        glds16_batch<4>(hi, lds_byte_addr + 4 * stride, stride);
    } else if constexpr (NP == 3) {
        const float *const lo[2] = {src[0], src[1]}, *const hi[1] = {src[2]};
        glds16_batch<2>(lo, lds_byte_addr, stride);
        glds16_batch<1>(hi, lds_byte_addr + 2 * stride, stride);
    } else if constexpr (NP == 6) {
        const float *const lo[4] = {src[0], src[1], src[2], src[3]}, *const hi[2] = {src[4], src[5]};
        glds16_batch<4>(lo, lds_byte_addr, stride);
        glds16_batch<2>(hi, lds_byte_addr + 4 * stride, stride);
    } else {
        static_assert(NP == 4, "1, 2, 3, 4, 6 or 8 pieces per wave");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                     "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                     "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "s"(lds_byte_addr), "s"(stride)
                     : "memory", "scc");
    }
}
__device__ __forceinline__ void glds4(const float *gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// 4-byte LDS-DMA path for operands that cannot be moved 16 bytes at a time (256 B pieces, one element per
// lane).  Out of line on purpose: inlined, its address arithmetic costs every tile ~22 VGPRs, i.e. one
// resident workgroup per CU; only a handful of tiny tasks take this path.
template <int R, int NW>
__device__ __attribute__((noinline)) void issue_slow(const float *__restrict__ origin, int ld, int kmajor, int r0, int rvalid, int k0,
                                                     int krem, unsigned lds_addr, int wave, int lane, const float *__restrict__ zeros) {
#pragma unroll 1
    for (int q = wave; q < R; q += NW) {
        const float *src;
        if (!kmajor) {                             // piece = one row; lane -> (slot, element)
            const int k = 4 * ((lane >> 2) ^ (q & 15)) + (lane & 3);
            src = (r0 + q < rvalid && k < krem) ? origin + (size_t)(r0 + q) * ld + k0 + k : zeros;
        } else {
            constexpr int KPP = 64 / R;            // k rows per piece (2 for R = 32, 1 for R = 64)
            const int k = q * KPP + lane / R;
            const int r = r0 + lane % R;
            src = (r < rvalid && k < krem) ? origin + (size_t)(k0 + k) * ld + r : zeros;
        }
        glds4(src, lds_addr + q * 256);
    }
}

// Per-lane LDS-DMA state of one operand for the Seg being streamed.  Set up once per
// Seg (the 64-bit address arithmetic lives there); per 64-deep chunk the fast path is
// compare + select + DMA + pointer bump per 1 KiB piece.  R rows (32 or 64), NW waves.
// All branches are wave-uniform.
constexpr int K_NEVER = 1 << 28;   // "k offset" of a lane whose row is out of range: never < remaining K

// TW: the operand is a bf16 twin (origin points into the twin region; ld, klen, rows in ELEMENTS).  The stage holds
// 128 k: K-contiguous rows are 16 slots of 8 bf16, a k-major stage is [128][R] bf16 - the same bytes per stage, the
// same number of 1 KiB pieces.  The plan only marks launches whose every operand moves 16 bytes at a time.
// PAIR (pair twins, Geom::pair_delta): the stage holds 64 k twice - the hi plane and, pair_delta floats behind it in memory, the lo
// plane (x = hi + lo).  K-contiguous: logical slots 0-7 of a row are the hi k-groups, 8-15 the lo ones; k-major: image rows 0-63
// are the hi plane's k rows, 64-127 the lo plane's.  Same bytes, pieces and DMA instructions per stage as the plain twin stage.
// HS ("half stages", bf16 twins only): the stage holds 64 k in HALF the bytes - K-contiguous rows are 8 slots of 8 bf16 (128-byte rows: a
// 1 KiB piece is 8 rows; slot s of row r holds k-group s ^ ((r >> 1) & 7), conflict-free for ds_read_b128: the 16 lanes of a read group
// cover all sixteen 16-byte bank quads), a k-major stage is [64][R] bf16.  Half the bytes per stage = twice the stages in the same LDS:
// a 128x128 tile keeps four (three in flight) where it kept two (one in flight) - the K loops wait for LDS-DMA round trips, not for
// bandwidth (profiles/r04_pmc_per_launch.txt).
template <int R, int NW, bool TW = false, bool PAIR = false, bool HS = false>
struct OperandStream {
    static constexpr int NP = HS ? R / 8 / NW : R / 4 / NW;   // 1 KiB pieces per wave per stage (16-byte path)
    static_assert(HS ? (R / 8) % NW == 0 && TW && !PAIR : (R / 4) % NW == 0, "pieces must divide over the waves");
    const float *p[NP];     // this lane's source address in the current chunk
    int kofs[NP];           // this lane's k offset inside a chunk (K_NEVER: row out of range)
    int step;               // floats between consecutive chunks
    // slow (4-byte) path
    const float *origin;
    int ld, kmajor, r0, rvalid;
    bool vec;

    __device__ __forceinline__ void setup(const float *__restrict__ origin_, int ld_, int kmajor_, int klen, int r0_, int rvalid_,
                                          int wave, int lane, int pair_delta = 0) {
        origin = origin_; ld = ld_; kmajor = kmajor_; r0 = r0_; rvalid = rvalid_;
        if constexpr (TW && PAIR) {
            vec = true;
            const int ldf = ld_ >> 1;                  // floats per row of a twin plane
            step = kmajor_ ? 64 * ldf : BKC / 2;       // 64 k per stage
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int q = wave + NW * i;
                if (!kmajor_) {
                    const int row = q * 4 + (lane >> 4);
                    const int slot = (lane & 15) ^ (row & 15);            // logical slot held at physical slot (lane & 15)
                    p[i] = origin_ + (size_t)(r0_ + row) * ldf + 4 * (slot & 7) + (slot >= 8 ? pair_delta : 0);
                    kofs[i] = (r0_ + row < rvalid_) ? 8 * (slot & 7) : K_NEVER;
                } else {
                    constexpr int LPR = R / 8, KPP = 64 / LPR;
                    const int kk = q * KPP + lane / LPR;                  // image row: 0-63 hi plane, 64-127 lo plane
                    const int chunk = (lane % LPR) ^ (R == 64 ? ((kk >> 1) & 1) << 2 : R == 128 ? (kk & 3) << 2 : 0);
                    const int r = r0_ + chunk * 8;
                    p[i] = origin_ + (size_t)(kk & 63) * ldf + (r >> 1) + (kk >= 64 ? pair_delta : 0);
                    kofs[i] = (r < rvalid_) ? (kk & 63) : K_NEVER;
                }
            }
            return;
        }
        if constexpr (TW && HS) {
            vec = true;
            const int ldf = ld_ >> 1;                  // floats per row of the twin
            step = kmajor_ ? 64 * ldf : BKC / 2;       // 64 k per stage
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int q = wave + NW * i;
                if (!kmajor_) {                        // piece = 8 rows x 128 B; slot s of row r holds k = 8 (s ^ ((r >> 1) & 7)) .. + 7
                    const int row = q * 8 + (lane >> 3);
                    const int slot = (lane & 7) ^ ((row >> 1) & 7);
                    p[i] = origin_ + (size_t)(r0_ + row) * ldf + 4 * slot;
                    kofs[i] = (r0_ + row < rvalid_) ? 8 * slot : K_NEVER;
                } else {                               // [64][R] bf16: the image rows and their chunk swizzle are those of the 128-k image
                    constexpr int LPR = R / 8, KPP = 64 / LPR;
                    const int k = q * KPP + lane / LPR;
                    const int chunk = (lane % LPR) ^ (R == 64 ? ((k >> 1) & 1) << 2 : R == 128 ? (k & 3) << 2 : 0);
                    const int r = r0_ + chunk * 8;
                    p[i] = origin_ + (size_t)k * ldf + (r >> 1);
                    kofs[i] = (r < rvalid_) ? k : K_NEVER;
                }
            }
            return;
        }
        if constexpr (TW) {
            vec = true;
            const int ldf = ld_ >> 1;                  // floats per row of the twin
            step = kmajor_ ? 128 * ldf : BKC;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int q = wave + NW * i;
                if (!kmajor_) {                        // piece = 4 rows x 256 B; slot s of row r holds k = 8 (s ^ (r & 15)) .. + 7
                    const int row = q * 4 + (lane >> 4);
                    const int slot = (lane & 15) ^ (row & 15);
                    p[i] = origin_ + (size_t)(r0_ + row) * ldf + 4 * slot;
                    kofs[i] = (r0_ + row < rvalid_) ? 8 * slot : K_NEVER;
                } else {                               // [k][R] bf16: R/8 lanes (of 8 rows each) per k row
                    constexpr int LPR = R / 8, KPP = 64 / LPR;
                    const int k = q * KPP + lane / LPR;
                    // R = 64: image rows k with (k >> 1) & 1 hold their 16-byte chunks swapped by four (the transpose reads of
                    // compute_stage then touch every LDS bank once); R = 128 (256-byte rows): chunks rotated by 4 (k & 3);
                    // R = 32 rows are 64 bytes and need no swizzle
                    const int chunk = (lane % LPR) ^ (R == 64 ? ((k >> 1) & 1) << 2 : R == 128 ? (k & 3) << 2 : 0);
                    const int r = r0_ + chunk * 8;
                    p[i] = origin_ + (size_t)k * ldf + (r >> 1);
                    kofs[i] = (r < rvalid_) ? k : K_NEVER;
                }
            }
            return;
        }
        // 16-byte movability (base pointers and Seg offsets of 16-byte aligned regions: API contract + plan builder)
        const int off_bits = (int)(reinterpret_cast<uintptr_t>(origin_) >> 2);
        vec = kmajor_ ? (((off_bits | ld_ | r0_ | rvalid_) & 3) == 0) : (((off_bits | ld_ | klen) & 3) == 0);
        if (!vec) return;
        step = kmajor_ ? BKC * ld_ : BKC;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = wave + NW * i;
            if (!kmajor_) {                        // piece = 4 rows x 256 B; slot s of row r holds k group s ^ (r & 15)
                const int row = q * 4 + (lane >> 4);
                const int k = 4 * ((lane & 15) ^ (row & 15));
                p[i] = origin_ + (size_t)(r0_ + row) * ld_ + k;
                kofs[i] = (r0_ + row < rvalid_) ? k : K_NEVER;
            } else {                               // [k][R]: R/4 lanes per k row
                constexpr int LPR = R / 4, KPP = 64 / LPR;
                const int k = q * KPP + lane / LPR;
                const int r = r0_ + (lane % LPR) * 4;
                p[i] = origin_ + (size_t)k * ld_ + r;
                kofs[i] = (r < rvalid_) ? k : K_NEVER;
            }
        }
    }

    // stream the chunk starting at k0 (krem = klen - k0 valid k remain) into the stage image at lds_addr
    __device__ __forceinline__ void issue(int k0, int krem, unsigned lds_addr, int wave, int lane, const float *__restrict__ zeros) {
        if (vec) {
            const float *src[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                src[i] = kofs[i] < krem ? p[i] : zeros;
                p[i] += step;
            }
            glds16_batch<NP>(src, lds_addr + wave * 1024, NW * 1024);
        } else {
            issue_slow<R, NW>(origin, ld, kmajor, r0, rvalid, k0, krem, lds_addr, wave, lane, zeros);
        }
    }
};

// MFMAs of one wave over its K slice of one stage.  sa / sb: stage images.
// FULL: all 64 k of the stage are valid (no wave-uniform skip tests).
// All fragment reads of the stage are issued before the first MFMA (the compiler then waits with
// counted lgkmcnt): one exposed LDS latency per stage instead of one per group of four MFMAs.
// RS: also accumulate the K-sum of this lane's A values (bias gradient of a weight-gradient tile; elements past the
// K tail and rows past m_valid were staged as zeros, so no masking is needed).
// BF: the operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) in registers and multiplied by the bf16 MFMA with
// fp32 accumulation; the LDS images and their reads are the fp32 ones (BASELINE configs[1] arithmetic).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


// BF == 2: the stage bytes are bf16 already (twins): a 16-byte slot is 8 consecutive k of one row, fed to the MFMA as is.
// RM x RN: 32x32 blocks per wave (rows ra + 32 i, columns rb + 32 j): every fragment read feeds RN (RM) MFMAs, and the
// workgroup's tile - what the LDS-DMA has to bring per flop - grows with it.  Twin path only.
template <int BM, int BN, int WK, bool AKM, bool BKM, bool FULL, bool RS, int BF, int RM = 1, int RN = 1>
__device__ __forceinline__ void compute_stage(f32x16 (&accs)[RM][RN], float (&rss)[RM], const float *__restrict__ sa,
                                              const float *__restrict__ sb, int ra, int rb, int wk, int lh, int krem) {
    constexpr bool HS = BF == 5;                   // bf16 twins in 64-k stages (128-byte image rows)
    constexpr int GPW = (HS ? 8 : 16) / WK;        // k groups per wave per stage (4-deep fp32 groups; 8-deep bf16 groups on the twin paths)
    constexpr int NQ = GPW / 2;
    constexpr int ROWF = HS ? BKC / 2 : BKC;       // floats per row of a K-contiguous stage image
    static_assert(GPW >= 2, "the K split leaves every wave a pair of k groups");
    static_assert(BF == 2 || BF == 5 || (RM == 1 && RN == 1), "register blocking exists on the bf16-twin path only");
    auto swz = [](int row) { return HS ? ((row >> 1) & 7) : (row & 15); };      // physical slot = logical k group ^ swz(row)
    f32x16 &acc = accs[0][0];
    float &rs = rss[0];
    if constexpr ((BF == 2 || BF == 5) && RM * RN > 1) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
        constexpr int QG = (RM + RN) * NQ <= 16 ? NQ : NQ / 2;   // k groups whose fragments are in registers together (<= 64 VGPRs)
        const int lane = lh * 32 + (ra & 31);
        const int trow = 8 * lh + ((lane >> 2) & 3);
        const int tcol = (16 * ((lane >> 4) & 1) + 4 * (lane & 3));
        const unsigned short *sa16 = reinterpret_cast<const unsigned short *>(sa);
        const unsigned short *sb16 = reinterpret_cast<const unsigned short *>(sb);
        static_assert(!AKM || BM <= 128, "k-major A images exist for 32-, 64- and 128-row tiles (the plan keeps taller tiles off such launches)");
        const int swa = BM == 64 ? 32 * ((lane >> 3) & 1) : BM == 128 ? 32 * ((lane >> 2) & 3) : 0;
        const int swb = BN == 64 ? 32 * ((lane >> 3) & 1) : BN == 128 ? 32 * ((lane >> 2) & 3) : 0;
#pragma unroll
        for (int q0 = 0; q0 < NQ; q0 += QG) {
            u32x4 ta[RM][QG], tb[RN][QG];
#pragma unroll
            for (int qq = 0; qq < QG; ++qq) {
                const int q = q0 + qq;
                const int G = wk * GPW + 2 * q + lh;
                const int k0 = 8 * (wk * GPW + 2 * q) + trow;
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    if (!AKM) {
                        ta[i][qq] = *reinterpret_cast<const u32x4 *>(sa + (ra + 32 * i) * ROWF + ((G ^ swz(ra + 32 * i)) << 2));
                    } else {
                        const int acol = ((ra & ~31) + 32 * i + tcol) ^ swa;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sa16 + k0 * BM + acol));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sa16 + (k0 + 4) * BM + acol));
                        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                        ta[i][qq] = u32x4{l2[0], l2[1], h2[0], h2[1]};
                    }
                }
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    if (!BKM) {
                        tb[j][qq] = *reinterpret_cast<const u32x4 *>(sb + (rb + 32 * j) * ROWF + ((G ^ swz(rb + 32 * j)) << 2));
                    } else {
                        const int bcol = ((rb & ~31) + 32 * j + tcol) ^ swb;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sb16 + k0 * BN + bcol));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sb16 + (k0 + 4) * BN + bcol));
                        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                        tb[j][qq] = u32x4{l2[0], l2[1], h2[0], h2[1]};
                    }
                }
            }
            if (RS) {
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int qq = 0; qq < QG; ++qq)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            rss[i] += __builtin_bit_cast(float, ta[i][qq][j] << 16) + __builtin_bit_cast(float, ta[i][qq][j] & 0xFFFF0000u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qq = 0; qq < QG; ++qq) {
                if (FULL || 8 * (wk * GPW + 2 * (q0 + qq)) < krem) {
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int j = 0; j < RN; ++j)
                            accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ta[i][qq]),
                                                                                 __builtin_bit_cast(bf16x8, tb[j][qq]), accs[i][j], 0, 0, 0);
                }
            }
        }
        return;
    }
    if constexpr (BF == 4) {
        // Pair twins: the stage holds 64 k as a hi and a lo plane (OperandStream PAIR); slot G (0-7) = k 8G .. 8G+7.  a b ~ a_lo b_hi +
        // a_hi b_lo + a_hi b_hi (the lo lo term, ~2^-16 of the product, is dropped) with fp32 accumulation in the MFMA - the arithmetic
        // of BF == 3 without splitting anything in the loop.  A wave owns 8 / WK slots: pairs of slots feed the 16-deep MFMA (one slot
        // per half-wave), a single slot (WK = 8) the 8-deep one (4 k per half-wave).
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
        constexpr int SPW = 8 / WK;                     // slots per wave
        const int lane = lh * 32 + (ra & 31);
        const unsigned short *sa16 = reinterpret_cast<const unsigned short *>(sa);
        const unsigned short *sb16 = reinterpret_cast<const unsigned short *>(sb);
        const int tcol = (16 * ((lane >> 4) & 1) + 4 * (lane & 3));
        const int acol = ((ra & ~31) + tcol) ^ (BM == 64 ? 32 * ((lane >> 3) & 1) : 0);
        const int bcol = ((rb & ~31) + tcol) ^ (BN == 64 ? 32 * ((lane >> 3) & 1) : 0);
        if constexpr (SPW >= 2) {
            constexpr int NM = SPW / 2;
            u32x4 ah[NM], al[NM], bh[NM], bl[NM];
            const int trow = 8 * lh + ((lane >> 2) & 3);
            auto rd_kc = [&](const float *img, int row, int G) {       // K-contiguous: logical slot G of `row` sits at physical slot G ^ (row & 15)
                return *reinterpret_cast<const u32x4 *>(img + row * BKC + ((G ^ (row & 15)) << 2));
            };
            auto rd_km = [&](const unsigned short *img16, int R, int k0, int col) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(img16 + k0 * R + col));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(img16 + (k0 + 4) * R + col));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                return u32x4{l2[0], l2[1], h2[0], h2[1]};
            };
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                const int G = wk * SPW + 2 * q + lh;
                const int k0 = 8 * (wk * SPW + 2 * q) + trow;
                if (!AKM) { ah[q] = rd_kc(sa, ra, G); al[q] = rd_kc(sa, ra, G + 8); }
                else { ah[q] = rd_km(sa16, BM, k0, acol); al[q] = rd_km(sa16, BM, k0 + 64, acol); }
                if (!BKM) { bh[q] = rd_kc(sb, rb, G); bl[q] = rd_kc(sb, rb, G + 8); }
                else { bh[q] = rd_km(sb16, BN, k0, bcol); bl[q] = rd_km(sb16, BN, k0 + 64, bcol); }
            }
            if (RS) {
#pragma unroll
                for (int q = 0; q < NM; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        rs += (__builtin_bit_cast(float, ah[q][j] << 16) + __builtin_bit_cast(float, ah[q][j] & 0xFFFF0000u)) +
                              (__builtin_bit_cast(float, al[q][j] << 16) + __builtin_bit_cast(float, al[q][j] & 0xFFFF0000u));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                if (FULL || 8 * (wk * SPW + 2 * q) < krem) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[q]), __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[q]), __builtin_bit_cast(bf16x8, bl[q]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[q]), __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
                }
            }
        } else {
            // one slot per wave: lanes 0-31 take k 8 wk .. + 3, lanes 32-63 k 8 wk + 4 .. + 7
            const int trow = 4 * lh + ((lane >> 2) & 3);
            auto rd_kc = [&](const float *img, int row, int G) {
                return *reinterpret_cast<const u32x2 *>(img + row * BKC + ((G ^ (row & 15)) << 2) + 2 * lh);
            };
            auto rd_km = [&](const unsigned short *img16, int R, int k0, int col) {
                return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(img16 + k0 * R + col)));
            };
            u32x2 ah, al, bh, bl;
            if (!AKM) { ah = rd_kc(sa, ra, wk); al = rd_kc(sa, ra, wk + 8); }
            else { ah = rd_km(sa16, BM, 8 * wk + trow, acol); al = rd_km(sa16, BM, 8 * wk + trow + 64, acol); }
            if (!BKM) { bh = rd_kc(sb, rb, wk); bl = rd_kc(sb, rb, wk + 8); }
            else { bh = rd_km(sb16, BN, 8 * wk + trow, bcol); bl = rd_km(sb16, BN, 8 * wk + trow + 64, bcol); }
            if (RS) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    rs += (__builtin_bit_cast(float, ah[j] << 16) + __builtin_bit_cast(float, ah[j] & 0xFFFF0000u)) +
                          (__builtin_bit_cast(float, al[j] << 16) + __builtin_bit_cast(float, al[j] & 0xFFFF0000u));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (FULL || 8 * wk < krem) {
                acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, al), __builtin_bit_cast(s16x4, bh), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bl), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, ah), __builtin_bit_cast(s16x4, bh), acc, 0, 0, 0);
            }
        }
        return;
    }
    if constexpr (BF == 2 || BF == 5) {
        // stage of 128 k (64 with half stages); slot G = k 8G .. 8G+7.  K-contiguous: one 16-byte slot read.  k-major ([k][R] bf16): two transpose
        // reads (ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 rows] block, 8 contiguous bytes per lane, and lane i
        // receives row i's four k) - k 8G .. 8G+3 and 8G+4 .. 8G+7 of this lane's row.  For R = 64 the 16-byte chunks of image
        // rows k with (k >> 1) & 1 are stored swapped by 4 chunks (OperandStream::setup), which makes the reads conflict-free.
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
        u32x4 ta[NQ], tb[NQ];
        const int lane = lh * 32 + (ra & 31);
        const int trow = 8 * lh + ((lane >> 2) & 3);                              // k row of this lane inside the 16-k block of one MFMA
        const int tcol = (16 * ((lane >> 4) & 1) + 4 * (lane & 3));                 // first of this lane's 4 contiguous rows of the block
        const unsigned short *sa16 = reinterpret_cast<const unsigned short *>(sa);
        const unsigned short *sb16 = reinterpret_cast<const unsigned short *>(sb);
        const int acol = ((ra & ~31) + tcol) ^ (BM == 64 ? 32 * ((lane >> 3) & 1) : 0);
        const int bcol = ((rb & ~31) + tcol) ^ (BN == 64 ? 32 * ((lane >> 3) & 1) : 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int G = wk * GPW + 2 * q + lh;
            if (!AKM) {
                ta[q] = *reinterpret_cast<const u32x4 *>(sa + ra * ROWF + ((G ^ swz(ra)) << 2));
            } else {
                const int k0 = 8 * (wk * GPW + 2 * q) + trow;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sa16 + k0 * BM + acol));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sa16 + (k0 + 4) * BM + acol));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                ta[q] = u32x4{l2[0], l2[1], h2[0], h2[1]};
            }
            if (!BKM) {
                tb[q] = *reinterpret_cast<const u32x4 *>(sb + rb * ROWF + ((G ^ swz(rb)) << 2));
            } else {
                const int k0 = 8 * (wk * GPW + 2 * q) + trow;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sb16 + k0 * BN + bcol));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t *)(sb16 + (k0 + 4) * BN + bcol));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                tb[q] = u32x4{l2[0], l2[1], h2[0], h2[1]};
            }
        }
        if (RS) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rs += __builtin_bit_cast(float, ta[q][j] << 16) + __builtin_bit_cast(float, ta[q][j] & 0xFFFF0000u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (FULL || 8 * (wk * GPW + 2 * q) < krem)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ta[q]), __builtin_bit_cast(bf16x8, tb[q]), acc,
                                                              0, 0, 0);
        }
        return;
    }
    float av[NQ][4], bv[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int G = wk * GPW + 2 * q + lh;       // lanes 0-31: k = 4 G .. 4 G + 3 of the lower group, lanes 32-63: the next group
        if (!AKM) {
            const float4 t = *reinterpret_cast<const float4 *>(sa + ra * BKC + ((G ^ (ra & 15)) << 2));
            av[q][0] = t.x; av[q][1] = t.y; av[q][2] = t.z; av[q][3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[q][j] = sa[(4 * G + j) * BM + ra];
        }
        if (!BKM) {
            const float4 t = *reinterpret_cast<const float4 *>(sb + rb * BKC + ((G ^ (rb & 15)) << 2));
            bv[q][0] = t.x; bv[q][1] = t.y; bv[q][2] = t.z; bv[q][3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[q][j] = sb[(4 * G + j) * BN + rb];
        }
    }
    if (RS) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) rs += (av[q][0] + av[q][1]) + (av[q][2] + av[q][3]);
    }
    __builtin_amdgcn_sched_barrier(0);             // keep the reads above: hipcc otherwise sinks each one next to its MFMA
    if constexpr (BF == 3) {
        // fp32-grade on the bf16 matrix cores: x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi
        // (the a_lo b_lo term, ~2^-16 of the product, is dropped); fp32 accumulation in the MFMA
        auto hi2 = [](float x0, float x1) { return pack_bf16(x0, x1); };
        auto lo2 = [](float x0, float x1, unsigned hi) {      // the two subtractions as one packed op (v_pk_add_f32)
            const f32x2 x = {x0, x1};
            const f32x2 h = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xFFFF0000u)};
            const f32x2 d = x - h;
            return pack_bf16(d[0], d[1]);
        };
        if (NQ % 2 == 0) {
#pragma unroll
            for (int q = 0; q < NQ; q += 2) {
                if (FULL || 4 * (wk * GPW + 2 * q) < krem) {
                    unsigned ah[4], al[4], bh[4], bl[4];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            ah[2 * h + e] = hi2(av[q + h][2 * e], av[q + h][2 * e + 1]);
                            al[2 * h + e] = lo2(av[q + h][2 * e], av[q + h][2 * e + 1], ah[2 * h + e]);
                            bh[2 * h + e] = hi2(bv[q + h][2 * e], bv[q + h][2 * e + 1]);
                            bl[2 * h + e] = lo2(bv[q + h][2 * e], bv[q + h][2 * e + 1], bh[2 * h + e]);
                        }
                    const u32x4 AH = {ah[0], ah[1], ah[2], ah[3]}, AL = {al[0], al[1], al[2], al[3]};
                    const u32x4 BH = {bh[0], bh[1], bh[2], bh[3]}, BL = {bl[0], bl[1], bl[2], bl[3]};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AL), __builtin_bit_cast(bf16x8, BH), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AH), __builtin_bit_cast(bf16x8, BL), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, AH), __builtin_bit_cast(bf16x8, BH), acc, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (FULL || 4 * (wk * GPW + 2 * q) < krem) {
                    unsigned ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        ah[e] = hi2(av[q][2 * e], av[q][2 * e + 1]); al[e] = lo2(av[q][2 * e], av[q][2 * e + 1], ah[e]);
                        bh[e] = hi2(bv[q][2 * e], bv[q][2 * e + 1]); bl[e] = lo2(bv[q][2 * e], bv[q][2 * e + 1], bh[e]);
                    }
                    const u32x2 AH = {ah[0], ah[1]}, AL = {al[0], al[1]}, BH = {bh[0], bh[1]}, BL = {bl[0], bl[1]};
                    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, AL), __builtin_bit_cast(s16x4, BH), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, AH), __builtin_bit_cast(s16x4, BL), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, AH), __builtin_bit_cast(s16x4, BH), acc, 0, 0, 0);
                }
            }
        }
        return;
    }
    if (BF) {
        if (NQ % 2 == 0) {
#pragma unroll
            for (int q = 0; q < NQ; q += 2) {      // 16 k per MFMA: two of this lane's 4-deep groups
                if (FULL || 4 * (wk * GPW + 2 * q) < krem) {
                    const u32x4 a = {pack_bf16(av[q][0], av[q][1]), pack_bf16(av[q][2], av[q][3]),
                                     pack_bf16(av[q + 1][0], av[q + 1][1]), pack_bf16(av[q + 1][2], av[q + 1][3])};
                    const u32x4 b = {pack_bf16(bv[q][0], bv[q][1]), pack_bf16(bv[q][2], bv[q][3]),
                                     pack_bf16(bv[q + 1][0], bv[q + 1][1]), pack_bf16(bv[q + 1][2], bv[q + 1][3])};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                                  acc, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (FULL || 4 * (wk * GPW + 2 * q) < krem) {
                    const u32x2 a = {pack_bf16(av[q][0], av[q][1]), pack_bf16(av[q][2], av[q][3])};
                    const u32x2 b = {pack_bf16(bv[q][0], bv[q][1]), pack_bf16(bv[q][2], bv[q][3])};
                    acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), acc,
                                                                   0, 0, 0);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (FULL || 4 * (wk * GPW + 2 * q) < krem) {   // groups past the K tail hold zeros: skip them (wave-uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][j], bv[q][j], acc, 0, 0, 0);
        }
    }
}

// Write-through stores (sc0 sc1: the bytes leave the XCD's L2 at once) for tiles whose output another tile of the SAME launch
// reads (chained launches): after s_waitcnt vmcnt(0) the data is visible to every CU, no L2 write-back (release fence) needed.
__device__ __forceinline__ void st_pub(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_pub(void *p, u32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_pub(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_pub(unsigned short *p, unsigned v) { asm volatile("global_store_short %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

// four bf16 values (two packed dwords) of one output float4 to a twin plane; nrem valid columns; vec: 8-byte aligned
__device__ __forceinline__ void store_twin4(unsigned short *tp, unsigned w0, unsigned w1, int nrem, bool vec, bool pub) {
    if (nrem >= 4 && vec) {
        if (pub) st_pub(tp, u32x2{w0, w1});
        else *reinterpret_cast<u32x2 *>(tp) = u32x2{w0, w1};
    } else {
        const unsigned short h[4] = {(unsigned short)(w0 & 0xFFFF), (unsigned short)(w0 >> 16), (unsigned short)(w1 & 0xFFFF), (unsigned short)(w1 >> 16)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nrem) { if (pub) st_pub(tp + e, (unsigned)h[e]); else tp[e] = h[e]; }
    }
}

// Chained launch, consumer side: wave 0 polls the task's (counter, target) pairs, one pair per lane (relaxed agent-scope loads:
// served by the L2 / fabric, never by this CU's L1), then one agent-scope acquire (invalidates this CU's L1) and a workgroup
// barrier.  A producer is always a lower-indexed task of the launch, i.e. already dispatched (workgroups are dispatched in
// index order), so the wait ends; the spin is bounded anyway and a timeout is recorded in the launch's error word.
constexpr int CHAIN_SPIN_LIMIT = 1 << 20;
__device__ __forceinline__ void chain_wait(const Task &t, const Wait *__restrict__ waits, int *__restrict__ cnt, int tid, int knobs) {
    const int nwait = (knobs & 1024) ? 0 : t.wait_count;       // (knob 1024: timing experiment without the waits - results void)
    if (nwait > 0) {
        if (tid < 64) {
            for (int w0 = 0; w0 < nwait; w0 += 64) {
                const bool mine = w0 + tid < nwait;
                Wait w{0, 0};
                if (mine) w = waits[t.wait_begin + w0 + tid];
                bool ok = !mine;
                int spins = 0;
                for (;;) {
                    if (!ok) ok = ((knobs & 2048) ? __hip_atomic_fetch_add(cnt + 2 + w.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                  : __hip_atomic_load(cnt + 2 + w.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= w.target;
                    if (__all(ok)) break;
                    if (++spins > CHAIN_SPIN_LIMIT) {
                        if (tid == 0) __hip_atomic_store(cnt + 1, 1 + (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    // somebody else already gave up: the launch's results are void, do not spin out the full limit as well
                    if ((spins & 1023) == 0 && __hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    for (int z = 0; z < (knobs & 255); ++z) __builtin_amdgcn_s_sleep(8);      // 8 x 64 cycles per unit
                }
            }
            if (!(knobs & 256)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}
// ... producer side and launch bookkeeping: every thread's stores have been acknowledged (vmcnt(0)), then one thread bumps the
// task's counter.  The last workgroup of the launch to get here zeroes the block for the next launch (nobody polls any more).
__device__ __forceinline__ void chain_exit(int sig, int *__restrict__ cnt, int n_counters, int tid, int knobs) {
    if ((knobs & 512) && sig < 0) return;       // A/B knob: the host resets the counters (memset before the launch), no launch accounting
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (knobs & 512) { __hip_atomic_fetch_add(cnt + 2 + sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        if (sig >= 0) {     // ... and has been performed before this workgroup counts as done (the reset below must not overtake it)
            const int before = __hip_atomic_fetch_add(cnt + 2 + sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(before) : "memory");
        }
        const int done = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            // (the bump above is ordered before this read-modify-write in program order on one address only; the counters of other
            // addresses were bumped by workgroups whose `done` increments this one has observed through the same atomic unit)
            for (int i = 0; i < n_counters; ++i) __hip_atomic_store(cnt + 2 + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ const float *base_ptr(const Ptrs &p, int base) {
    switch (base) {
        case BASE_X: return p.x;
        case BASE_P: return p.p;
        case BASE_G: return p.g;
        case BASE_P16: return p.p16;
        default: return p.ws;
    }
}

}  // namespace

// -DTA3N_GEMM_STAMPS (a build directory of its own: TA3N_LIBDIR=ta3n_amd/lib_stamps; tools/gemm_stamps.py): every workgroup of a GEMM launch leaves
// 16 x 8-byte stamps in the workspace region "stamps", which the plan builder of such a build lays out directly in front of "zeros" (the
// kernel knows zeros_off): [0] entry, [1] descriptors / bias loaded, [2] K loop done, [3] all waves done, [4] accumulators in LDS, [5] stores
// issued, [6] the task's cost (its K), [7] its Seg count, [8] cycles thread 0 spent in the K loop waiting for "stage landed + barrier",
// [9] cycles it spent issuing the next stage's DMA + LDS reads + MFMAs, [10] stages.  (Round 6: the stamps used to live in a __device__
// array, of which every instantiation unit has its own copy - the reader only saw the launcher's.)
#ifdef TA3N_GEMM_STAMPS
#define TA3N_STAMP_SLOTS 16
#define TA3N_STAMP_FLOATS (8192 * TA3N_STAMP_SLOTS * 2)
#define GSTAMP_PTR (reinterpret_cast<unsigned long long *>(const_cast<float *>(ptrs.ws) + zeros_off - TA3N_STAMP_FLOATS))
#define GSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) GSTAMP_PTR[blockIdx.x * TA3N_STAMP_SLOTS + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define GSTAMP_VAL(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) GSTAMP_PTR[blockIdx.x * TA3N_STAMP_SLOTS + (i)] = (unsigned long long)(v); } while (0)
#define KSTAMP_DECL unsigned long long ks_wait = 0, ks_comp = 0, ks_t = 0; int ks_n = 0
#define KSTAMP_BEGIN_WAIT do { ks_t = __builtin_amdgcn_s_memtime(); } while (0)
#define KSTAMP_END_WAIT do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ks_wait += n_ - ks_t; ks_t = n_; ++ks_n; } while (0)
#define KSTAMP_END_COMP do { if (ks_t != 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ks_comp += n_ - ks_t; } } while (0)
#else
#define GSTAMP(i) do { } while (0)
#define GSTAMP_VAL(i, v) do { } while (0)
#define KSTAMP_DECL do { } while (0)
#define KSTAMP_BEGIN_WAIT do { } while (0)
#define KSTAMP_END_WAIT do { } while (0)
#define KSTAMP_END_COMP do { } while (0)
#endif

namespace ta3n {

// BF: bf16 MFMA on operands rounded in registers; NS: LDS stages in flight (2, or 3 with BF: once the MFMA is cheap the
// loop is latency-bound and a third stage pays for long K; short-K tasks prefer the extra resident workgroup of NS = 2).
// RM x RN: 32x32 output blocks per wave (bf16 twins only): the tile is (32 WM RM) x (32 WN RN).  What bounds these launches is
// the rate at which a CU can fill its LDS (~41 B/clk measured, tools/proto_bf16.hip), i.e. the operand bytes brought per flop -
// which only the tile size lowers.
// KV: which operand-kind combinations of the K loop the kernel contains (bit 0: both K-contiguous, 1: A K-contiguous x B k-major,
// 2: A k-major x B K-contiguous, 3: both k-major, 4: both k-major with the bias-gradient row sums; bit 5: the optional epilogue
// paths - fused update, split-K, write-through stores of chained launches); 63 = everything.  A launch whose tasks
// use few of them can run a kernel a fraction of the size (the plan knows: ta3n_plan::phase_kinds) - less code to fetch when the
// kernel changes between launches.
template <int WM, int WN, int WK, int BF, int NS, int RM, int RN, int KV = 63>
__device__ __forceinline__ void gemm_tile(const Task &t, const Seg *__restrict__ segs, const Ptrs &ptrs, int hyper_off, int zeros_off,
                                          int twin_off, const SgdSide &side, int pair_delta, int knobs) {
    constexpr int NW = WM * WN * WK, NT = 64 * NW;
    constexpr int BM = 32 * WM * RM, BN = 32 * WN * RN;
    constexpr bool HS = BF == 5;                     // bf16 twins in 64-k stages (OperandStream HS): half the bytes per stage
    constexpr int ROWF = HS ? BKC / 2 : BKC;         // floats per row of a K-contiguous stage image
    constexpr int STAGE = (BM + BN) * ROWF;          // floats per stage
    constexpr int CH = BF == 2 ? 2 * BKC : BKC;      // K elements per stage (bf16 twins: 128; pair twins: 64 as hi + lo; half stages: 64)
    constexpr bool TW = BF == 2 || BF == 4 || BF == 5;
    constexpr bool PAIR = BF == 4;
    constexpr int EPI = NW * RM * RN * 32 * 36;      // epilogue staging (one padded 32x32 block per wave and register block)
    constexpr int LDS_FLOATS = NS * STAGE > EPI + 4 ? NS * STAGE : EPI + 4;      // (+ 4: the split-K ticket behind the epilogue staging)
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];   // the ONLY LDS object of the kernel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t *)lds);
    const int li = lane & 31, lh = lane >> 5;
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);

    GSTAMP(0);
    constexpr bool OPT = TA3N_EXPERIMENTS != 0 && (KV & 32) != 0;      // fused update / split-K / chained-launch stores compiled in (experiments build only)
    const bool pub = OPT && t.sig >= 0 && !(side.p16_off == -12345);    // chained launch: another task of this launch reads what this one writes -> write-through stores
    if (t.epi & EPI_SGD) {          // optimiser side job (uniform for the workgroup): arithmetic and summation order of sgd_range_kernel
        if (side.params == nullptr) return;   // launched without an update to apply (ta3n_time_phases)
        float part = 0.f;
        if (tid < 256)
            part = strided_partial_sum(ptrs.ws + side.norm_off, side.norm_n, tid, 256);
        part = wave_allreduce_sum(part);
        if (lane == 0 && wave < 4) lds[wave] = part;
        __syncthreads();
        const float total = sqrtf(((lds[0] + lds[1]) + lds[2]) + lds[3]);
        float coef = 1.f;
        if (side.clip > 0.f) coef = fminf(side.clip / (total + 1e-6f), 1.f);
        float4 *__restrict__ p4 = reinterpret_cast<float4 *>(side.params);
        float4 *__restrict__ m4 = reinterpret_cast<float4 *>(side.momentum);
        const float4 *__restrict__ g4 = reinterpret_cast<const float4 *>(ptrs.g);
        for (int i = t.pad[0] + tid; i < t.pad[1]; i += NT) {
            const float4 p = p4[i], m = m4[i], gr = g4[i];
            float gg[4] = {gr.x, gr.y, gr.z, gr.w}, pp[4] = {p.x, p.y, p.z, p.w}, mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = fmaf(side.wd, pp[e], gg[e] * coef);
                mm[e] = fmaf(side.mu, mm[e], d);
                d = fmaf(side.mu, mm[e], d);
                pp[e] = fmaf(-side.lr, d, pp[e]);
            }
            if (pub) st_pub(reinterpret_cast<float *>(p4 + i), f32x4{pp[0], pp[1], pp[2], pp[3]});
            else p4[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
            m4[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            if (side.p16_off >= 0) {
                uint2 *tw = reinterpret_cast<uint2 *>(ptrs.ws + side.p16_off) + i;
                const unsigned h0 = pack_bf16(pp[0], pp[1]), h1 = pack_bf16(pp[2], pp[3]);
                if (pub) st_pub(tw, u32x2{h0, h1});
                else *tw = make_uint2(h0, h1);
                if ((BF == 3 || BF == 4) && pair_delta) {       // pair twins: the lo plane (uint2 = 2 floats)
                    const u32x2 l = {pack_bf16_lo(pp[0], pp[1], h0), pack_bf16_lo(pp[2], pp[3], h1)};
                    if (pub) st_pub(tw + pair_delta / 2, l);
                    else tw[pair_delta / 2] = make_uint2(l[0], l[1]);
                }
            }
        }
        return;
    }
    if (t.epi & EPI_COLSUM) {       // exact column sums of a [rows][ld] table of partials (uniform for the workgroup)
        float *__restrict__ dst = const_cast<float *>(base_ptr(ptrs, t.c_base)) + (size_t)t.c_off;
        const float *__restrict__ src = ptrs.ws + t.pad[0];
        float sq = 0.f;
        for (int n = t.n0 + tid; n < t.n_valid; n += NT) {
            float v = 0.f;
            // rows added in order, EIGHT loads in flight per round trip (the plain loop compiled to load - s_waitcnt vmcnt(0) - add per
            // row: 32 - 72 dependent L2 round trips, longer than every tile of the launch this task rides in; ISA pass of round 5)
            const float *__restrict__ col = src + n;
            const int R = t.pad[1];
            const size_t ld = (size_t)t.pad[2];
            int r = 0;
#pragma unroll 1
            for (; r + 8 <= R; r += 8) {
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = col[(size_t)(r + i) * ld];
#pragma unroll
                for (int i = 0; i < 8; ++i) v += x[i];
            }
            for (; r < R; ++r) v += col[(size_t)r * ld];
            dst[n] = v;
            sq = fmaf(v, v, sq);
            if (OPT && side.p_new != nullptr && t.c_base == BASE_G) {     // fused update of these parameters (see the tile epilogue)
                const size_t pi = (size_t)t.c_off + n;
                const float p0 = ptrs.p[pi];
                float d = fmaf(side.wd, p0, v), mm = fmaf(side.mu, side.momentum[pi], d);
                d = fmaf(side.mu, mm, d);
                const float p1 = fmaf(-side.lr, d, p0);
                side.p_new[pi] = p1; side.momentum[pi] = mm;
                if (side.p16_new != nullptr) reinterpret_cast<unsigned short *>(side.p16_new)[pi] = (unsigned short)(pack_bf16(p1, 0.f) & 0xFFFF);
            }
        }
        if (t.epi & EPI_SUMSQ) {
            sq = wave_allreduce_sum(sq);
            if (lane == 0) lds[wave] = sq;
            __syncthreads();
            if (tid == 0) {
                float tot = 0.f;
                for (int i = 0; i < NW; ++i) tot += lds[i];
                ptrs.ws[t.pad[3]] = tot;
            }
        }
        return;
    }
    if (t.seg_count == 0) return;   // padding task of the XCD-aware ordering (uniform for the workgroup)
    if (t.epi & EPI_SUMROWS8) {   // side job of one workgroup per fused step: add up the heads kernel's loss partials in a fixed order
        const float *__restrict__ src = ptrs.ws + t.pad[1];
        float sacc = 0.f;
        for (int r = tid >> 3; r < t.pad[2]; r += NT / 8) sacc += src[r * 8 + (tid & 7)];
        lds[tid] = sacc;
        __syncthreads();
        if (tid < 8) {
            float v = 0.f;
            for (int i = 0; i < NT / 8; ++i) v += lds[i * 8 + tid];
            ptrs.ws[t.pad[0] + tid] = v;
        }
        __syncthreads();
    }
    const Hyper *__restrict__ hy = reinterpret_cast<const Hyper *>(ptrs.ws + hyper_off);
    const float *__restrict__ zeros = ptrs.ws + zeros_off;   // 64 floats that are never written
    // What the epilogue needs that does not depend on the accumulators is REQUESTED now and consumed after the K loop:
    // the per-step scalars (one unconditional block load, independent of the Task - it travels beside the descriptor
    // instead of behind it; which of them the tile uses is decided in the epilogue) and this thread's four bias entries
    // (its column group is the same in every epilogue iteration).
    const uint32_t epi = t.epi;
    struct HyperHead { float beta[3], gamma, lr, momentum, weight_decay, clip, p_drop_i, p_drop_v; uint32_t seed_i, seed_v; };
    const HyperHead hh = *reinterpret_cast<const HyperHead *>(hy);      // first 12 dwords of Hyper (ta3n_types.h)
    const int h_train = hy->train;
    auto scale_of = [&](int kind) -> float {
        switch (kind) {
            case SK_NEG_BETA_REL: return -hh.beta[0];
            case SK_NEG_BETA_VID: return -hh.beta[1];
            case SK_NEG_BETA_FRM: return -hh.beta[2];
            case SK_INV_KEEP_I: return (h_train && hh.p_drop_i > 0.f) ? (hh.p_drop_i < 1.f ? 1.f / (1.f - hh.p_drop_i) : 0.f) : 1.f;
            case SK_INV_KEEP_V: return (h_train && hh.p_drop_v > 0.f) ? (hh.p_drop_v < 1.f ? 1.f / (1.f - hh.p_drop_v) : 0.f) : 1.f;
            case SK_REVERSE_MU: return hy->reverse ? -hy->mu : 1.f;      // (read here: only the one tile kind that needs it pays the load)
            default: return 1.f;
        }
    };
    float ebias[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi & EPI_BIAS) {
        const float *__restrict__ bias = base_ptr(ptrs, t.bias_base) + t.bias_off;
        const int n = t.n0 + (tid % (BN / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < t.n_valid) ebias[e] = bias[n + e];
    }

    // Fused update (SgdSide::p_new, gradient tiles only): the old parameter and momentum entries this thread will update are
    // requested NOW, like the bias, and consumed in the epilogue - in the epilogue itself the two dependent loads were ~1.5 us on
    // the critical path of every weight-gradient tile.  Tiles with more than two epilogue passes per thread (register-blocked
    // 128-wide tiles) load them in the epilogue instead: 8 registers per pass are too many to carry through the K loop.
    constexpr int ITER_E = (BM * BN / 4 + NT - 1) / NT;
    constexpr bool UPD_PREFETCH = ITER_E <= 2;
    const bool upd_early = OPT && UPD_PREFETCH && side.p_new != nullptr && t.c_base == BASE_G;
    float up_e[UPD_PREFETCH ? ITER_E : 1][4], um_e[UPD_PREFETCH ? ITER_E : 1][4];
    if (upd_early) {
        const bool cv = ((t.c_off | t.c_ld) & 3) == 0;
#pragma unroll
        for (int it = 0; it < (UPD_PREFETCH ? ITER_E : 1); ++it) {
            const int idx = tid + it * NT;
            const int mm_ = t.m0 + idx / (BN / 4), nn_ = t.n0 + (idx % (BN / 4)) * 4;
            const int nr_ = (idx < BM * BN / 4 && mm_ < t.m_valid) ? t.n_valid - nn_ : 0;
            const size_t pi = (size_t)t.c_off + (size_t)mm_ * t.c_ld + nn_;
#pragma unroll
            for (int e = 0; e < 4; ++e) { up_e[it][e] = 0.f; um_e[it][e] = 0.f; }
            if (nr_ >= 4 && cv) {
                const float4 q4 = *reinterpret_cast<const float4 *>(ptrs.p + pi), r4 = *reinterpret_cast<const float4 *>(side.momentum + pi);
                up_e[it][0] = q4.x; up_e[it][1] = q4.y; up_e[it][2] = q4.z; up_e[it][3] = q4.w;
                um_e[it][0] = r4.x; um_e[it][1] = r4.y; um_e[it][2] = r4.z; um_e[it][3] = r4.w;
            } else if (nr_ > 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nr_) { up_e[it][e] = ptrs.p[pi + e]; um_e[it][e] = side.momentum[pi + e]; }
            }
        }
    }

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    GSTAMP(1);

    const int m0 = t.m0, n0 = t.n0, m_valid = t.m_valid, n_valid = t.n_valid;
    const int seg_end = t.seg_begin + t.seg_count;
    int cseg = t.seg_begin;
    // K loop: NS stages, chunk c + NS - 1 is streamed while chunk c is computed, across Seg boundaries.  All Segs of a
    // task have the same operand kinds (the plan builder guarantees it), so the whole loop nest is instantiated per
    // kind combination and selected once per task: the register allocation is the maximum over the combinations,
    // not their union (loop-invariant LDS addresses of every combination used to be live together).
    const int ra = wm * (32 * RM) + li, rb = wn * (32 * RN) + li;   // first of this lane's RM (RN) rows, 32 apart
    float rs[RM];     // EPI_ROWSUM_A: K-sum of A(row ra + 32 i, this half-wave's k) over this wave's K slices
#pragma unroll
    for (int i = 0; i < RM; ++i) rs[i] = 0.f;
    // (Two bit-identical variants of this loop - the next chunk's DMA pieces issued BETWEEN the matrix instructions, and a refill of the
    // buffer just read into registers - were built and measured slower in round 4 (profiles/r04_dma_interleave_ab.txt,
    // r04_early_refill_ab.txt) and removed from the tree in round 5: docs/history/round4.md, git 545b5f2.)
    (void)knobs;
    KSTAMP_DECL;
    constexpr int KVE = BM > 128 ? (KV & (1 | 2 | 32)) : KV;      // tiles taller than 128 rows: K-contiguous A only (the plan keeps them off other launches)
    auto k_loop = [&](auto akm, auto bkm, auto rsum) {
        constexpr bool AKM = decltype(akm)::value, BKM = decltype(bkm)::value, RS = decltype(rsum)::value;
        if constexpr (NS == 2) {
            // Two stages: chunk c + 1 is streamed while chunk c is computed.  Per Seg, the iterations whose successor is in
            // the same Seg run in a tight loop; the iteration that computes the Seg's last chunk opens the next Seg.
            OperandStream<BM, NW, TW, PAIR, HS> oa;
            OperandStream<BN, NW, TW, PAIR, HS> ob;
            int klen = 0, scale = SK_ONE;
            // The descriptor of the NEXT Seg is fetched while the current one streams (scalar loads, consumed at the next
            // open): opening a Seg used to start with a dependent global load between "stage landed" and "next DMA issued" -
            // pure load-path idle time, once per Seg, and the gradient-at-F1 tiles have a Seg every two chunks.
            Seg nx = t.seg0;
            auto open_seg = [&](int sidx) {            // wave-uniform: Seg fields live in SGPRs
                const Seg sg = nx;
                if (sidx + 1 < seg_end) nx = segs[sidx + 1];
                klen = sg.klen; scale = sg.scale_kind;
                oa.setup(base_ptr(ptrs, sg.a_base) + (size_t)sg.a_off, sg.a_ld, AKM, sg.klen, m0, sg.pad[0] > 0 ? sg.pad[0] : m_valid,
                         wave, lane, pair_delta);
                ob.setup(base_ptr(ptrs, sg.b_base) + (size_t)sg.b_off, sg.b_ld, BKM, sg.klen, n0, n_valid, wave, lane, pair_delta);
            };
            auto issue = [&](int buf, int k0) {
                const unsigned st = lds_base + (unsigned)(buf * STAGE * 4);
                oa.issue(k0, klen - k0, st, wave, lane, zeros);
                ob.issue(k0, klen - k0, st + BM * ROWF * 4, wave, lane, zeros);
            };
            auto stage_ready = [&]() {
                KSTAMP_END_COMP;
                KSTAMP_BEGIN_WAIT;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the current stage have landed
                __builtin_amdgcn_s_barrier();                        // ... everyone's; and everyone is done reading the other stage
                asm volatile("" ::: "memory");
                KSTAMP_END_WAIT;
            };
            int buf = 0;
            open_seg(cseg);
            issue(0, 0);
            for (;;) {
                const int n_chunks = (klen + CH - 1) / CH;
                for (int c = 0; c < n_chunks - 1; ++c) {             // chunks whose successor is in the same Seg (all full)
                    stage_ready();
                    const float *sa = lds + buf * STAGE;
                    issue(buf ^ 1, (c + 1) * CH);
                    compute_stage<BM, BN, WK, AKM, BKM, true, RS, BF, RM, RN>(acc, rs, sa, sa + BM * ROWF, ra, rb, wk, lh, CH);
                    buf ^= 1;
                }
                // last chunk of this Seg; the next Seg (if any) starts streaming underneath it
                stage_ready();
                const int krem = klen - (n_chunks - 1) * CH, c_scale = scale;
                ++cseg;
                const bool more = cseg < seg_end;
                if (more) {
                    open_seg(cseg);
                    issue(buf ^ 1, 0);
                }
                const float *sa = lds + buf * STAGE;
                compute_stage<BM, BN, WK, AKM, BKM, false, RS, BF, RM, RN>(acc, rs, sa, sa + BM * ROWF, ra, rb, wk, lh, krem);
                if (c_scale != SK_ONE) {
                    const float sc = scale_of(c_scale);
    #pragma unroll
                    for (int i = 0; i < RM; ++i)
    #pragma unroll
                        for (int j = 0; j < RN; ++j)
    #pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
                }
                if (cseg >= seg_end) break;
                buf ^= 1;
            }
            } else {
            // Two cursors walk the task's Segs chunk by chunk: the issue cursor (per-lane DMA state) runs up to NS - 1
            // chunks ahead of the compute cursor (scalar state only).
            OperandStream<BM, NW, TW, PAIR, HS> oa;
            OperandStream<BN, NW, TW, PAIR, HS> ob;
            constexpr int LPW = OperandStream<BM, NW, TW, PAIR, HS>::NP + OperandStream<BN, NW, TW, PAIR, HS>::NP;   // DMAs per lane and chunk (16-byte path;
                                                                                         // the 4-byte path issues more, never fewer)
            int i_seg = cseg, i_chunk = 0, i_nchunks = 0, i_klen = 0, i_buf = 0, ahead = 0;
            Seg nx = t.seg0;                         // descriptor of the Seg the issue cursor opens next, fetched one Seg ahead
            auto open_issue_seg = [&]() {            // wave-uniform: Seg fields live in SGPRs
                const Seg sg = nx;
                if (i_seg + 1 < seg_end) nx = segs[i_seg + 1];
                i_klen = sg.klen; i_nchunks = (sg.klen + CH - 1) / CH; i_chunk = 0;
                oa.setup(base_ptr(ptrs, sg.a_base) + (size_t)sg.a_off, sg.a_ld, AKM, sg.klen, m0, sg.pad[0] > 0 ? sg.pad[0] : m_valid,
                         wave, lane, pair_delta);
                ob.setup(base_ptr(ptrs, sg.b_base) + (size_t)sg.b_off, sg.b_ld, BKM, sg.klen, n0, n_valid, wave, lane, pair_delta);
            };
            auto issue_one = [&]() {                 // stream the chunk under the issue cursor, advance the cursor
                const unsigned st = lds_base + (unsigned)(i_buf * STAGE * 4);
                const int k0 = i_chunk * CH;
                oa.issue(k0, i_klen - k0, st, wave, lane, zeros);
                ob.issue(k0, i_klen - k0, st + BM * ROWF * 4, wave, lane, zeros);
                i_buf = (i_buf + 1 == NS) ? 0 : i_buf + 1;
                ++ahead;
                if (++i_chunk == i_nchunks) {
                    if (++i_seg < seg_end) open_issue_seg();
                }
            };
            auto wait_landed = [&](int younger) {     // the oldest chunk in flight has landed once only the younger ones' DMAs are out
                static_assert((NS - 2) * LPW <= 63, "vmcnt is a 6-bit counter");
                KSTAMP_END_COMP;
                KSTAMP_BEGIN_WAIT;
                if (NS == 2 || younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (NS == 3 || younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
                else if (NS == 4 || younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPW) : "memory");
                else if (NS == 5 || younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPW) : "memory");
                __builtin_amdgcn_s_barrier();          // ... everyone's pieces have; and everyone is done reading the stage refilled next
                asm volatile("" ::: "memory");
                KSTAMP_END_WAIT;
            };
            open_issue_seg();
    #pragma unroll 1
            for (int sidx = 0; sidx < NS - 1 && i_seg < seg_end; ++sidx) issue_one();

            int c_buf = 0;
            int c_klen_nx = t.seg0.klen, c_scale_nx = t.seg0.scale_kind;   // compute cursor: the same one-ahead fetch of (klen, scale)
            for (;;) {
                const int klen = c_klen_nx, c_scale = c_scale_nx;
                if (cseg + 1 < seg_end) { c_klen_nx = segs[cseg + 1].klen; c_scale_nx = segs[cseg + 1].scale_kind; }
                const int n_chunks = (klen + CH - 1) / CH;
                int c = 0;
                // interior of the Seg: both cursors inside it, NS - 1 chunks in flight, nothing to decide per chunk
                if (i_seg == cseg) {
                    for (; c + (NS - 1) < n_chunks; ++c) {
                        wait_landed(NS - 2);
                        const unsigned st = lds_base + (unsigned)(i_buf * STAGE * 4);
                        const int k0 = (c + NS - 1) * CH;
                        const float *sa = lds + c_buf * STAGE;
                        oa.issue(k0, klen - k0, st, wave, lane, zeros);
                        ob.issue(k0, klen - k0, st + BM * ROWF * 4, wave, lane, zeros);
                        compute_stage<BM, BN, WK, AKM, BKM, true, RS, BF, RM, RN>(acc, rs, sa, sa + BM * ROWF, ra, rb, wk, lh, CH);
                        i_buf = (i_buf + 1 == NS) ? 0 : i_buf + 1;
                        c_buf = (c_buf + 1 == NS) ? 0 : c_buf + 1;
                    }
                    // (issue cursor inside this Seg => NS - 1 of its chunks were in flight => the loop ran and sent its last chunk)
                    if (++i_seg < seg_end) open_issue_seg();
                }
                // last NS - 1 chunks of the Seg: the issue cursor is in a later Seg (or done)
    #pragma unroll 1
                for (; c < n_chunks; ++c) {
                    wait_landed(ahead - 1);
                    if (i_seg < seg_end) issue_one();
                    const float *sa = lds + c_buf * STAGE;
                    if (c < n_chunks - 1)
                        compute_stage<BM, BN, WK, AKM, BKM, true, RS, BF, RM, RN>(acc, rs, sa, sa + BM * ROWF, ra, rb, wk, lh, CH);
                    else
                        compute_stage<BM, BN, WK, AKM, BKM, false, RS, BF, RM, RN>(acc, rs, sa, sa + BM * ROWF, ra, rb, wk, lh, klen - c * CH);
                    c_buf = (c_buf + 1 == NS) ? 0 : c_buf + 1;
                    --ahead;
                }
                if (c_scale != SK_ONE) {
                    const float sc = scale_of(c_scale);
    #pragma unroll
                    for (int i = 0; i < RM; ++i)
    #pragma unroll
                        for (int j = 0; j < RN; ++j)
    #pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
                }
                if (++cseg >= seg_end) break;
            }
            }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    {
        const Seg &s0 = t.seg0;
        switch (s0.a_kmajor * 2 + s0.b_kmajor) {
            case 0: if constexpr (KVE & 1) k_loop(F_{}, F_{}, F_{}); break;
            case 1: if constexpr (KVE & 2) k_loop(F_{}, T_{}, F_{}); break;
            case 2: if constexpr (KVE & 4) k_loop(T_{}, F_{}, F_{}); break;
            default:   // weight gradients (both operands k-major) are the only tiles that also produce a bias gradient
                if (t.epi & EPI_ROWSUM_A) { if constexpr (KVE & 16) k_loop(T_{}, T_{}, T_{}); }
                else { if constexpr (KVE & 8) k_loop(T_{}, T_{}, F_{}); }
                break;
        }
    }

    // ---- epilogue: accumulators -> LDS (reduces the K split, makes rows contiguous) ----
    KSTAMP_END_COMP;
    GSTAMP(2);
#ifdef TA3N_GEMM_STAMPS
    GSTAMP_VAL(8, ks_wait); GSTAMP_VAL(9, ks_comp); GSTAMP_VAL(10, ks_n);
#endif
    __syncthreads();
    GSTAMP(3);
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            float *cs = lds + ((wave * RM + i) * RN + j) * (32 * 36);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;   // 32x32 C/D fragment layout
                cs[row * 36 + li] = acc[i][j][r];
            }
        }
    __syncthreads();
    GSTAMP(4);

    // Split-K (EPI_SPLITK): this workgroup holds the tile's sum over ITS share of the K segments.  It publishes that partial tile
    // (write-through stores: visible to every CU once acknowledged), takes a ticket, and if it is the first of the pair it is done;
    // the second adds the other's partial (coherent loads) to its own in the loop below.  a + b = b + a: who finishes is immaterial.
    const float *__restrict__ split_other = nullptr;
    if (OPT && (epi & EPI_SPLITK)) {
        int &split_ticket = *reinterpret_cast<int *>(&lds[EPI]);
        const int half = t.pad[2] - 1;
        float *mine = ptrs.ws + t.pad[0] + (size_t)half * (BM * BN);
        constexpr int ITER_S = (BM * BN / 4 + NT - 1) / NT;
#pragma unroll
        for (int it = 0; it < ITER_S; ++it) {
            const int idx = tid + it * NT;
            if (idx < BM * BN / 4) {
                const int r = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
                const int br = r >> 5, bc = c4 >> 5;
                const int slot0 = ((((br / RM) * WN + (bc / RN)) * WK) * RM + (br % RM)) * RN + (bc % RN);
                float4 v4 = zero4();
#pragma unroll
                for (int q = 0; q < WK; ++q) {
                    const float4 part = *reinterpret_cast<const float4 *>(&lds[(slot0 + q * RM * RN) * (32 * 36) + (r & 31) * 36 + (c4 & 31)]);
                    v4.x += part.x; v4.y += part.y; v4.z += part.z; v4.w += part.w;
                }
                st_pub(mine + (size_t)r * BN + c4, f32x4{v4.x, v4.y, v4.z, v4.w});
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *ticket = reinterpret_cast<int *>(ptrs.ws + t.pad[1]);
        if (tid == 0) split_ticket = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (split_ticket == 0) return;                          // (uniform) the partner finishes the tile
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next step
        split_other = ptrs.ws + t.pad[0] + (size_t)(1 - half) * (BM * BN);
    }
    // ... its partial tile is read in the store loop below with agent-scope loads (coherent: never served from a stale line of this
    // XCD's L2; compiler-managed, so nothing reads the registers before the data has landed)
    const float alpha = scale_of(t.alpha_kind);
    const float gamma = scale_of(t.gamma_kind);
    const bool drop_on = (epi & (EPI_DROP_I | EPI_DROP_V)) && h_train != 0;
    const uint32_t dseed = (epi & EPI_DROP_I) ? hh.seed_i : hh.seed_v;
    const float dp = (epi & EPI_DROP_I) ? hh.p_drop_i : hh.p_drop_v;
    float *__restrict__ cbase = const_cast<float *>(base_ptr(ptrs, t.c_base)) + (size_t)t.c_off;
    const float *__restrict__ aux = (epi & EPI_MASK) ? base_ptr(ptrs, t.aux_base) + t.aux_off : nullptr;
    const float *__restrict__ add = (epi & EPI_ADD) ? base_ptr(ptrs, t.add_base) + t.add_off : nullptr;
    const bool c_vec = ((t.c_off | t.c_ld) & 3) == 0;
    const bool aux_vec = ((t.aux_off | t.aux_ld) & 3) == 0, add_vec = ((t.add_off | t.add_ld) & 3) == 0;
    const int nfan = t.fan_count;
    // Fused update (SgdSide::p_new): a gradient tile is also the optimiser step of its block of parameters.  torch.optim.SGD with
    // nesterov and weight decay (main.py:83, 583) in the arithmetic of sgd_kernel with the clip coefficient taken as 1:
    //   d = wd p + g;  m = mu m + d;  p_new = p - lr (d + mu m)
    // (a step whose gradient norm exceeds clip_gradient is corrected afterwards by sgd_fixup_kernel: the update is linear in g).
    const bool upd = OPT && side.p_new != nullptr && t.c_base == BASE_G;      // workgroup-uniform
    float sumsq = 0.f;   // EPI_SUMSQ: this thread's share of the tile's sum of squares
    // Every flag test below is workgroup-uniform: a tile without bias / mask / residual issues no load for it (the
    // bias and the per-step scalars were fetched before the K loop, so the first dependent global access of a plain
    // tile is its store).
    constexpr int ITER = (BM * BN / 4 + NT - 1) / NT;
    constexpr int UNR = ITER <= 4 ? ITER : 2;
#pragma unroll UNR
    for (int it = 0; it < ITER; ++it) {
        const int idx = tid + it * NT;
        if (idx >= BM * BN / 4) break;
        const int r = idx / (BN / 4);
        const int c4 = (idx % (BN / 4)) * 4;        // == ec4 for every iteration: NT is a multiple of BN / 4
        // block (r >> 5, c4 >> 5) belongs to wave row (r >> 5) / RM, wave column (c4 >> 5) / RN; staging slot of K-split wave q:
        // (((wm WN + wn) WK + q) RM + i) RN + j
        const int br = r >> 5, bc = c4 >> 5;
        const int slot0 = ((((br / RM) * WN + (bc / RN)) * WK) * RM + (br % RM)) * RN + (bc % RN);
        const int m = m0 + r, n = n0 + c4;
        const bool row_ok = m < m_valid;
        const int nrem = row_ok ? n_valid - n : 0;   // number of valid columns of this float4 (<= 0: none)
        float av[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {1.f, 1.f, 1.f, 1.f};
        if (add != nullptr && nrem > 0) {           // issued first: their latency overlaps the LDS reads below
            const float *ap = add + (size_t)m * t.add_ld + n;
            if (nrem >= 4 && add_vec) { const float4 q4 = *reinterpret_cast<const float4 *>(ap); av[0] = q4.x; av[1] = q4.y; av[2] = q4.z; av[3] = q4.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nrem) av[e] = ap[e];
            }
        }
        if (aux != nullptr && nrem > 0) {
            const float *xp = aux + (size_t)m * t.aux_ld + n;
            if (nrem >= 4 && aux_vec) { const float4 q4 = *reinterpret_cast<const float4 *>(xp); mv[0] = q4.x; mv[1] = q4.y; mv[2] = q4.z; mv[3] = q4.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nrem) mv[e] = xp[e];
            }
        }
        float up[4] = {0.f, 0.f, 0.f, 0.f}, um[4] = {0.f, 0.f, 0.f, 0.f};      // fused update: old parameters and momentum of these entries
        if (upd_early) {                          // requested before the K loop
#pragma unroll
            for (int e = 0; e < 4; ++e) { up[e] = up_e[UPD_PREFETCH ? it : 0][e]; um[e] = um_e[UPD_PREFETCH ? it : 0][e]; }
        } else if (upd && nrem > 0) {             // requested here: the latency overlaps the LDS reads below
            const size_t pi = (size_t)t.c_off + (size_t)m * t.c_ld + n;
            if (nrem >= 4 && c_vec) {
                const float4 q4 = *reinterpret_cast<const float4 *>(ptrs.p + pi), r4 = *reinterpret_cast<const float4 *>(side.momentum + pi);
                up[0] = q4.x; up[1] = q4.y; up[2] = q4.z; up[3] = q4.w; um[0] = r4.x; um[1] = r4.y; um[2] = r4.z; um[3] = r4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nrem) { up[e] = ptrs.p[pi + e]; um[e] = side.momentum[pi + e]; }
            }
        }
        float other4[4] = {0.f, 0.f, 0.f, 0.f};
        if (split_other != nullptr) {
            const float *op = split_other + (size_t)r * BN + c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) other4[e] = __hip_atomic_load(op + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float4 v4 = zero4();
#pragma unroll
        for (int q = 0; q < WK; ++q) {
            const float4 part = *reinterpret_cast<const float4 *>(&lds[(slot0 + q * RM * RN) * (32 * 36) + (r & 31) * 36 + (c4 & 31)]);
            v4.x += part.x; v4.y += part.y; v4.z += part.z; v4.w += part.w;
        }
        if (split_other != nullptr) { v4.x += other4[0]; v4.y += other4[1]; v4.z += other4[2]; v4.w += other4[3]; }
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = (v[e] + ebias[e]) * alpha + av[e];
            if (epi & EPI_RELU) x = fmaxf(x, 0.f);
            x = mv[e] > 0.f ? x : 0.f;
            if (drop_on) x *= keep_mask(dseed, (uint32_t)(m * t.drop_ld + n + e), dp);
            v[e] = x * gamma;
        }
        if (nrem <= 0) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nrem) sumsq = fmaf(v[e], v[e], sumsq);
        float *cp = cbase + (size_t)m * t.c_ld + n;
        if (epi & EPI_TWIN_ONLY) {
            // every consumer of this tile reads its bf16 twin: the fp32 copy is not written (a third of the store burst)
        } else if (nrem >= 4 && c_vec) {
            if (pub) st_pub(cp, f32x4{v[0], v[1], v[2], v[3]});
            else *reinterpret_cast<float4 *>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nrem) { if (pub) st_pub(cp + e, v[e]); else cp[e] = v[e]; }
        }
        if (upd) {
            const size_t pi = (size_t)t.c_off + (size_t)m * t.c_ld + n;
            float pn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = fmaf(side.wd, up[e], v[e]);
                um[e] = fmaf(side.mu, um[e], d);
                d = fmaf(side.mu, um[e], d);
                pn[e] = fmaf(-side.lr, d, up[e]);
            }
            if (nrem >= 4 && c_vec) {
                *reinterpret_cast<float4 *>(side.p_new + pi) = make_float4(pn[0], pn[1], pn[2], pn[3]);
                *reinterpret_cast<float4 *>(side.momentum + pi) = make_float4(um[0], um[1], um[2], um[3]);
                if (side.p16_new != nullptr)
                    *reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned short *>(side.p16_new) + pi) = u32x2{pack_bf16(pn[0], pn[1]), pack_bf16(pn[2], pn[3])};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < nrem) {
                        side.p_new[pi + e] = pn[e]; side.momentum[pi + e] = um[e];
                        if (side.p16_new != nullptr) reinterpret_cast<unsigned short *>(side.p16_new)[pi + e] = (unsigned short)(pack_bf16(pn[e], 0.f) & 0xFFFF);
                    }
            }
        }
        if (epi & EPI_TWIN16) {          // bf16 twin of the stored values (TA3N_FLAG_BF16_STORE)
            unsigned short *tp = reinterpret_cast<unsigned short *>(ptrs.ws + twin_off) + ((size_t)t.c_off + (size_t)m * t.c_ld + n);
            const unsigned lo = pack_bf16(v[0], v[1]), hi = pack_bf16(v[2], v[3]);
            store_twin4(tp, lo, hi, nrem, c_vec, pub);
            if ((BF == 3 || BF == 4) && pair_delta)      // pair twins (split-arithmetic launches only): the lo plane of the same four values
                store_twin4(tp + 2 * (size_t)pair_delta, pack_bf16_lo(v[0], v[1], lo), pack_bf16_lo(v[2], v[3], hi), nrem, c_vec, pub);
        }
        if (nfan > 0) {
            const bool fan_vec = nrem >= 4 && (t.fan_ld & 3) == 0;
            float fm[3][4];
#pragma unroll
            for (int f = 0; f < 3; ++f) {   // all mask loads first
                if (f < nfan) {
                    const float *mp = ptrs.ws + (size_t)t.fan_mask_off[f] + (size_t)m * t.fan_ld + n;
                    if (fan_vec && (t.fan_mask_off[f] & 3) == 0) { const float4 q4 = *reinterpret_cast<const float4 *>(mp); fm[f][0] = q4.x; fm[f][1] = q4.y; fm[f][2] = q4.z; fm[f][3] = q4.w; }
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) fm[f][e] = e < nrem ? mp[e] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < 3; ++f) {   // same value through several ReLU masks (TRN tuples of one scale)
                if (f < nfan) {
                    float *op = ptrs.ws + (size_t)t.fan_out_off[f] + (size_t)m * t.fan_ld + n;
                    float ov[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (e < nrem && fm[f][e] > 0.f) ? v[e] : 0.f;
                    if (epi & EPI_TWIN_ONLY_FAN) {
                    } else if (fan_vec && (t.fan_out_off[f] & 3) == 0) {
                        if (pub) st_pub(op, f32x4{ov[0], ov[1], ov[2], ov[3]});
                        else *reinterpret_cast<float4 *>(op) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (e < nrem) { if (pub) st_pub(op + e, ov[e]); else op[e] = ov[e]; }
                    }
                    if (epi & EPI_TWIN16_FAN) {
                        unsigned short *tp = reinterpret_cast<unsigned short *>(ptrs.ws + twin_off) +
                                             ((size_t)t.fan_out_off[f] + (size_t)m * t.fan_ld + n);
                        const unsigned lo = pack_bf16(ov[0], ov[1]), hi = pack_bf16(ov[2], ov[3]);
                        const bool fvec = ((t.fan_out_off[f] | t.fan_ld) & 3) == 0;
                        store_twin4(tp, lo, hi, nrem, fvec, pub);
                        if ((BF == 3 || BF == 4) && pair_delta)
                            store_twin4(tp + 2 * (size_t)pair_delta, pack_bf16_lo(ov[0], ov[1], lo), pack_bf16_lo(ov[2], ov[3], hi), nrem, fvec, pub);
                    }
                }
            }
        }
    }
    if (epi & EPI_ROWSUM_A) {   // wave-uniform: bias gradient = K-sums of the A rows, added over the two k halves and the K-split waves
        __syncthreads();
        if (wn == 0) {
#pragma unroll
            for (int i = 0; i < RM; ++i) lds[(wk * 2 + lh) * BM + ra + 32 * i] = rs[i];
        }
        __syncthreads();
        if (tid < BM) {
            float b = 0.f;
#pragma unroll
            for (int i = 0; i < 2 * WK; ++i) b += lds[i * BM + tid];
            if (m0 + tid < m_valid) {
                const_cast<float *>(base_ptr(ptrs, t.bias_base))[(size_t)t.bias_off + m0 + tid] = b;
                sumsq = fmaf(b, b, sumsq);
                if (upd && t.bias_base == BASE_G) {      // the bias gradient's own parameter entries
                    const size_t pi = (size_t)t.bias_off + m0 + tid;
                    const float p0 = ptrs.p[pi];
                    float d = fmaf(side.wd, p0, b), mm = fmaf(side.mu, side.momentum[pi], d);
                    d = fmaf(side.mu, mm, d);
                    const float p1 = fmaf(-side.lr, d, p0);
                    side.p_new[pi] = p1; side.momentum[pi] = mm;
                    if (side.p16_new != nullptr) reinterpret_cast<unsigned short *>(side.p16_new)[pi] = (unsigned short)(pack_bf16(p1, 0.f) & 0xFFFF);
                }
            }
        }
    }
    GSTAMP(5);
#if defined(TA3N_GEMM_STAMPS) && TA3N_GEMM_STAMPS == 1
    GSTAMP_VAL(6, t.cost); GSTAMP_VAL(7, t.seg_count);
#endif
    if (epi & EPI_SUMSQ) {   // wave-uniform: fixed-order block sum -> this tile's slot (fused grad-norm partial)
        __syncthreads();
        sumsq = wave_allreduce_sum(sumsq);
        if (lane == 0) lds[wave] = sumsq;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int i = 0; i < NW; ++i) tot += lds[i];
            ptrs.ws[t.pad[3]] = tot;
        }
    }
}

// One workgroup = one Task of the launch's list.  chain_off >= 0: a chained launch (several dependency levels; ta3n_types.h).
template <int WM, int WN, int WK, int BF, int NS, int RM, int RN, int KV = 63>
__global__ __launch_bounds__(64 * WM * WN * WK) void gemm_tiles(const Task *__restrict__ tasks, const Seg *__restrict__ segs,
                                                                 Ptrs ptrs, int hyper_off, int zeros_off, int twin_off, SgdSide side,
                                                                 const Wait *__restrict__ waits, int chain_off, int chain_n, int knobs,
                                                                 int pair_delta) {
    int *cnt = reinterpret_cast<int *>(ptrs.ws + (chain_off >= 0 ? chain_off : 0));
    if (TA3N_EXPERIMENTS != 0 && chain_off >= 0) {
        const Task &t = tasks[blockIdx.x];
#if defined(TA3N_GEMM_STAMPS) && TA3N_GEMM_STAMPS == 2      // tools/chain_stamps.py: [6] = workgroup entry (before the wait), [7] = after the exit bookkeeping
        GSTAMP(6);
#endif
        chain_wait(t, waits, cnt, (int)threadIdx.x, knobs);
        gemm_tile<WM, WN, WK, BF, NS, RM, RN, KV>(t, segs, ptrs, hyper_off, zeros_off, twin_off, side, pair_delta, knobs);
        chain_exit(t.sig, cnt, chain_n, (int)threadIdx.x, knobs);
#if defined(TA3N_GEMM_STAMPS) && TA3N_GEMM_STAMPS == 2
        GSTAMP(7);
#endif
        return;
    }
    // One workgroup = one task.  (Round 5 measured the alternative the round-3 / round-4 verdicts asked for - resident workgroups walking
    // the per-XCD task list with stride gridDim.x, TA3N_PERSIST - without overlap between a tile's epilogue and its successor's first
    // stages: slower on every configuration (profiles/r05_persistent_ab.txt: fp32 259 -> 265-274 us, configs[3] 506 -> 523-592 us,
    // headline 113.2 -> 120.9 us), because static assignment loses the hardware dispatcher's balancing of tiles whose K differs 6x, and
    // the task loop itself cost the fp32 32x64 kernel its second resident workgroup (129 VGPRs).  Removed again)
    gemm_tile<WM, WN, WK, BF, NS, RM, RN, KV>(tasks[blockIdx.x], segs, ptrs, hyper_off, zeros_off, twin_off, side, pair_delta, knobs);
}

#define TA3N_TILE_CONFIGS(X) X(1, 1, 4) X(1, 1, 8) X(2, 1, 2) X(1, 2, 2) X(2, 1, 4) X(1, 2, 4) X(2, 2, 1) X(2, 2, 2)

// register-blocked tiles (bf16 twins only): (wm, wn, wk, rm, rn, stages...) - 128x128 with 8 or 4 waves, 64x128, 128x64
#define TA3N_BLOCKED_CONFIGS(X) X(2, 2, 2, 2, 2, 2) X(2, 2, 1, 2, 2, 2) X(2, 2, 2, 1, 2, 2) X(2, 2, 2, 1, 2, 3) X(2, 2, 2, 2, 1, 2) X(2, 2, 2, 2, 1, 3)

#define TA3N_INSTANTIATE(wm, wn, wk)                                                                        \
    template __global__ void gemm_tiles<wm, wn, wk, 0, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 1, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 1, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 2, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 2, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 3, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 3, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 4, 2, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int); \
    template __global__ void gemm_tiles<wm, wn, wk, 4, 3, 1, 1>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);
// half-stage kernels (MODE 5: bf16 twins in 64-k stages, OperandStream HS): (wm, wn, wk, rm, rn, stages) - the 128x128 and 64x64 tiles
// and - four waves, no K split inside the workgroup, 3 x 2 / 4 x 2 blocks per wave - the 192x128 and the 256x128 tile (A K-contiguous only)
#if TA3N_EXPERIMENTS
#define TA3N_HS_CONFIGS(X) X(2, 2, 2, 2, 2, 3) X(2, 2, 2, 2, 2, 4) X(2, 2, 2, 1, 1, 3) X(2, 2, 2, 1, 1, 4) X(2, 2, 1, 3, 2, 3) X(2, 2, 1, 4, 2, 3) X(2, 2, 1, 2, 2, 2)
#else       // default library: three half stages of the 128x128 and the 64x64 tile (four stages and the four-wave tall tiles measured slower)
// (2, 2, 1, 2, 2, 2), tile code 35221 (round 5): the 128x128 tile with FOUR waves (64x64 per wave, no K split inside the workgroup) on two
// half stages - 64 KB of stages, 72 KB of epilogue staging, <= 256 registers: TWO such workgroups per compute unit, where the eight-wave
// 128x128 tile (219 VGPRs) runs alone - one tile's epilogue and descriptor fetch under the other's K loop
#define TA3N_HS_CONFIGS(X) X(2, 2, 2, 2, 2, 3) X(2, 2, 2, 1, 1, 3) X(2, 2, 1, 2, 2, 2)
#endif
#define TA3N_INSTANTIATE_HS(wm, wn, wk, rm, rn, ns) \
    template __global__ void gemm_tiles<wm, wn, wk, 5, ns, rm, rn>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);

#define TA3N_INSTANTIATE_BLOCKED(wm, wn, wk, rm, rn, ns) \
    template __global__ void gemm_tiles<wm, wn, wk, 2, ns, rm, rn>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);


// kind-specialised kernels of the benchmarked bf16-twin step: (wm, wn, wk, mode, stages, KV)
// (bf16 twins: forward levels, the head-gradient level that rounds fp32 operands, backward levels; the same three for the split
// arithmetic on pair twins.  Measured on one MI355X, pipelined step: bf16 110.2 -> 107.8 us, split 143.7 -> 140.8 us.  The fp32-MFMA
// kernels are MFMA-bound and got 1.2 us SLOWER specialised (216.2 -> 217.4): none built.)
#define TA3N_KIND_CONFIGS(X) X(1, 2, 4, 2, 3, 1) X(1, 2, 4, 1, 2, 26) X(2, 2, 2, 2, 2, 26) \
                             X(2, 1, 4, 4, 3, 1) X(1, 2, 4, 3, 2, 26) X(1, 2, 2, 4, 2, 26)
#define TA3N_INSTANTIATE_KIND(wm, wn, wk, bf, ns, kv) \
    template __global__ void gemm_tiles<wm, wn, wk, bf, ns, 1, 1, kv>(const Task *, const Seg *, Ptrs, int, int, int, SgdSide, const Wait *, int, int, int, int);


}  // namespace ta3n
