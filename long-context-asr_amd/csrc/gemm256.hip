// bf16 MFMA GEMM, 256x256x64 block tile, 8 waves, for the large projections of the SConformerXL path (same contract and
// epilogues as gemm.hip; chosen by sconf_gemm_bf16 when M and N are multiples of 256 and K of 64).
//
// Why a second kernel: a 128x128 tile moves 2 x 128 x 64 x 2 B = 32 KiB through the CU's 64 B/clk vector-memory path
// for 2 x 128 x 128 x 64 flop, i.e. exactly the time the four SIMDs need for the MFMAs: the load path and the matrix
// cores are co-critical and any stall shows.  A 256x256 tile halves the bytes per flop, and a 128x64 wave tile cuts the
// LDS fragment reads per MFMA from 0.5 to 0.375.
//
// Structure (one workgroup per CU, 128 KiB LDS = 2 K-tile buffers x {A0, A1, B0, B1} half-tile images of 128 x 64):
//  * waves 2 (M) x 4 (N); a wave owns rows {64wr..+63} of BOTH A halves and 32 + 32 columns from the two B halves, so
//    each half-tile image is read in exactly one of the four phases of a K-tile and can be restaged right after it;
//  * a K-tile is 4 phases (one 64 x 32 quadrant of the wave tile each): { fragment ds_reads, one half-tile of LDS-DMA
//    prefetch, counted s_waitcnt vmcnt } -> s_barrier -> 16 MFMA -> s_barrier.  The DMA stream runs 5 phases ahead of
//    its consumer and is never drained inside the loop (raw s_barrier, never __syncthreads);
//  * the two wave rows run one barrier apart (wr = 1 executes one extra s_barrier per work item), so that while one
//    half of the waves is in its MFMA cluster the other half issues its LDS reads and DMA;
//  * persistent workgroups walk (K-split, tile) work items; the prefetch cursor crosses item boundaries, so an item's
//    epilogue stores overlap the next item's first loads.
//
// Hazards, by construction (phase P = 4 * k_tile + q; both wave rows issue their share of a half-tile in their own L_P):
//   WAR  an image read in L_P is restaged in L_(P+2) at the earliest: every reader has passed the lgkmcnt(0) that
//        precedes its MFMAs of phase P, and two barriers lie in between even for the lagging wave row;
//   RAW  the wait at the end of L_P leaves the 4 youngest half-tiles (8 DMA instructions per wave) in flight, so every
//        half-tile issued in phases <= P-4 has landed for the issuing wave; the barrier that ends L_P (L_P of the lagging
//        row included) makes that true for all waves before any L_(P+1) read.  Issue phases: A0, B0 of K-tile s+2 in
//        (s, q2), (s, q3); B1, A1 of K-tile s+1 in (s, q0), (s, q1); read phases: A0/B0 q0, B1 q1, A1 q2 - always >= 5
//        phases after the issue.
#include "gemm_tile.h"
#include <algorithm>
#include <stdlib.h>
#include <stdio.h>

SCONF_API int sconf_num_cus(void);

namespace {
using namespace gemm_tile;

// Tile = 256 rows x (128 + 64 * JH) columns: JH = 2 -> 256x256; JH = 1 -> 256x192 (NT only), which turns the 384 tiles of
// an N = 768 projection (1.5 rounds over 256 CUs) into 512 (exactly 2 rounds).
constexpr int TM = 256, TK = 64;
constexpr int HT = 128 * 64 * 2;                   // one half-tile image, 16 KiB (B1 of the 192-wide tile uses half of it)
constexpr int BUF = 4 * HT;                        // A0 | A1 | B0 | B1
constexpr int GM2 = 4;                             // row panels per L2 patch (4 x 8 tiles = the 32 workgroups of one XCD);
                                                   // 8 x 4 when the output is >= 8 tiles wide (measured +2-4 % at N >= 2304)

typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
typedef __attribute__((address_space(3))) bf16x4* lds_p4;

// K-contiguous images are [rows][64 k] with a 16-B-chunk XOR swizzle per row.
// A image: the half's 128 rows in natural order, read 16 consecutive rows at a time.
// B images: a wave owns WC = 32 + 16 * JH tile columns, WC * wc + (0 .. WC-1): the first 32 live in B0 (row 32wc + c), the
//   rest in B1 (row 16 JH wc + c - 32).  A wave reads the PERMUTED row set  base + 8 * (a >> 2) + 4j + (a & 3)  (a = lane & 15;
//   B1 of the 192-wide tile: base + 4 * (a >> 2) + (a & 3)) for column tile j, so that with the MFMA issued operand-swapped
//   lane (r, g) accumulates the 8 consecutive columns 8g .. 8g+7 of the wave's B0 part (and 8 or 4 of its B1 part): each
//   epilogue store instruction then writes 64 contiguous bytes per row.
__device__ __forceinline__ int swz_a(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_b(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
__device__ __forceinline__ int swz_b1n(int row) { return ((row >> 1) & 1) | (((row >> 2) & 3) << 1); }   // 64-row B1 image

// per-lane source byte offsets (relative to the half-tile's wave-uniform base) of the DMA pieces of each half
template <bool KS, bool BOP, int JH> struct DmaOffs {
    unsigned off[2][2];
    // rot (B operand of the 256-wide tile only): the wave's lo / hi column runs are d and d + 64 of ONE head (tile = 2 heads of 128),
    // so that a lane holds both partners of the rotary rotation: wave wc takes head wc >> 1, d in [32 (wc & 1), +32) | that + 64.
    __device__ __forceinline__ void set(long ld, int tid, bool rot = false) {
        constexpr int WC = 32 + 16 * JH;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + 512 * i;
                if (!KS) {
                    const int row = c >> 3, pos = c & 7;
                    int kc, mrow;
                    if (!BOP)              { kc = pos ^ swz_a(row); mrow = 128 * h + row; }
                    else if (h == 0)       { kc = pos ^ swz_b(row); mrow = rot ? 128 * (row >> 6) + 32 * ((row >> 5) & 1) + (row & 31) : WC * (row >> 5) + (row & 31); }
                    else if (JH == 2)      { kc = pos ^ swz_b(row); mrow = rot ? 128 * (row >> 6) + 32 * ((row >> 5) & 1) + 64 + (row & 31) : WC * (row >> 5) + 32 + (row & 31); }
                    else                   { kc = pos ^ swz_b1n(row & 63); mrow = WC * ((row & 63) >> 4) + 32 + (row & 15); }
                    off[h][i] = (unsigned)(((long)mrow * ld + kc * 8) * 2);
                } else {
                    const int kr = c >> 4, pos = c & 15, rc = pos ^ swz_strided(kr);
                    off[h][i] = (unsigned)(((long)kr * ld + 128 * h + rc * 8) * 2);
                }
            }
    }
};
template <int NP>
__device__ __forceinline__ void dma_half(const char* base, const unsigned (&off)[2], char* dst, int tid) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
        __builtin_amdgcn_global_load_lds((gptr)(base + off[i]), (lptr)(dst + ((tid & ~63) + 512 * i) * 16), 16, 0, 0);
}

// Transposed LDS read as inline asm.  Through the builtin, hipcc puts an unconditional `s_waitcnt vmcnt(0)` in front of the
// reads (it cannot tell them from the in-flight LDS-DMA destinations), which drains the whole prefetch pipeline twice per
// K-tile - the reason the K-strided (weight-gradient) kernel sat parked 57 % of its wave-cycles.  The asm form is invisible to
// that logic AND to the compiler's lgkmcnt tracking: the consumer must run LGKM_FENCE (wait + register tie) before using it.
__device__ __forceinline__ bf16x4 lds_read_tr(const char* p) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(unsigned long)(lptr)p));
    return v;
}
template <int OFF> __device__ __forceinline__ bf16x4 lds_read_tr_off(unsigned a) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
    return v;
}
// Both k-steps' fragments of one 16-wide block of a K-strided image from ONE per-lane base address: k-step 1 is +32 k-rows
// (8192 B) and the second half of each fragment +4 k-rows (1024 B); the XOR swizzle depends on (k & 3, (k >> 3) & 1), which
// those offsets leave unchanged, so all four reads are base + immediate.
__device__ __forceinline__ void frag_pair_ks(const char* s, int col, int lane, bf16x8& f0, bf16x8& f1) {
    const int a = lane & 15, q = a >> 2, g = lane >> 4;
    const int chunk = col >> 3, sub = (col & 4) * 2, k0 = 8 * g + q;
    const unsigned base = (unsigned)(unsigned long)(lptr)(s + k0 * 256 + ((chunk ^ swz_strided(k0)) << 4) + sub);
    const bf16x4 l0 = lds_read_tr_off<0>(base), h0 = lds_read_tr_off<1024>(base);
    const bf16x4 l1 = lds_read_tr_off<8192>(base), h1 = lds_read_tr_off<9216>(base);
    f0 = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
    f1 = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
}
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define TIE(x) asm volatile("" : "+v"(x))
// 16 x 32 MFMA fragments.  K-contiguous: one ds_read_b128; K-strided ([64 k][128 cols] image): two transposed reads.
template <bool KS> __device__ __forceinline__ bf16x8 frag_a(const char* s, int wr, int i, int kk, int lane) {
    if (!KS) {
        const int r = 64 * wr + 16 * i + (lane & 15), c = kk * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(s + r * 128 + ((c ^ swz_a(r)) << 4));
    } else {
        const int a = lane & 15, q = a >> 2, pp = a & 3, g = lane >> 4;
        const int col = 64 * wr + 16 * i + 4 * pp, chunk = col >> 3, sub = (col & 4) * 2;
        const int k0 = kk * 32 + 8 * g + q, k1 = k0 + 4;
        bf16x4 lo = lds_read_tr(s + k0 * 256 + ((chunk ^ swz_strided(k0)) << 4) + sub);
        bf16x4 hi = lds_read_tr(s + k1 * 256 + ((chunk ^ swz_strided(k1)) << 4) + sub);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
template <bool KS, bool NARROW> __device__ __forceinline__ bf16x8 frag_b(const char* s, int wc, int j, int kk, int lane) {
    if (!KS) {
        const int a = lane & 15, c = kk * 4 + (lane >> 4);
        if (!NARROW) {
            const int r = 32 * wc + 8 * (a >> 2) + 4 * j + (a & 3);
            return *reinterpret_cast<const bf16x8*>(s + r * 128 + ((c ^ swz_b(r)) << 4));
        } else {
            const int r = 16 * wc + 4 * (a >> 2) + (a & 3);
            return *reinterpret_cast<const bf16x8*>(s + r * 128 + ((c ^ swz_b1n(r)) << 4));
        }
    } else {
        const int a = lane & 15, q = a >> 2, pp = a & 3, g = lane >> 4;
        const int col = 32 * wc + 16 * j + 4 * pp, chunk = col >> 3, sub = (col & 4) * 2;
        const int k0 = kk * 32 + 8 * g + q, k1 = k0 + 4;
        bf16x4 lo = lds_read_tr(s + k0 * 256 + ((chunk ^ swz_strided(k0)) << 4) + sub);
        bf16x4 hi = lds_read_tr(s + k1 * 256 + ((chunk ^ swz_strided(k1)) << 4) + sub);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// Work items (K-split, 256x256 tile): XCD-contiguous ids, split-major, GM2 x tiles_n patches (see gemm.hip tile_coords).
struct Item { int m0, n0, kbeg, nkt, split; };
struct Sched { int total, ntiles, tiles_m, tiles_n, tn, gm; };
__device__ __forceinline__ Item item_coords(const GemmParams& p, const Sched& sc, int v) {
    const int qx = sc.total >> 3, rx = sc.total & 7, xcd = v & 7;
    const int lid = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + (v >> 3);
    const int split = lid / sc.ntiles, t = lid - split * sc.ntiles;
    const int per_group = sc.gm * sc.tiles_n;
    const int g = t / per_group, r = t - g * per_group;
    const int first_m = g * sc.gm, gm = min(sc.gm, sc.tiles_m - first_m);
    Item w;
    w.m0 = (first_m + r % gm) * TM; w.n0 = (r / gm) * sc.tn;
    w.split = split;
    w.kbeg = split * p.k_per_split;
    w.nkt = (min(p.K, w.kbeg + p.k_per_split) - w.kbeg) / TK;
    return w;
}
// position in the workgroup's stream of K-tiles (all wave-uniform)
struct Cursor {
    int v, kt, par; bool valid; Item it;
    __device__ __forceinline__ void advance(const GemmParams& p, const Sched& sc) {
        par ^= 1;
        if (++kt < it.nkt) return;
        kt = 0; v += gridDim.x;
        if (v < sc.total) it = item_coords(p, sc, v); else valid = false;
    }
};

// Epilogue of one work item for one wave (ACT and the presence of a residual fixed at compile time).  All loads are issued
// before the stores they could queue behind: the bias run once, the aux tile (bf16) per half of 4 row blocks, the residual
// (f32) per pair of row blocks - as far ahead as the 256-VGPR budget beside the 128 accumulator registers allows.
template <int W> struct RawBf;
template <> struct RawBf<8> { typedef bf16x8 type; };
template <> struct RawBf<4> { typedef bf16x4 type; };
template <int ACT, bool RES, bool KS, int JH, int FL = -1, bool ROT = false, int DEEP = 0>
__device__ __forceinline__ void epilogue256(const GemmParams& p, const f32x4 (&acc)[2][4][2 + JH], const Item& it, int wr, int wc, int lane) {
    constexpr int WC = 32 + 16 * JH, WH = 4 * JH;
    constexpr bool AUX = ACT == SCONF_ACT_DGELU || ACT == SCONF_ACT_DSILU || ACT == SCONF_ACT_MULAUX || ACT == SCONF_ACT_SMAXBWD;
    constexpr bool SMB = ACT == SCONF_ACT_SMAXBWD;    // softmax backward: per-row scalar in, column sums out
    const int g = lane >> 4, mbase = it.m0 + 64 * wr + (lane & 15);
    if constexpr (ROT) {
        // qkv projection with the rotary rotation: with the rot row permutation (DmaOffs::set) the lane's lo run is (head, d0 .. d0 + 7)
        // and its hi run (head, d0 + 64 ..): out[d] = x[d] cos - x[d + 64] sin, out[d + 64] = x[d + 64] cos + x[d] sin
        // (apply_rotary_pos_emb, rotary_emb.py:61-73) on the f32 accumulators - one bf16 rounding instead of two.  Tiles past
        // rot_cols (the v block) are stored as they are.  cos / sin of the next row block are requested before this one is stored.
        static_assert(JH == 2 && !RES && !AUX && ACT == SCONF_ACT_NONE && FL >= 0 && !(FL & 2), "rotary epilogue: 256-wide tile, plain bf16 output");
        const int d0 = 32 * (wc & 1) + 8 * g;
        const int nlo = it.n0 + 128 * (wc >> 1) + d0, nhi = nlo + 64;
        float blo[8], bhi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { blo[e] = 0.f; bhi[e] = 0.f; }
        if (FL & 1) { loadv<8>(p.bias + nlo, blo); loadv<8>(p.bias + nhi, bhi); }
        const bool rot = it.n0 < p.rot_cols;                                       // uniform over the workgroup
        const long crow0 = (long)mbase * p.ldc;
        float cs[2][8], sn[2][8];
        auto fetch = [&](int rb, int slot) {
            const int pos = (mbase + 128 * (rb >> 2) + 16 * (rb & 3)) % p.rot_n;
            loadv<8>(p.rot_cos + (long)pos * 64 + d0, cs[slot]); loadv<8>(p.rot_sin + (long)pos * 64 + d0, sn[slot]);
        };
        if (rot) fetch(0, 0);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            if (rot && rb + 1 < 8) fetch(rb + 1, (rb + 1) & 1);
            const int h = rb >> 2, i = rb & 3, sl = rb & 1;
            float lo[8], hi[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { lo[e] = acc[h][i][e >> 2][e & 3] + blo[e]; hi[e] = acc[h][i][2 + (e >> 2)][e & 3] + bhi[e]; }
            if (rot) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = lo[e], b = hi[e];
                    lo[e] = a * cs[sl][e] - b * sn[sl][e]; hi[e] = b * cs[sl][e] + a * sn[sl][e];
                }
            }
            const long crow = crow0 + (long)(128 * h + 16 * i) * p.ldc;
            storev<8>(reinterpret_cast<bf16*>(p.C) + crow + nlo, lo); storev<8>(reinterpret_cast<bf16*>(p.C) + crow + nhi, hi);
        }
        return;
    }
    if constexpr (!KS) {
        const int nlo = it.n0 + WC * wc + 8 * g, nhi = it.n0 + WC * wc + 32 + WH * g;
        float blo[8], bhi[WH];
#pragma unroll
        for (int e = 0; e < 8; ++e) blo[e] = 0.f;
#pragma unroll
        for (int e = 0; e < WH; ++e) bhi[e] = 0.f;
        if (FL >= 0 ? (FL & 1) != 0 : p.bias != nullptr) { loadv<8>(p.bias + nlo, blo); loadv<WH>(p.bias + nhi, bhi); }
        const long crow0 = (long)mbase * p.ldc, prow0 = (long)mbase * p.ldpre;      // per-lane row offsets; the rest is wave-uniform
        if constexpr (FL >= 0 && (RES || ACT == SCONF_ACT_MULAUX || SMB)) {
            // Software-pipelined over the 8 row blocks: residual / aux loads are issued blocks ahead of the math and stores that
            // use them, so by the time they are needed the stores they queue behind have long drained.  (Loading each pair of blocks right after the previous pair's stores made every pair wait for a full
            // store round trip: 38-41 k cycles for an f32 + residual tile against ~8 k for a bf16 one.)
            constexpr int NB = 8;                             // row blocks
            // register slots for blocks in flight (f32 residual: 16 VGPRs per block, aux: 8).  DEEP (the one-epilogue kernels, which have
            // ~30 VGPRs to spare): all aux blocks / 3-4 residual blocks requested up front - one exposed round trip instead of two.
            constexpr int NS = DEEP ? (RES ? (JH == 2 ? 4 : 5) : 8) : (RES ? (JH == 2 ? 2 : 3) : 5);
            constexpr bool FIXED = RES || DEEP;              // fixed look-ahead of NS - 1 blocks
            float rl[NS][8], rh[NS][WH];
            typename RawBf<8>::type xl[NS];
            typename RawBf<WH>::type xh[NS];
            float rowv[NS], cslo[8], cshi[WH];
            if constexpr (SMB) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cslo[e] = 0.f;
#pragma unroll
                for (int e = 0; e < WH; ++e) cshi[e] = 0.f;
            }
            auto fetch = [&](int rb, int slot) {
                const int row = mbase + 128 * (rb >> 2) + 16 * (rb & 3);
                if constexpr (SMB) rowv[slot] = p.rowv[row];
                if constexpr (RES) { const float* q = p.resid + (long)row * p.ldr; loadv<8>(q + nlo, rl[slot]); loadv<WH>(q + nhi, rh[slot]); }
                if constexpr (AUX) {
                    const bf16* q = p.aux + (long)row * p.ldaux;
                    xl[slot] = *reinterpret_cast<const typename RawBf<8>::type*>(q + nlo);
                    xh[slot] = *reinterpret_cast<const typename RawBf<WH>::type*>(q + nhi);
                }
            };
#pragma unroll
            for (int nb = 0; nb < (FIXED ? NS - 1 : 1); ++nb) fetch(nb, nb);
            // Aux tiles (bf16, 8 VGPRs per block): the look-ahead GROWS - blocks 1 | 2,3 | 4,5 | 6,7 are requested before blocks
            // 0 | 1 | 2 | 3 are processed, so after the first two round trips every remaining load is already in flight (one block
            // ahead throughout left 8 exposed round trips: 25 k cycles for the aux tile of an FF dgrad item).
#pragma unroll
            for (int rb = 0; rb < NB; ++rb) {
                {
                    // residual (arch-VGPR bound): fixed look-ahead of NS - 1 blocks; aux only: the growing schedule
                    const int lo_b = FIXED ? rb + NS - 1 : (rb == 0 ? 1 : 2 * rb), hi_b = FIXED ? rb + NS - 1 : 2 * rb + 1;
#pragma unroll
                    for (int nb = lo_b; nb <= hi_b; ++nb) if (nb < NB) fetch(nb, nb % NS);
                }
                const int h = rb >> 2, i = rb & 3, sl = rb % NS;
                float vlo[8], vhi[WH], alo[8], ahi[WH], zlo[8], zhi[WH];
#pragma unroll
                for (int e = 0; e < 8; ++e) { vlo[e] = acc[h][i][e >> 2][e & 3]; alo[e] = AUX ? (float)xl[sl][e] : 0.f; zlo[e] = RES ? rl[sl][e] : (SMB ? rowv[sl] : 0.f); }
#pragma unroll
                for (int e = 0; e < WH; ++e) { vhi[e] = acc[h][i][2 + (e >> 2)][e & 3]; ahi[e] = AUX ? (float)xh[sl][e] : 0.f; zhi[e] = RES ? rh[sl][e] : (SMB ? rowv[sl] : 0.f); }
                const long crow = crow0 + (long)(128 * h + 16 * i) * p.ldc, prow = prow0 + (long)(128 * h + 16 * i) * p.ldpre;
                epi_math_store_at<8, ACT, FL>(p, vlo, blo, alo, zlo, crow + nlo, prow + nlo, it.split);
                epi_math_store_at<WH, ACT, FL>(p, vhi, bhi, ahi, zhi, crow + nhi, prow + nhi, it.split);
                if constexpr (SMB) {                           // (epi_math_store_at leaves the stored values in vlo / vhi)
#pragma unroll
                    for (int e = 0; e < 8; ++e) cslo[e] += vlo[e];
#pragma unroll
                    for (int e = 0; e < WH; ++e) cshi[e] += vhi[e];
                }
            }
            if constexpr (SMB) {
                // column sums of the wave's 128 rows: the 16 lanes of a DPP row hold 16 different rows of the same columns - butterfly
                // (quad swaps, half-row mirror, row mirror), then lane 0 of each row stores its 8 + WH sums to the wave row's slab line
                auto row_sum = [](float v) {
                    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
                    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
                    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
                    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
                    return v;
                };
#pragma unroll
                for (int e = 0; e < 8; ++e) cslo[e] = row_sum(cslo[e]);
#pragma unroll
                for (int e = 0; e < WH; ++e) cshi[e] = row_sum(cshi[e]);
                if ((lane & 15) == 0) {
                    float* q = p.colslab + (long)(2 * (it.m0 / TM) + wr) * p.N;
                    storev<8>(q + nlo, cslo); storev<WH>(q + nhi, cshi);
                }
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            typename RawBf<8>::type axlo[4];
            typename RawBf<WH>::type axhi[4];
            if constexpr (AUX) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16* q = p.aux + (long)(mbase + 128 * h + 16 * i) * p.ldaux;
                    axlo[i] = *reinterpret_cast<const typename RawBf<8>::type*>(q + nlo);
                    axhi[i] = *reinterpret_cast<const typename RawBf<WH>::type*>(q + nhi);
                }
            }
#pragma unroll
            for (int i2 = 0; i2 < 4; i2 += 2) {
                float rlo[2][8], rhi[2][WH];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rlo[ii][e] = 0.f;
#pragma unroll
                    for (int e = 0; e < WH; ++e) rhi[ii][e] = 0.f;
                    if constexpr (RES) {
                        const float* q = p.resid + (long)(mbase + 128 * h + 16 * (i2 + ii)) * p.ldr;
                        loadv<8>(q + nlo, rlo[ii]); loadv<WH>(q + nhi, rhi[ii]);
                    }
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = i2 + ii;
                    float vlo[8], vhi[WH], alo[8], ahi[WH];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { vlo[e] = acc[h][i][e >> 2][e & 3]; alo[e] = AUX ? (float)axlo[i][e] : 0.f; }
#pragma unroll
                    for (int e = 0; e < WH; ++e) { vhi[e] = acc[h][i][2 + (e >> 2)][e & 3]; ahi[e] = AUX ? (float)axhi[i][e] : 0.f; }
                    const long crow = crow0 + (long)(128 * h + 16 * i) * p.ldc, prow = prow0 + (long)(128 * h + 16 * i) * p.ldpre;
                    epi_math_store_at<8, ACT, FL>(p, vlo, blo, alo, rlo[ii], crow + nlo, prow + nlo, it.split);
                    epi_math_store_at<WH, ACT, FL>(p, vhi, bhi, ahi, rhi[ii], crow + nhi, prow + nhi, it.split);
                }
            }
        }
    } else {
        // K-strided operands keep natural column order: 4 consecutive columns per lane and 16x16 tile (plain epilogue only)
        const int nb = it.n0 + 32 * wc + 4 * g;
        const int ncol[4] = {nb, nb + 16, nb + 128, nb + 144};
        float zero[4] = {0.f, 0.f, 0.f, 0.f}, bs[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bs[j][e] = 0.f;
            if (p.bias) loadv<4>(p.bias + ncol[j], bs[j]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[4] = {acc[h][i][j][0], acc[h][i][j][1], acc[h][i][j][2], acc[h][i][j][3]};
                    epi_math_store<4, ACT>(p, v, bs[j], zero, zero, mbase + 128 * h + 16 * i, ncol[j], it.split);
                }
    }
}

// One specialised, contiguous epilogue per (activation, residual, bias, output type) combination the model's NT GEMMs use;
// `kind` is computed once per launch (wave-uniform), everything else takes the runtime-flag path.
//   0 bf16 plain   1 bf16 + bias   2 f32 + residual   3 f32 + residual + bias   4 gelu' save (no bias)   5 * aux   6 generic
//   7 bf16 + rotary   8 bf16 + bias + rotary   (256-wide tile only)
//   9 f32 + residual + bias, acc + bias also saved in bf16 (the self-conditioning reprojection)   10 softmax backward: (acc - rowv) * aux + column sums
//   11 f32 plain   12 f32 + bias   (the head's logits, the subsampler's output projection)
//   - 9 .. 12 live only in kernel instantiations of their own (template parameter EK): in the common kernels their register needs made
//     hipcc spill loop-carried values around EVERY item's epilogue (18 VGPRs to scratch, reloaded behind a vmcnt(0))
__host__ __device__ __forceinline__ int epilogue_kind(const GemmParams& p) {
    if (p.rot_cos) return p.bias ? 8 : 7;                        // (the launcher has checked everything else)
    const bool b = p.bias != nullptr, unit = p.alpha == 1.f;   // kinds 0, 1, 4 skip the alpha * v + residual step (the FF dgrad, kind 5, carries the branch scale in alpha)
    if (p.act == SCONF_ACT_GELU_DSAVE) return (!b && !p.out_f32 && !p.resid && unit) ? 4 : 6;
    if (p.act == SCONF_ACT_MULAUX) return (!b && !p.out_f32 && !p.resid && !p.pre) ? 5 : 6;
    if (p.act == SCONF_ACT_SMAXBWD) return 10;                  // (the launcher has checked everything else)
    if (p.act == SCONF_ACT_NONE && p.pre && p.out_f32 && p.resid && b) return 9;
    if (p.act != SCONF_ACT_NONE || p.pre) return 6;
    if (!p.out_f32 && !p.resid) return unit ? (b ? 1 : 0) : 6;
    if (p.out_f32 && p.resid) return b ? 3 : 2;
    if (p.out_f32 && unit) return b ? 12 : 11;
    return 6;
}
#define EPILOGUE_NT(JH_)                                                                                              \
    switch (ekind) {                                                                                                  \
        case 0: epilogue256<SCONF_ACT_NONE, false, false, JH_, 4>(p, acc, cit, wr, wc, lane); break;                  \
        case 1: epilogue256<SCONF_ACT_NONE, false, false, JH_, 5>(p, acc, cit, wr, wc, lane); break;                  \
        case 2: epilogue256<SCONF_ACT_NONE, true, false, JH_, 2>(p, acc, cit, wr, wc, lane); break;                   \
        case 3: epilogue256<SCONF_ACT_NONE, true, false, JH_, 3>(p, acc, cit, wr, wc, lane); break;                   \
        case 4: epilogue256<SCONF_ACT_GELU_DSAVE, false, false, JH_, 4>(p, acc, cit, wr, wc, lane); break;            \
        case 5: epilogue256<SCONF_ACT_MULAUX, false, false, JH_, 0>(p, acc, cit, wr, wc, lane); break;                \
        case 7: if constexpr (JH_ == 2) epilogue256<SCONF_ACT_NONE, false, false, JH_, 0, true>(p, acc, cit, wr, wc, lane); break;  \
        case 8: if constexpr (JH_ == 2) epilogue256<SCONF_ACT_NONE, false, false, JH_, 1, true>(p, acc, cit, wr, wc, lane); break;  \
        default:                                                                                                      \
            if (p.act == SCONF_ACT_GELU_DSAVE)  epilogue256<SCONF_ACT_GELU_DSAVE, false, false, JH_>(p, acc, cit, wr, wc, lane); \
            else if (p.act == SCONF_ACT_MULAUX) epilogue256<SCONF_ACT_MULAUX, false, false, JH_>(p, acc, cit, wr, wc, lane);     \
            else if (p.resid)                   epilogue256<SCONF_ACT_NONE, true, false, JH_>(p, acc, cit, wr, wc, lane);        \
            else                                epilogue256<SCONF_ACT_NONE, false, false, JH_>(p, acc, cit, wr, wc, lane);       \
    }

// One kernel instantiation per specialised epilogue (template parameter EK = the kind): a kernel that holds ONE epilogue is allocated
// for that epilogue alone (206-226 VGPRs, no SGPR spills, against 255 + 6 for the kernel that switches between all of them).
#ifndef DEEPV
#define DEEPV 1
#endif
#define EPILOGUE_ONE(EK_, JH_)                                                                                        \
    do {                                                                                                              \
        if constexpr (EK_ == 0) epilogue256<SCONF_ACT_NONE, false, false, JH_, 4>(p, acc, cit, wr, wc, lane);         \
        else if constexpr (EK_ == 1) epilogue256<SCONF_ACT_NONE, false, false, JH_, 5>(p, acc, cit, wr, wc, lane);    \
        else if constexpr (EK_ == 2) epilogue256<SCONF_ACT_NONE, true, false, JH_, 2, false, DEEPV>(p, acc, cit, wr, wc, lane);     \
        else if constexpr (EK_ == 3) epilogue256<SCONF_ACT_NONE, true, false, JH_, 3, false, DEEPV>(p, acc, cit, wr, wc, lane);     \
        else if constexpr (EK_ == 4) epilogue256<SCONF_ACT_GELU_DSAVE, false, false, JH_, 4>(p, acc, cit, wr, wc, lane); \
        else if constexpr (EK_ == 5) epilogue256<SCONF_ACT_MULAUX, false, false, JH_, 0, false, DEEPV>(p, acc, cit, wr, wc, lane);  \
        else if constexpr (EK_ == 7) epilogue256<SCONF_ACT_NONE, false, false, JH_, 0, true>(p, acc, cit, wr, wc, lane); \
        else if constexpr (EK_ == 8) epilogue256<SCONF_ACT_NONE, false, false, JH_, 1, true>(p, acc, cit, wr, wc, lane); \
        else if constexpr (EK_ == 9) epilogue256<SCONF_ACT_NONE, true, false, JH_, 11, false, DEEPV>(p, acc, cit, wr, wc, lane);    \
        else if constexpr (EK_ == 11) epilogue256<SCONF_ACT_NONE, false, false, JH_, 6>(p, acc, cit, wr, wc, lane);   \
        else if constexpr (EK_ == 12) epilogue256<SCONF_ACT_NONE, false, false, JH_, 7>(p, acc, cit, wr, wc, lane);   \
        else epilogue256<SCONF_ACT_SMAXBWD, false, false, JH_, 4, false, DEEPV>(p, acc, cit, wr, wc, lane);                         \
    } while (0)

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// In-kernel time stamps (cdna_hip_programming.md section 7), DIAGNOSTIC build only (make EXTRA=-DSCONF_GEMM_STAMP; tools/gemm_stamp.py):
// per wave, cycle sums of the fragment reads, the LDS-DMA issue, the counted wait, the waits at the two barriers of a phase, the
// MFMA clusters and the epilogue.  Nothing of it exists in the product build.
#ifdef SCONF_GEMM_STAMP
__device__ unsigned long long g_gemm_stamps[256 * 8 * 8];
#define GSTAMP_DECL unsigned long long st_acc_[8] = {}, st_last_ = 0; { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_last_ = t_; }
#define GSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                       __builtin_amdgcn_sched_barrier(0); st_acc_[i] += t_ - st_last_; st_last_ = t_; } while (0)
#define GSTAMP_OUT do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) for (int i_ = 0; i_ < 8; ++i_) g_gemm_stamps[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + i_] = st_acc_[i_]; } while (0)
#else
#define GSTAMP_DECL
#define GSTAMP(i)
#define GSTAMP_OUT
#endif
// leave the 4 youngest half-tiles in flight: 2 + 2 + 2 + JH DMA instructions per wave
template <int JH> __device__ __forceinline__ void wait_window(bool streaming) {
    if (!streaming) VMCNT(0);
    else if (JH == 2) VMCNT(8);
    else VMCNT(7);
}

template <bool KS, int JH, int EK = -1>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
    constexpr int WC = 32 + 16 * JH, TN = 4 * WC;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A0 | A1 | B0 | B1], 128 KiB: the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    Sched sc;
    sc.tn = TN; sc.tiles_n = p.N / TN; sc.tiles_m = p.M / TM; sc.ntiles = sc.tiles_m * sc.tiles_n; sc.total = sc.ntiles * p.splits; sc.gm = p.gm > 0 ? p.gm : (sc.tiles_n >= 8 ? 2 * GM2 : GM2);
    if ((int)blockIdx.x >= sc.total) return;
#ifdef SCONF_GEMM_PROBE
    if (p.stagger > 0) {
        const int b = blockIdx.x;
        const int ph = p.stagger_mode == 0 ? (b & 7) : p.stagger_mode == 1 ? ((b >> 3) & 7) : ((b + (b >> 3)) & 7);
        for (int i = 0; i < ph * p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
    int probe_item = 0;
#endif

    const int ekind = __builtin_amdgcn_readfirstlane(epilogue_kind(p));
    DmaOffs<KS, false, JH> oa; DmaOffs<KS, true, JH> ob;
    oa.set(p.lda, tid); ob.set(p.ldb, tid, !KS && JH == 2 && p.rot_cos != nullptr);
    auto issue_a = [&](const Cursor& c, int h) {
        if (!c.valid) return;
        const long k0 = c.it.kbeg + c.kt * TK;
        const char* base = reinterpret_cast<const char*>(p.A) + (KS ? k0 * p.lda + c.it.m0 : (long)c.it.m0 * p.lda + k0) * 2;
        dma_half<2>(base, oa.off[h], smem + c.par * BUF + h * HT, tid);
    };
    auto issue_b = [&](const Cursor& c, int h) {
        if (!c.valid) return;
        const long k0 = c.it.kbeg + c.kt * TK;
        const char* base = reinterpret_cast<const char*>(p.B) + (KS ? k0 * p.ldb + c.it.n0 : (long)c.it.n0 * p.ldb + k0) * 2;
        if (h == 0) dma_half<2>(base, ob.off[0], smem + c.par * BUF + 2 * HT, tid);
        else        dma_half<JH>(base, ob.off[1], smem + c.par * BUF + 3 * HT, tid);
    };

    f32x4 acc[2][4][2 + JH];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2 + JH; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Cursor pc;                                        // prefetch cursor
    pc.v = blockIdx.x; pc.kt = 0; pc.par = 0; pc.valid = true; pc.it = item_coords(p, sc, pc.v);
    int cv = pc.v; Item cit = pc.it;                  // compute position
    // prologue: K-tile 0 entirely, A0/B0 of K-tile 1 (the "phases -6 .. -1" of the schedule)
    issue_a(pc, 0); issue_b(pc, 0); issue_b(pc, 1); issue_a(pc, 1);
    pc.advance(p, sc);
    issue_a(pc, 0); issue_b(pc, 0);
    if (pc.valid) wait_window<JH>(true);             // A0, B0 of K-tile 0 have landed
    else if (JH == 2) VMCNT(4); else VMCNT(3);
    __builtin_amdgcn_s_barrier();

    int cur = 0;
    bf16x8 fa[4][2], blo[2][2], bhi[JH][2];
    GSTAMP_DECL
    while (true) {
        if (wr) __builtin_amdgcn_s_barrier();         // stagger the second wave row by one barrier
        for (int kt = 0; kt < cit.nkt; ++kt) {
            const char* buf = smem + cur * BUF;
            const bool last = kt == cit.nkt - 1;
            // ---- q0: B-lo, A-lo -> acc[0][.][0..1] ---------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (KS) frag_pair_ks(buf + 2 * HT, 32 * wc + 16 * j + 4 * (lane & 3), lane, blo[j][0], blo[j][1]);
                else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) blo[j][kk] = frag_b<KS, false>(buf + 2 * HT, wc, j, kk, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (KS) frag_pair_ks(buf, 64 * wr + 16 * i + 4 * (lane & 3), lane, fa[i][0], fa[i][1]);
                else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) fa[i][kk] = frag_a<KS>(buf, wr, i, kk, lane);
                }
            }
            GSTAMP(6);
            issue_b(pc, 1);
            GSTAMP(7);
            wait_window<JH>(pc.valid);
            GSTAMP(0);
            __builtin_amdgcn_s_barrier();
            GSTAMP(1);
            if constexpr (KS) {
                LGKM0();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) TIE(blo[j][kk]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) TIE(fa[i][kk]);
                }
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo[j][kk], fa[i][kk], acc[0][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            GSTAMP(2);
            __builtin_amdgcn_s_barrier();
            GSTAMP(3);
            // ---- q1: B-hi -> acc[0][.][2..3] ---------------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < JH; ++j) {
                if constexpr (KS) frag_pair_ks(buf + 3 * HT, 32 * wc + 16 * j + 4 * (lane & 3), lane, bhi[j][0], bhi[j][1]);
                else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) bhi[j][kk] = frag_b<KS, JH == 1>(buf + 3 * HT, wc, j, kk, lane);
                }
            }
            GSTAMP(6);
            issue_a(pc, 1);
            GSTAMP(7);
            wait_window<JH>(pc.valid);
            GSTAMP(0);
            __builtin_amdgcn_s_barrier();
            GSTAMP(1);
            if constexpr (KS) {
                LGKM0();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < JH; ++j) TIE(bhi[j][kk]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < JH; ++j)
                        acc[0][i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi[j][kk], fa[i][kk], acc[0][i][2 + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            GSTAMP(2);
            __builtin_amdgcn_s_barrier();
            GSTAMP(3);
            // ---- q2: A-hi -> acc[1][.][2..3] ---------------------------------------------------------------------
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (KS) frag_pair_ks(buf + HT, 64 * wr + 16 * i + 4 * (lane & 3), lane, fa[i][0], fa[i][1]);
                else {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) fa[i][kk] = frag_a<KS>(buf + HT, wr, i, kk, lane);
                }
            }
            pc.advance(p, sc);
            GSTAMP(6);
            issue_a(pc, 0);
            GSTAMP(7);
            wait_window<JH>(pc.valid);
            GSTAMP(0);
            __builtin_amdgcn_s_barrier();
            GSTAMP(1);
            if constexpr (KS) {
                LGKM0();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) TIE(fa[i][kk]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < JH; ++j)
                        acc[1][i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi[j][kk], fa[i][kk], acc[1][i][2 + j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            GSTAMP(2);
            __builtin_amdgcn_s_barrier();
            GSTAMP(3);
            // ---- q3: (B-lo still in registers) -> acc[1][.][0..1] ------------------------------------------------
            GSTAMP(6);
            issue_b(pc, 0);
            GSTAMP(7);
            wait_window<JH>(pc.valid);
            GSTAMP(0);
            __builtin_amdgcn_s_barrier();
            GSTAMP(1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo[j][kk], fa[i][kk], acc[1][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            GSTAMP(2);
            if (!(wr && last)) __builtin_amdgcn_s_barrier();   // the lagging row goes straight into its epilogue
            GSTAMP(3);
            cur ^= 1;
        }
        // ---- epilogue: both wave rows concurrently; one specialised, contiguous code path per (activation, residual) ------
#ifdef SCONF_GEMM_PROBE
        long long* st = (p.stamps && tid == 0 && probe_item < 64) ? p.stamps + ((long)blockIdx.x * 64 + probe_item) * 4 : nullptr;
        if (st) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = __builtin_amdgcn_s_memtime(); }
        if (p.debug != 2 || acc[0][0][0][0] == 1.2345e-30f)
#endif
        {
            if constexpr (KS) epilogue256<SCONF_ACT_NONE, false, true, JH>(p, acc, cit, wr, wc, lane);
            else if constexpr (EK >= 0) EPILOGUE_ONE(EK, JH);
            else EPILOGUE_NT(JH);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2 + JH; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef SCONF_GEMM_PROBE
        if (st) { st[2] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[3] = __builtin_amdgcn_s_memtime(); }
        ++probe_item;
#endif
        GSTAMP(4);                                    // epilogue (+ accumulator clears)
        __builtin_amdgcn_s_barrier();                 // re-align the two wave rows
        GSTAMP(5);
        cv += gridDim.x;
        if (cv >= sc.total) break;
        cit = item_coords(p, sc, cv);
    }
    VMCNT(0);
    GSTAMP_OUT;
}

// ---- 256x192 tile, NT, THREE phases per K-tile -------------------------------------------------------------------------
// The 4-phase schedule above on a 192-wide tile has MFMA clusters of 16 / 8 / 8 / 16 (the hi column half is one 16-wide
// tile per wave), so in the two short phases the other wave row's load section outlasts the MFMAs.  With both A quadrants
// resident (32 more VGPRs; the 192-wide tile has 32 fewer accumulators) the K-tile folds into three 16-MFMA phases:
//   P0  read B-lo, A-lo     issue A1, B1 of K-tile s+1    MFMA  A-lo x B-lo
//   P1  read B-hi, A-hi     -                             MFMA  A-lo x B-hi, A-hi x B-hi
//   P2  -                   issue A0, B0 of K-tile s+2    MFMA  A-hi x B-lo
// WAR: A0/B0 are read in P0 and restaged in P2 of the same K-tile, A1/B1 are read in P1 and restaged in P0 of the next one -
// two phases after the read, as above.  RAW: 7 DMA instructions per wave and K-tile (2 + 2 + 2 + 1); the wait at the end of
// a phase leaves the 7 youngest in flight, which retires A0/B0(s+2) at the end of P2(s+1) and A1/B1(s+2) at the end of
// P0(s+2) - each one phase (and one barrier, two for the lagging wave row's partner) before its first read.
template <int EK = -1>
__global__ __launch_bounds__(512) void gemm192_kernel(const GemmParams p) {
    constexpr int JH = 1, WC = 48, TN = 192;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A0 | A1 | B0 | B1], the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    Sched sc;
    sc.tn = TN; sc.tiles_n = p.N / TN; sc.tiles_m = p.M / TM; sc.ntiles = sc.tiles_m * sc.tiles_n; sc.total = sc.ntiles * p.splits; sc.gm = p.gm > 0 ? p.gm : (sc.tiles_n >= 8 ? 2 * GM2 : GM2);
    if ((int)blockIdx.x >= sc.total) return;

    const int ekind = __builtin_amdgcn_readfirstlane(epilogue_kind(p));
    DmaOffs<false, false, JH> oa; DmaOffs<false, true, JH> ob;
    oa.set(p.lda, tid); ob.set(p.ldb, tid);
    auto issue_a = [&](const Cursor& c, int h) {
        if (!c.valid) return;
        const long k0 = c.it.kbeg + c.kt * TK;
        dma_half<2>(reinterpret_cast<const char*>(p.A) + ((long)c.it.m0 * p.lda + k0) * 2, oa.off[h], smem + c.par * BUF + h * HT, tid);
    };
    auto issue_b = [&](const Cursor& c, int h) {
        if (!c.valid) return;
        const long k0 = c.it.kbeg + c.kt * TK;
        const char* base = reinterpret_cast<const char*>(p.B) + ((long)c.it.n0 * p.ldb + k0) * 2;
        if (h == 0) dma_half<2>(base, ob.off[0], smem + c.par * BUF + 2 * HT, tid);
        else        dma_half<1>(base, ob.off[1], smem + c.par * BUF + 3 * HT, tid);
    };

    f32x4 acc[2][4][3];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Cursor pc;                                        // prefetch cursor: the K-tile whose A0/B0 were issued last
    pc.v = blockIdx.x; pc.kt = 0; pc.par = 0; pc.valid = true; pc.it = item_coords(p, sc, pc.v);
    int cv = pc.v; Item cit = pc.it;
    // prologue: K-tile 0 entirely, A0/B0 of K-tile 1
    issue_a(pc, 0); issue_b(pc, 0); issue_a(pc, 1); issue_b(pc, 1);
    pc.advance(p, sc);
    issue_a(pc, 0); issue_b(pc, 0);
    if (pc.valid) VMCNT(7); else VMCNT(3);           // A0, B0 of K-tile 0 have landed
    __builtin_amdgcn_s_barrier();

    int cur = 0;
    bf16x8 flo[4][2], fhi[4][2], blo[2][2], bhi[2];
    while (true) {
        if (wr) __builtin_amdgcn_s_barrier();         // stagger the second wave row by one barrier
        for (int kt = 0; kt < cit.nkt; ++kt) {
            const char* buf = smem + cur * BUF;
            const bool last = kt == cit.nkt - 1;
            // ---- P0 ------------------------------------------------------------------------------------------------
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) blo[j][kk] = frag_b<false, false>(buf + 2 * HT, wc, j, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) flo[i][kk] = frag_a<false>(buf, wr, i, kk, lane);
            issue_a(pc, 1); issue_b(pc, 1);             // pc = K-tile s+1
            wait_window<JH>(pc.valid);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo[j][kk], flo[i][kk], acc[0][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- P1 ------------------------------------------------------------------------------------------------
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bhi[kk] = frag_b<false, true>(buf + 3 * HT, wc, 0, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fhi[i][kk] = frag_a<false>(buf + HT, wr, i, kk, lane);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0][i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi[kk], flo[i][kk], acc[0][i][2], 0, 0, 0);
                    acc[1][i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi[kk], fhi[i][kk], acc[1][i][2], 0, 0, 0);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- P2 ------------------------------------------------------------------------------------------------
            pc.advance(p, sc);                          // pc = K-tile s+2
            issue_a(pc, 0); issue_b(pc, 0);
            wait_window<JH>(pc.valid);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo[j][kk], fhi[i][kk], acc[1][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (!(wr && last)) __builtin_amdgcn_s_barrier();   // the lagging row goes straight into its epilogue
            cur ^= 1;
        }
#ifdef SCONF_GEMM_PROBE
        if (p.debug != 2 || acc[0][0][0][0] == 1.2345e-30f)
#endif
        {
            if constexpr (EK >= 0) EPILOGUE_ONE(EK, JH);
            else EPILOGUE_NT(JH);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_barrier();                 // re-align the two wave rows
        cv += gridDim.x;
        if (cv >= sc.total) break;
        cit = item_coords(p, sc, cv);
    }
    VMCNT(0);
}

}  // namespace

// Tile width for the NT layout: fewest "rounds x width" over the CUs (ties -> the wider tile); 0 = not eligible.
static int pick_width(const GemmParams& p, int layout, int cus) {
    if (p.M % TM || p.K % TK || p.k_per_split % TK) return 0;                 // whole tiles only: the kernels floor M / TM and K / TK
    if (p.rot_cos) return (layout == 0 && p.N % 256 == 0) ? 256 : 0;          // the rotary epilogue exists for the 256-wide tile
    if (const char* e = getenv("SCONF_GEMM_256_WIDTH")) {                 // benchmarking override
        const int w = atoi(e);
        return ((w == 256 || (w == 192 && layout == 0)) && p.N % w == 0) ? w : 0;
    }
    long best = 0; int bw = 0;
    for (int w : {256, 192}) {
        if (p.N % w || (w == 192 && layout != 0)) continue;
        const long items = (long)(p.M / TM) * (p.N / w) * p.splits;
        const long cost = ((items + cus - 1) / cus) * w * (w == 192 ? 9 : 8);   // the 192-wide tile runs ~12 % below the 256 one per flop
        if (!bw || cost < best) { best = cost; bw = w; }
    }
    return bw;
}
static int num_cus_cached() {
    static int cus = 0;
    if (!cus) { const int n = sconf_num_cus(); cus = n > 0 ? n : 256; }
    return cus;
}

bool sconf_gemm256_eligible(const GemmParams& p, int layout) {
    if (layout != 0 && layout != 2) return false;
    const bool ks = layout == 2;
    // specialised epilogues: NT {plain, +residual, GELU_DSAVE, MULAUX} (bias / pre-save / f32 output in any of them); TN plain
    if (ks ? (p.act != SCONF_ACT_NONE || p.resid || p.pre)
           : !(p.act == SCONF_ACT_NONE || ((p.act == SCONF_ACT_GELU_DSAVE || p.act == SCONF_ACT_MULAUX) && !p.resid) ||
               (p.act == SCONF_ACT_SMAXBWD && !p.resid && !p.bias && !p.out_f32 && !p.pre && p.splits == 1))) return false;
    // 32-bit per-lane source offsets relative to a half-tile base
    if ((ks ? 64 : 256) * p.lda * 2 >= (1L << 32) || (ks ? 64 : 256) * p.ldb * 2 >= (1L << 32)) return false;
    const int cus = num_cus_cached();
    const int w = pick_width(p, layout, cus);
    if (!w || (p.act == SCONF_ACT_SMAXBWD && w != 256)) return false;
    // one workgroup per CU: below ~3/4 of a round the 128x128 kernel (2 per CU, 4x the tiles) fills the chip better
    return (long)(p.M / TM) * (p.N / w) * p.splits * 4 >= 3L * cus;
}

int sconf_gemm256_width(const GemmParams& p, int layout) { return pick_width(p, layout, num_cus_cached()); }

int sconf_gemm256_launch(const GemmParams& p, int layout, hipStream_t stream) {
    static bool attr_set = false;
    const size_t shmem = 2 * BUF;
    if (!attr_set) {
        const void* fns[] = {(const void*)gemm256_kernel<false, 2>, (const void*)gemm256_kernel<false, 1>, (const void*)gemm256_kernel<true, 2>,
                             (const void*)gemm256_kernel<false, 2, 0>, (const void*)gemm256_kernel<false, 2, 1>, (const void*)gemm256_kernel<false, 2, 2>,
                             (const void*)gemm256_kernel<false, 2, 3>, (const void*)gemm256_kernel<false, 2, 4>, (const void*)gemm256_kernel<false, 2, 5>,
                             (const void*)gemm256_kernel<false, 2, 7>, (const void*)gemm256_kernel<false, 2, 8>, (const void*)gemm256_kernel<false, 2, 9>,
                             (const void*)gemm256_kernel<false, 2, 10>, (const void*)gemm256_kernel<false, 2, 11>, (const void*)gemm256_kernel<false, 2, 12>,
                             (const void*)gemm192_kernel<11>, (const void*)gemm192_kernel<12>, (const void*)gemm192_kernel<-1>, (const void*)gemm192_kernel<0>,
                             (const void*)gemm192_kernel<1>, (const void*)gemm192_kernel<2>, (const void*)gemm192_kernel<3>, (const void*)gemm192_kernel<4>,
                             (const void*)gemm192_kernel<5>, (const void*)gemm192_kernel<9>};
        for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        attr_set = true;
    }
    const int cus = num_cus_cached();
    const int w = pick_width(p, layout, cus);
    const int total = (p.M / TM) * (p.N / w) * p.splits;
    dim3 grid(std::min(total, cus)), block(512);
    // NT: one kernel instantiation per specialised epilogue kind (SCONF_GEMM_ONE_KERNEL: the kernel that switches, A/B; kinds 9 / 10
    // exist only as instantiations)
    int ek = layout == 2 ? -1 : epilogue_kind(p);
    if (ek == 6 || (ek < 9 && getenv("SCONF_GEMM_ONE_KERNEL"))) ek = -1;
    if (ek >= 11 && getenv("SCONF_GEMM_ONE_KERNEL")) ek = -1;
#define L256(EK_) hipLaunchKernelGGL((gemm256_kernel<false, 2, EK_>), grid, block, shmem, stream, p)
#define L192(EK_) hipLaunchKernelGGL(gemm192_kernel<EK_>, grid, block, shmem, stream, p)
    if (layout == 2)   hipLaunchKernelGGL((gemm256_kernel<true, 2>), grid, block, shmem, stream, p);
    else if (w == 256) {
        switch (ek) {
            case 0: L256(0); break; case 1: L256(1); break; case 2: L256(2); break; case 3: L256(3); break; case 4: L256(4); break;
            case 5: L256(5); break; case 7: L256(7); break; case 8: L256(8); break; case 9: L256(9); break; case 10: L256(10); break;
            case 11: L256(11); break; case 12: L256(12); break;
            default: L256(-1);
        }
    } else if (getenv("SCONF_GEMM_192_4PHASE")) hipLaunchKernelGGL((gemm256_kernel<false, 1>), grid, block, shmem, stream, p);   // A/B
    else {
        switch (ek) {
            case 0: L192(0); break; case 1: L192(1); break; case 2: L192(2); break; case 3: L192(3); break; case 4: L192(4); break;
            case 5: L192(5); break; case 9: L192(9); break; case 11: L192(11); break; case 12: L192(12); break;
            default: L192(-1);
        }
    }
#undef L256
#undef L192
    SCONF_LAUNCH_OK("sconf_gemm_bf16 (256-row tile)");
#ifdef SCONF_GEMM_STAMP
    if (getenv("SCONF_GEMM_STAMP_PRINT") && (layout == 2 || w == 256)) {
        (void)hipStreamSynchronize(stream);
        static unsigned long long hb[256 * 8 * 8];
        (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(g_gemm_stamps), sizeof(hb));
        const int nwg = std::min(total, 256);
        const double ktiles = (double)total / nwg * (p.k_per_split / TK);           // K-tiles per workgroup
        static const char* nm[8] = {"vmcnt wait", "barrier 1", "MFMA cluster", "barrier 2", "epilogue", "re-align", "frag reads", "DMA issue"};
        for (int g = 0; g < 2; ++g) {
            double sums[8] = {}, tot = 0;
            for (int wg = 0; wg < nwg; ++wg) for (int wv = 4 * g; wv < 4 * g + 4; ++wv) for (int i = 0; i < 8; ++i) sums[i] += (double)hb[(wg * 8 + wv) * 8 + i];
            for (int i = 0; i < 8; ++i) tot += sums[i];
            fprintf(stderr, "[gemm256 %s M=%d N=%d K=%d stamps] wave row %d: cycles per K-tile %.0f:", layout == 2 ? "TN" : "NT", p.M, p.N, p.K, g, tot / (nwg * 4.0 * ktiles));
            for (int i = 0; i < 8; ++i) fprintf(stderr, "  %s %.0f (%.1f%%)", nm[i], sums[i] / (nwg * 4.0 * ktiles), 100.0 * sums[i] / tot);
            fprintf(stderr, "\n");
        }
    }
#endif
    return 0;
}
