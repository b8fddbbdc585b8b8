// bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
//
// Each operand is either K-contiguous (row-major [rows][K], the nn.Linear forward layout) or
// K-strided (stored [K][rows]); the three combinations used by the SConformerXL path are
//   NT  (A contig, B contig)   y  = x W^T          (fused_dense.py:465-469, attention.py:513,549, decoder.py:24)
//   NN  (A contig, B strided)  dx = dy W           (dgrad; reference fused_dense_cuda.bias_act_linear_dgrad_bgrad)
//   TN  (A strided, B strided) dW = dy^T x         (wgrad; reference fused_dense_cuda.linear_bias_wgrad)
// so no activation or weight is ever transposed in HBM.
//
// Design (MI355X): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  Operands are staged global -> registers -> LDS (issue-early /
// write-late, double-buffered LDS, one barrier per K-tile).  K-contiguous tiles sit in LDS as
// 128-B rows with a 16-B-chunk XOR swizzle (conflict-free ds_read_b128); K-strided tiles sit as
// [k][rows] 256-B rows with a swizzle chosen so that ds_read_b64_tr_b16 (hardware transpose read)
// is conflict-free.  The MFMA is issued with the operands swapped (acc = B·A^T) so that each lane
// owns 4 *consecutive output columns* of one row: bias/residual/aux traffic and the C store are
// 8-16 B per lane.  Block ids are remapped so that consecutive tiles stay on one XCD (shared L2).
#include "common.h"
#include <stdlib.h>
#include <algorithm>

SCONF_API int sconf_num_cus(void);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;          // 16 KiB per operand tile

struct GemmParams {
    const bf16* A; const bf16* B; void* C;
    int M, N, K;
    long lda, ldb, ldc;
    const float* bias;                              // [N] or null
    const float* resid; long ldr;                   // f32 [M][ldr] or null  (out = resid + alpha*val)
    const bf16* aux; long ldaux;                    // bf16 [M][ldaux] for DGELU / DSILU
    bf16* pre; long ldpre;                          // optional pre-activation save (acc + bias)
    float alpha;
    int act;                                        // SconfAct
    int out_f32;                                    // 1: C is float, 0: C is bf16
    long split_stride;                              // split-K: partial-sum slab s lives at C + s * split_stride (f32)
    int k_per_split;                                // multiple of BK
    int splits;
};

__device__ __forceinline__ int swz_strided(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }

// ---- global -> register staging of one 128 x 64 operand tile (4 x 16 B per thread) ----------
template <bool KS>
__device__ __forceinline__ void tile_gload(uint4 (&r)[4], const bf16* __restrict__ P, long ld, int rows,
                                           int row0, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!KS) {
            const int row = c >> 3, kc = c & 7;
            const int gr = row0 + row, gk = k0 + kc * 8;
            if (gr < rows && gk < kend) v = *reinterpret_cast<const uint4*>(P + (long)gr * ld + gk);
        } else {
            const int kr = c >> 4, rc = c & 15;
            const int gk = k0 + kr, gr = row0 + rc * 8;
            if (gk < kend && gr < rows) v = *reinterpret_cast<const uint4*>(P + (long)gk * ld + gr);
        }
        r[i] = v;
    }
}

template <bool KS>
__device__ __forceinline__ void tile_lstore(const uint4 (&r)[4], char* s, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        int off;
        if (!KS) {
            const int row = c >> 3, kc = c & 7;
            off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
        } else {
            const int kr = c >> 4, rc = c & 15;
            off = kr * 256 + ((rc ^ swz_strided(kr)) << 4);
        }
        *reinterpret_cast<uint4*>(s + off) = r[i];
    }
}

// ---- global -> LDS directly (LDS-DMA, global_load_lds_dwordx4): no staging VGPRs, no ds_write ----------------
// The LDS destination of one wave-instruction is wave-uniform base + lane*16 (linear), so the swizzle is applied to
// the per-lane SOURCE address and undone by the same involution on the fragment read (rule "both sides or neither").
// Rows beyond the matrix are clamped to a valid row (their products are never stored); the K range must be whole
// 64-deep tiles (checked on the host) because a DMA cannot zero-fill.
// Per-lane byte offsets of a tile's 4 chunks are loop-invariant over K: computed once per tile (GldsOffs), while the
// K advance lives in the wave-uniform (SGPR) base pointer, so the staging address math costs no VALU per K-step
// (global_load_lds saddr + 32-bit voffset form).
template <bool KS> struct GldsOffs {
    unsigned off[4];
    __device__ __forceinline__ void set(long ld, int rows, int row0, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            if (!KS) {
                const int row = c >> 3, p = c & 7, kc = p ^ ((row >> 1) & 7);
                off[i] = (unsigned)(((long)min(row0 + row, rows - 1) * ld + kc * 8) * 2);
            } else {
                const int kr = c >> 4, p = c & 15, rc = p ^ swz_strided(kr);
                off[i] = (unsigned)(((long)kr * ld + min(row0 + rc * 8, rows - 8)) * 2);
            }
        }
    }
};
template <bool KS>
__device__ __forceinline__ void tile_glds(const bf16* __restrict__ P, long ld, int k0, const GldsOffs<KS>& o, char* s, int tid) {
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const char* base = reinterpret_cast<const char*>(P) + (KS ? (long)k0 * ld * 2 : (long)k0 * 2);   // wave-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr)(base + o.off[i]), (lptr)(s + ((tid & ~63) + 256 * i) * 16), 16, 0, 0);
}

// ---- LDS -> MFMA fragment: 16 rows [rbase, rbase+16) x 32 k of k-step kk --------------------
template <bool KS>
__device__ __forceinline__ bf16x8 frag_read(const char* s, int rbase, int kk, int lane) {
    if (!KS) {
        const int r = rbase + (lane & 15), c = kk * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(s + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
    } else {
        const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
        const int chunk = (rbase >> 3) + (p >> 1);
        const int k0 = kk * 32 + 8 * g + q, k1 = k0 + 4;
        typedef __attribute__((address_space(3))) bf16x4* lds_p;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_p)(s + k0 * 256 + ((chunk ^ swz_strided(k0)) << 4) + (p & 1) * 8));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (lds_p)(s + k1 * 256 + ((chunk ^ swz_strided(k1)) << 4) + (p & 1) * 8));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// ---- epilogue: lane owns row m, columns n..n+3 of each 16x16 tile ---------------------------------------------
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x4 (&acc)[4][4], int m0, int n0, int split, int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.bias) {
                float b[4]; load4(p.bias + n, b);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += b[e];
            }
            if (p.pre) store4(p.pre + (long)m * p.ldpre + n, v);
            if (p.act == SCONF_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = geluf_(v[e]);
            } else if (p.act == SCONF_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = siluf_(v[e]);
            } else if (p.act == SCONF_ACT_DGELU || p.act == SCONF_ACT_DSILU) {
                float a[4]; load4(p.aux + (long)m * p.ldaux + n, a);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (p.act == SCONF_ACT_DGELU ? dgeluf_(a[e]) : dsiluf_(a[e]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
            if (p.resid) {
                float r[4]; load4(p.resid + (long)m * p.ldr + n, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += r[e];
            }
            if (p.out_f32) store4(reinterpret_cast<float*>(p.C) + split * p.split_stride + (long)m * p.ldc + n, v);
            else           store4(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n, v);
        }
    }
}

// Work item v in [0, ntiles * splits) -> (m0, n0, K range).
//  1. XCD-aware bijective remap: workgroups v and v+8 share an XCD (round-robin dispatch), so give each XCD a CONTIGUOUS
//     range of logical ids: the ~64 workgroups resident on one XCD then work on ~64 consecutive logical ids.
//  2. Logical ids are ordered split-major, then in groups of GM row-panels with the row index fastest inside a group,
//     so 64 consecutive ids cover a GM x 8 patch of output tiles of ONE K-split: every A panel and every B panel is
//     fetched from HBM once per patch and then served by that XCD's 4 MiB L2 (the weight matrices of the model,
//     4.7-6.3 MB, do not fit an L2, so a row-major tile order re-streams them for every 128-row panel).
constexpr int GM = 8;
struct WorkItem { int m0, n0, kbeg, kend, split; };
__device__ __forceinline__ WorkItem tile_coords(const GemmParams& p, int v, int total, int ntiles, int tiles_m, int tiles_n) {
    const int qx = total >> 3, rx = total & 7, xcd = v & 7;
    const int lid = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + (v >> 3);
    const int split = lid / ntiles, t = lid - split * ntiles;
    const int per_group = GM * tiles_n;
    const int g = t / per_group, r = t - g * per_group;
    const int first_m = g * GM, gm = min(GM, tiles_m - first_m);
    WorkItem w;
    w.m0 = (first_m + r % gm) * BM; w.n0 = (r / gm) * BN;
    w.split = split;
    w.kbeg = split * p.k_per_split; w.kend = min(p.K, w.kbeg + p.k_per_split);
    return w;
}

template <bool AKS, bool BKS>
__device__ __forceinline__ void tile_mma(const char* sA, const char* sB, f32x4 (&acc)[4][4], int wm, int wn, int lane) {
    // (requesting both k-steps' fragments up front costs 32 more VGPRs and measured 8 % slower on the NN/TN shapes)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = frag_read<AKS>(sA, wm * 64 + i * 16, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_read<BKS>(sB, wn * 64 + j * 16, kk, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
}

// GLDS = true : persistent workgroups (gridDim.x <= 2 per CU) walk output tiles v = blockIdx.x, +gridDim.x, ...; operands
//               stream global -> LDS by LDS-DMA one K-tile ahead, ACROSS tile boundaries, so a tile's epilogue stores
//               overlap the next tile's first loads and there is no per-tile prologue bubble.
// GLDS = false: general path (any K): one tile per workgroup, register-staged (issue-early / write-late).
template <bool AKS, bool BKS, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int ntiles = tiles_m * tiles_n, total = ntiles * p.splits;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int v = blockIdx.x;
    if (v >= total) return;
    WorkItem w = tile_coords(p, v, total, ntiles, tiles_m, tiles_n);

    if (GLDS) {
        GldsOffs<AKS> oa; GldsOffs<BKS> ob;
        oa.set(p.lda, p.M, w.m0, tid); ob.set(p.ldb, p.N, w.n0, tid);
        tile_glds<AKS>(p.A, p.lda, w.kbeg, oa, smem, tid);
        tile_glds<BKS>(p.B, p.ldb, w.kbeg, ob, smem + TILE_BYTES, tid);
        __syncthreads();                                     // hipcc drains the LDS-DMA (vmcnt(0)) before the barrier
        int cur = 0;
        while (true) {
            const int vn = v + gridDim.x;
            WorkItem wn_ = w;
            if (vn < total) wn_ = tile_coords(p, vn, total, ntiles, tiles_m, tiles_n);
            const int nkt = (w.kend - w.kbeg) / BK;
            for (int kt = 0; kt < nkt; ++kt) {
                const char* sA = smem + cur * 2 * TILE_BYTES;
                char* dA = smem + (cur ^ 1) * 2 * TILE_BYTES;
                if (kt + 1 < nkt) {                          // next K-tile streams into the other buffer during the MFMAs
                    tile_glds<AKS>(p.A, p.lda, w.kbeg + (kt + 1) * BK, oa, dA, tid);
                    tile_glds<BKS>(p.B, p.ldb, w.kbeg + (kt + 1) * BK, ob, dA + TILE_BYTES, tid);
                } else if (vn < total) {                     // ... or the NEXT work item's first K-tile
                    oa.set(p.lda, p.M, wn_.m0, tid); ob.set(p.ldb, p.N, wn_.n0, tid);
                    tile_glds<AKS>(p.A, p.lda, wn_.kbeg, oa, dA, tid);
                    tile_glds<BKS>(p.B, p.ldb, wn_.kbeg, ob, dA + TILE_BYTES, tid);
                }
                tile_mma<AKS, BKS>(sA, sA + TILE_BYTES, acc, wm, wn, lane);
                if (kt == nkt - 1) {
                    gemm_epilogue(p, acc, w.m0, w.n0, w.split, wm, wn, lane);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                __syncthreads();
                cur ^= 1;
            }
            if (vn >= total) break;
            v = vn; w = wn_;
        }
    } else {
        const int kbeg = w.kbeg, kend = w.kend, m0 = w.m0, n0 = w.n0;
        const int nkt = (kend - kbeg + BK - 1) / BK;
        uint4 ra[4], rb[4];
        tile_gload<AKS>(ra, p.A, p.lda, p.M, m0, kbeg, kend, tid);
        tile_gload<BKS>(rb, p.B, p.ldb, p.N, n0, kbeg, kend, tid);
        tile_lstore<AKS>(ra, smem, tid);
        tile_lstore<BKS>(rb, smem + TILE_BYTES, tid);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const char* sA = smem + cur * 2 * TILE_BYTES;
            if (kt + 1 < nkt) {                              // issue next tile's HBM loads early
                tile_gload<AKS>(ra, p.A, p.lda, p.M, m0, kbeg + (kt + 1) * BK, kend, tid);
                tile_gload<BKS>(rb, p.B, p.ldb, p.N, n0, kbeg + (kt + 1) * BK, kend, tid);
            }
            tile_mma<AKS, BKS>(sA, sA + TILE_BYTES, acc, wm, wn, lane);
            if (kt + 1 < nkt) {                              // write late, into the other buffer
                char* dA = smem + (cur ^ 1) * 2 * TILE_BYTES;
                tile_lstore<AKS>(ra, dA, tid);
                tile_lstore<BKS>(rb, dA + TILE_BYTES, tid);
            }
            __syncthreads();
        }
        gemm_epilogue(p, acc, m0, n0, w.split, wm, wn, lane);
    }
}

// =================================================================================================================
// Deep-pipelined variant: 256x128x64 block tile, 8 waves (4x2, each 64x64), ONE workgroup per CU, 3-stage LDS ring
// (3 x 48 KiB), LDS-DMA prefetch distance 2 with a COUNTED s_waitcnt vmcnt(6) and a raw s_barrier per K-step, so two
// K-steps of loads stay in flight across every barrier (the 2-stage kernel above drains to vmcnt(0) each step).
// Persistent over (tile, split) work items with the same XCD/L2-patch ordering; the prefetch cursor runs two steps
// ahead of the compute cursor across item boundaries.  Every step issues exactly 6 LDS-DMA instructions per lane
// (A: 2 sub-tiles x 2, B: 2) — past the last step the cursor stays clamped and re-loads into the free ring slot —
// so the wait count is a constant.  Epilogue stores are also VMEM ops: they only make the wait conservative.
// =================================================================================================================
constexpr int BM3 = 256, STAGE3 = 3 * TILE_BYTES;      // A: 2 sub-tiles of 128 rows (32 KiB) | B: 16 KiB

template <bool KS> struct Offs512 {                    // 128-row sub-tile, 512 threads -> 2 chunks per thread
    unsigned off[2];
    __device__ __forceinline__ void set(long ld, int rows, int row0, int tid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 512 * i;
            if (!KS) {
                const int row = c >> 3, p = c & 7, kc = p ^ ((row >> 1) & 7);
                off[i] = (unsigned)(((long)min(row0 + row, rows - 1) * ld + kc * 8) * 2);
            } else {
                const int kr = c >> 4, p = c & 15, rc = p ^ swz_strided(kr);
                off[i] = (unsigned)(((long)kr * ld + min(row0 + rc * 8, rows - 8)) * 2);
            }
        }
    }
};
template <bool KS>
__device__ __forceinline__ void sub_glds(const bf16* __restrict__ P, long ld, int k0, const Offs512<KS>& o, char* s, int tid) {
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const char* base = reinterpret_cast<const char*>(P) + (KS ? (long)k0 * ld * 2 : (long)k0 * 2);   // wave-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_global_load_lds((gptr)(base + o.off[i]), (lptr)(s + ((tid & ~63) + 512 * i) * 16), 16, 0, 0);
}

__device__ __forceinline__ WorkItem tile_coords3(const GemmParams& p, int v, int total, int ntiles, int tiles_m, int tiles_n) {
    const int qx = total >> 3, rx = total & 7, xcd = v & 7;
    const int lid = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + (v >> 3);
    const int split = lid / ntiles, t = lid - split * ntiles;
    constexpr int GM3 = 4;                               // 4 row panels of 256 = the same 1024-row L2 patch as GM=8 x 128
    const int per_group = GM3 * tiles_n;
    const int g = t / per_group, r = t - g * per_group;
    const int first_m = g * GM3, gm = min(GM3, tiles_m - first_m);
    WorkItem w;
    w.m0 = (first_m + r % gm) * BM3; w.n0 = (r / gm) * BN;
    w.split = split;
    w.kbeg = split * p.k_per_split; w.kend = min(p.K, w.kbeg + p.k_per_split);
    return w;
}

template <bool AKS, bool BKS>
__global__ __launch_bounds__(512, 2) void gemm_kernel3(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][A0 16K | A1 16K | B 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;             // wm 0..3
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM3 - 1) / BM3;
    const int ntiles = tiles_m * tiles_n, total = ntiles * p.splits;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int cv = blockIdx.x;                                 // compute cursor
    if (cv >= total) return;
    WorkItem cw = tile_coords3(p, cv, total, ntiles, tiles_m, tiles_n);
    int ckt = 0, cnkt = (cw.kend - cw.kbeg) / BK;

    int pv = cv; WorkItem pw = cw; int pkt = 0, pnkt = cnkt;   // prefetch cursor (2 steps ahead)
    Offs512<AKS> oa0, oa1; Offs512<BKS> ob;
    auto set_offs = [&]() { oa0.set(p.lda, p.M, pw.m0, tid); oa1.set(p.lda, p.M, pw.m0 + 128, tid); ob.set(p.ldb, p.N, pw.n0, tid); };
    auto issue = [&](int slot) {
        char* d = smem + slot * STAGE3;
        const int k0 = pw.kbeg + pkt * BK;
        sub_glds<AKS>(p.A, p.lda, k0, oa0, d, tid);
        sub_glds<AKS>(p.A, p.lda, k0, oa1, d + TILE_BYTES, tid);
        sub_glds<BKS>(p.B, p.ldb, k0, ob, d + 2 * TILE_BYTES, tid);
    };
    auto advance = [&]() {
        if (pkt + 1 < pnkt) { ++pkt; return; }
        const int vn = pv + gridDim.x;
        if (vn < total) { pv = vn; pw = tile_coords3(p, pv, total, ntiles, tiles_m, tiles_n); pkt = 0; pnkt = (pw.kend - pw.kbeg) / BK; set_offs(); }
        // else: stay clamped on the last step (harmless re-load into the free ring slot keeps the wait count constant)
    };
    set_offs();
    issue(0); advance();
    issue(1); advance();

    int slot = 0;
    while (true) {
        // tile for this step has landed for THIS wave when at most the 6 youngest VMEM ops (next step's DMA) are pending;
        // the barrier then makes every wave's share visible and frees the slot that step+2 is about to overwrite.
        asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        int pslot = slot + 2; if (pslot >= 3) pslot -= 3;
        issue(pslot); advance();
        const char* sA = smem + slot * STAGE3 + (wm >> 1) * TILE_BYTES;
        const char* sB = smem + slot * STAGE3 + 2 * TILE_BYTES;
        tile_mma<AKS, BKS>(sA, sB, acc, wm & 1, wn, lane);
        if (++slot == 3) slot = 0;
        if (++ckt == cnkt) {
            gemm_epilogue(p, acc, cw.m0 + (wm >> 1) * 128, cw.n0, cw.split, wm & 1, wn, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int vn = cv + gridDim.x;
            if (vn >= total) break;
            cv = vn; cw = tile_coords3(p, cv, total, ntiles, tiles_m, tiles_n); ckt = 0; cnkt = (cw.kend - cw.kbeg) / BK;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup's LDS allocation
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int splits, long n, int accumulate) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (; i < n; i += stride) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (accumulate) load4(out + i, a);
        for (int s = 0; s < splits; ++s) {
            float v[4]; load4(slab + (long)s * n + i, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
        store4(out + i, a);
    }
}

}  // namespace

// C ABI -----------------------------------------------------------------------------------------
// layout: 0 = NT (A[M][K], B[N][K]); 1 = NN (A[M][K], B[K][N]); 2 = TN (A[K][M], B[K][N]).
// Replaces: F.linear / fused_dense_cuda.linear_act_forward (fused_dense.py:277-279,329-332),
// bias_act_linear_dgrad_bgrad (:354-356), linear_bias_wgrad (:113-115,338-340,375-378).
SCONF_API int sconf_gemm_bf16(int layout, const void* A, const void* B, void* C,
                              int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                              const float* bias, const float* resid, int64_t ldr,
                              const void* aux, int64_t ldaux, void* pre, int64_t ldpre,
                              float alpha, int act, int out_f32, int split_k, hipStream_t stream) {
    SCONF_REQUIRE(layout >= 0 && layout <= 2, "sconf_gemm_bf16: bad layout %d", layout);
    SCONF_REQUIRE(M > 0 && N > 0 && K > 0, "sconf_gemm_bf16: empty problem %ld x %ld x %ld", (long)M, (long)N, (long)K);
    SCONF_REQUIRE(M < (1L << 31) && N < (1L << 31) && K < (1L << 31), "sconf_gemm_bf16: dims must be < 2^31");
    SCONF_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "sconf_gemm_bf16: N and ldc must be multiples of 4 (N=%ld ldc=%ld)", (long)N, (long)ldc);
    const bool aks = layout == 2, bks = layout >= 1;
    // 16-byte global loads: contiguous dim must be a multiple of 8 elements
    SCONF_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "sconf_gemm_bf16: lda/ldb must be multiples of 8");
    SCONF_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 7) == 0, "sconf_gemm_bf16: misaligned operand");
    if (!aks) SCONF_REQUIRE(K % 8 == 0, "sconf_gemm_bf16: K must be a multiple of 8 for K-contiguous A");
    else      SCONF_REQUIRE(M % 8 == 0, "sconf_gemm_bf16: M must be a multiple of 8 for K-strided A");
    if (!bks) SCONF_REQUIRE(K % 8 == 0, "sconf_gemm_bf16: K must be a multiple of 8 for K-contiguous B");
    else      SCONF_REQUIRE(N % 8 == 0, "sconf_gemm_bf16: N must be a multiple of 8 for K-strided B");
    SCONF_REQUIRE(split_k >= 1, "sconf_gemm_bf16: split_k must be >= 1");
    SCONF_REQUIRE(split_k == 1 || out_f32, "sconf_gemm_bf16: split-K writes f32 partial slabs and needs out_f32");
    if (act == SCONF_ACT_DGELU || act == SCONF_ACT_DSILU) SCONF_REQUIRE(aux != nullptr, "sconf_gemm_bf16: dact epilogue needs aux");

    GemmParams p;
    p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.resid = resid; p.ldr = ldr; p.aux = (const bf16*)aux; p.ldaux = ldaux;
    p.pre = (bf16*)pre; p.ldpre = ldpre; p.alpha = alpha; p.act = act; p.out_f32 = out_f32;
    const int nkt = cdiv(K, BK);
    p.k_per_split = cdiv(nkt, split_k) * BK;
    const int splits = cdiv(K, p.k_per_split);
    p.splits = splits;
    p.split_stride = splits > 1 ? M * ldc : 0;
    if (splits > 1) SCONF_REQUIRE(!bias && !resid && act == SCONF_ACT_NONE && !pre, "sconf_gemm_bf16: split-K supports only the plain epilogue");

    const int ntiles = cdiv(M, BM) * cdiv(N, BN);
    dim3 grid(ntiles * splits), block(256);
    const size_t shmem = 4 * TILE_BYTES;
    static bool attr_set = false;
    static bool no_glds = false;
    if (!attr_set) {
        const void* fns[6] = {(const void*)gemm_kernel<false, false, false>, (const void*)gemm_kernel<false, true, false>,
                              (const void*)gemm_kernel<true, true, false>, (const void*)gemm_kernel<false, false, true>,
                              (const void*)gemm_kernel<false, true, true>, (const void*)gemm_kernel<true, true, true>};
        for (int i = 0; i < 6; ++i) (void)hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        no_glds = getenv("SCONF_GEMM_NO_GLDS") != nullptr;      // A/B switch for benchmarking the two staging paths
        attr_set = true;
    }
    // LDS-DMA staging needs whole 64-deep K tiles (no zero fill), at least one full 8-row chunk to clamp to, and
    // operands addressable with 32-bit byte offsets from a uniform base.
    const bool glds = !no_glds && K % BK == 0 && M >= 8 && N >= 8 &&
                      (long)(aks ? K : M) * lda * 2 < (1L << 32) && (long)(bks ? K : N) * ldb * 2 < (1L << 32);
    // The deep-pipelined 256x128 kernel measured no better than the 2-stage 128x128 one on this workload (it wins
    // ~9 % on long-K NT shapes, loses 10-20 % on the NN/TN shapes), i.e. the 2-stage kernel is not latency-bound:
    // it stays opt-in for experiments.
    static bool use_v3 = getenv("SCONF_GEMM_V3") != nullptr;
    if (glds && use_v3 && M >= 256) {
        static bool a3 = false;
        if (!a3) {
            (void)hipFuncSetAttribute((const void*)gemm_kernel3<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE3);
            (void)hipFuncSetAttribute((const void*)gemm_kernel3<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE3);
            (void)hipFuncSetAttribute((const void*)gemm_kernel3<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE3);
            a3 = true;
        }
        static int cus = 0;
        if (!cus) { int n = sconf_num_cus(); cus = n > 0 ? n : 256; }
        const int nt3 = cdiv(M, BM3) * cdiv(N, BN);
        dim3 g3(std::min(nt3 * splits, cus)), b3(512);
        if (layout == 0)      hipLaunchKernelGGL((gemm_kernel3<false, false>), g3, b3, 3 * STAGE3, stream, p);
        else if (layout == 1) hipLaunchKernelGGL((gemm_kernel3<false, true>), g3, b3, 3 * STAGE3, stream, p);
        else                  hipLaunchKernelGGL((gemm_kernel3<true, true>), g3, b3, 3 * STAGE3, stream, p);
    } else if (glds) {
        static int slots = 0;
        if (!slots) { int n = sconf_num_cus(); slots = 2 * (n > 0 ? n : 256); }
        grid.x = std::min(ntiles * splits, slots);                // persistent: <= 2 resident workgroups per CU
        if (layout == 0)      hipLaunchKernelGGL((gemm_kernel<false, false, true>), grid, block, shmem, stream, p);
        else if (layout == 1) hipLaunchKernelGGL((gemm_kernel<false, true, true>), grid, block, shmem, stream, p);
        else                  hipLaunchKernelGGL((gemm_kernel<true, true, true>), grid, block, shmem, stream, p);
    } else {
        if (layout == 0)      hipLaunchKernelGGL((gemm_kernel<false, false, false>), grid, block, shmem, stream, p);
        else if (layout == 1) hipLaunchKernelGGL((gemm_kernel<false, true, false>), grid, block, shmem, stream, p);
        else                  hipLaunchKernelGGL((gemm_kernel<true, true, false>), grid, block, shmem, stream, p);
    }
    SCONF_LAUNCH_OK("sconf_gemm_bf16");
    return 0;
}

// Number of K-splits sconf_gemm_bf16 will actually use for (K, split_k): the caller sizes the slab buffer with it.
SCONF_API int sconf_gemm_num_splits(int64_t K, int split_k) {
    const int nkt = cdiv(K, BK);
    const int kps = cdiv(nkt, split_k < 1 ? 1 : split_k) * BK;
    return cdiv(K, kps);
}

// out[i] (+)= sum_s slab[s][i], i < n (n % 4 == 0): deterministic split-K combine (fixed summation order).
SCONF_API int sconf_splitk_reduce(const float* slab, float* out, int64_t splits, int64_t n, int accumulate, hipStream_t stream) {
    SCONF_REQUIRE(n % 4 == 0, "sconf_splitk_reduce: n must be a multiple of 4");
    if (n == 0) return 0;
    const int blocks = (int)std::min<long>(cdiv(n, 1024), 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, slab, out, (int)splits, (long)n, accumulate);
    SCONF_LAUNCH_OK("sconf_splitk_reduce");
    return 0;
}
