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
#include "gemm_tile.h"
#include <stdlib.h>
#include <algorithm>

SCONF_API int sconf_num_cus(void);
#ifdef SCONF_GEMM_PROBE
static long long* g_probe_stamps = nullptr;
// probe builds only: device buffer of 256 x 64 x 4 int64 that the 256x256 kernel fills with per-item time stamps (or null)
SCONF_API int sconf_gemm_probe_stamps(void* buf) { g_probe_stamps = (long long*)buf; return 0; }
#endif

namespace {
using namespace gemm_tile;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;          // 16 KiB per operand tile

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

template <bool KS, bool BIMG>
__device__ __forceinline__ void tile_lstore(const uint4 (&r)[4], char* s, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        int off;
        if (!KS) {
            const int row = c >> 3, kc = c & 7;
            off = row * 128 + ((kc ^ swz_kc<BIMG>(row)) << 4);
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
template <bool KS, bool BIMG> struct GldsOffs {
    unsigned off[4];
    __device__ __forceinline__ void set(long ld, int rows, int row0, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            if (!KS) {
                const int row = c >> 3, p = c & 7, kc = p ^ swz_kc<BIMG>(row);
                off[i] = (unsigned)(((long)min(row0 + row, rows - 1) * ld + kc * 8) * 2);
            } else {
                const int kr = c >> 4, p = c & 15, rc = p ^ swz_strided(kr);
                off[i] = (unsigned)(((long)kr * ld + min(row0 + rc * 8, rows - 8)) * 2);
            }
        }
    }
};
template <bool KS, bool BIMG>
__device__ __forceinline__ void tile_glds(const bf16* __restrict__ P, long ld, int k0, const GldsOffs<KS, BIMG>& o, char* s, int tid) {
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    const char* base = reinterpret_cast<const char*>(P) + (KS ? (long)k0 * ld * 2 : (long)k0 * 2);   // wave-uniform
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr)(base + o.off[i]), (lptr)(s + ((tid & ~63) + 256 * i) * 16), 16, 0, 0);
}

// ---- LDS -> MFMA fragment, 16 rows x 32 k of k-step kk ------------------------------------------------------------
// A operand (BIMG = false): rows rbase + (0..15).
// B operand, K-contiguous (BIMG = true, NT layout): PERMUTED rows  wbase + 16*(r>>2) + 4*j + (r&3),  r = 0..15, for
//   column tile j of the wave's 64 columns.  (A K-strided B keeps natural columns: the permuted transposed read would be
//   2-way bank-conflicted, measured -9 % on the wgrad shapes; those kernels use the narrow epilogue.)  With the MFMA issued operand-swapped, lane (r = lane&15 -> output row m, g = lane>>4) then accumulates
//   columns wbase + 16*g + 4*j + (0..3) in tile j: over j = 0..3 that is 16 CONSECUTIVE output columns per lane, so the
//   epilogue moves 32 B (bf16) / 64 B (f32) per lane and 128-256 B per row per wave instead of 8-B pieces.
template <bool KS, bool BIMG>
__device__ __forceinline__ bf16x8 frag_read(const char* s, int rbase, int j, int kk, int lane) {
    if (!KS) {
        const int rl = lane & 15;
        const int r = BIMG ? rbase + ((rl >> 2) << 4) + 4 * j + (rl & 3) : rbase + 16 * j + rl;
        const int c = kk * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(s + r * 128 + ((c ^ swz_kc<BIMG>(r)) << 4));
    } else {
        const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
        // 16-B chunk and 8-B half of the 4 consecutive rows this lane addresses for the transposed read
        const int col = rbase + 16 * j + 4 * p;               // strided operands keep the natural column order (see epilogue)
        const int chunk = col >> 3, sub = (col & 4) * 2;
        const int k0 = kk * 32 + 8 * g + q, k1 = k0 + 4;
        typedef __attribute__((address_space(3))) bf16x4* lds_p;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(s + k0 * 256 + ((chunk ^ swz_strided(k0)) << 4) + sub));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(s + k1 * 256 + ((chunk ^ swz_strided(k1)) << 4) + sub));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// ---- epilogues of the 64x64 wave tile: four 16-row blocks, each handled by the shared row-block routines ----------------
__device__ __forceinline__ void gemm_epilogue_narrow(const GemmParams& p, const f32x4 (&acc)[4][4], int m0, int n0, int split, int wm, int wn, int lane) {
    const int nb = n0 + wn * 64 + (lane >> 4) * 4;
    const int ncol[4] = {nb, nb + 16, nb + 32, nb + 48};
#pragma unroll
    for (int i = 0; i < 4; ++i) epi_narrow_row(p, acc[i], m0 + wm * 64 + i * 16 + (lane & 15), ncol, split);
}
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x4 (&acc)[4][4], int m0, int n0, int split, int wm, int wn, int lane) {
    const int nrun = n0 + wn * 64 + (lane >> 4) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) epi_wide_row(p, acc[i], m0 + wm * 64 + i * 16 + (lane & 15), nrun, split);
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
        for (int i = 0; i < 4; ++i) af[i] = frag_read<AKS, false>(sA, wm * 64, i, kk, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = frag_read<BKS, true>(sB, wn * 64, j, kk, lane);
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
        GldsOffs<AKS, false> oa; GldsOffs<BKS, true> ob;
        oa.set(p.lda, p.M, w.m0, tid); ob.set(p.ldb, p.N, w.n0, tid);
        tile_glds<AKS, false>(p.A, p.lda, w.kbeg, oa, smem, tid);
        tile_glds<BKS, true>(p.B, p.ldb, w.kbeg, ob, smem + TILE_BYTES, tid);
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
                    tile_glds<AKS, false>(p.A, p.lda, w.kbeg + (kt + 1) * BK, oa, dA, tid);
                    tile_glds<BKS, true>(p.B, p.ldb, w.kbeg + (kt + 1) * BK, ob, dA + TILE_BYTES, tid);
                } else if (vn < total) {                     // ... or the NEXT work item's first K-tile
                    oa.set(p.lda, p.M, wn_.m0, tid); ob.set(p.ldb, p.N, wn_.n0, tid);
                    tile_glds<AKS, false>(p.A, p.lda, wn_.kbeg, oa, dA, tid);
                    tile_glds<BKS, true>(p.B, p.ldb, wn_.kbeg, ob, dA + TILE_BYTES, tid);
                }
                tile_mma<AKS, BKS>(sA, sA + TILE_BYTES, acc, wm, wn, lane);
                if (kt == nkt - 1) {
                    if (BKS) gemm_epilogue_narrow(p, acc, w.m0, w.n0, w.split, wm, wn, lane);
                    else     gemm_epilogue(p, acc, w.m0, w.n0, w.split, wm, wn, lane);
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
        tile_lstore<AKS, false>(ra, smem, tid);
        tile_lstore<BKS, true>(rb, smem + TILE_BYTES, tid);
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
                tile_lstore<AKS, false>(ra, dA, tid);
                tile_lstore<BKS, true>(rb, dA + TILE_BYTES, tid);
            }
            __syncthreads();
        }
        if (BKS) gemm_epilogue_narrow(p, acc, m0, n0, w.split, wm, wn, lane);
        else     gemm_epilogue(p, acc, m0, n0, w.split, wm, wn, lane);
    }
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
namespace { struct RotSpec { const float* cos = nullptr; const float* sin = nullptr; int n = 0, cols = 0; }; thread_local RotSpec g_rot;
            struct SmbSpec { const float* rowv = nullptr; float* colslab = nullptr; }; thread_local SmbSpec g_smb; }

// Replaces: F.linear / fused_dense_cuda.linear_act_forward (fused_dense.py:277-279,329-332),
// bias_act_linear_dgrad_bgrad (:354-356), linear_bias_wgrad (:113-115,338-340,375-378).
SCONF_API int sconf_gemm_bf16(int layout, const void* A, const void* B, void* C,
                              int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                              const float* bias, const float* resid, int64_t ldr,
                              const void* aux, int64_t ldaux, void* pre, int64_t ldpre,
                              float alpha, int act, int out_f32, int split_k, hipStream_t stream) {
    SCONF_REQUIRE(layout >= 0 && layout <= 2, "sconf_gemm_bf16: bad layout %d", layout);
    SCONF_REQUIRE(M >= 0 && N >= 0 && K > 0, "sconf_gemm_bf16: bad problem %ld x %ld x %ld (K must be positive)", (long)M, (long)N, (long)K);
    if (M == 0 || N == 0) return 0;                    // an empty output (e.g. a batch with no rows): nothing to compute
    SCONF_REQUIRE(M < (1L << 31) && N < (1L << 31) && K < (1L << 31), "sconf_gemm_bf16: dims must be < 2^31");
    SCONF_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "sconf_gemm_bf16: N and ldc must be multiples of 4 (N=%ld ldc=%ld)", (long)N, (long)ldc);
    if (layout == 0) SCONF_REQUIRE(N % 16 == 0 && ldc % 8 == 0, "sconf_gemm_bf16: the NT layout needs N %% 16 == 0 and ldc %% 8 == 0 (N=%ld ldc=%ld)", (long)N, (long)ldc);
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
    if (act == SCONF_ACT_DGELU || act == SCONF_ACT_DSILU || act == SCONF_ACT_MULAUX) SCONF_REQUIRE(aux != nullptr, "sconf_gemm_bf16: aux epilogue needs aux");
    if (act == SCONF_ACT_GELU_DSAVE) SCONF_REQUIRE(pre != nullptr, "sconf_gemm_bf16: GELU_DSAVE needs the pre buffer");
    SCONF_REQUIRE(act >= 0 && (act <= 6 || (act == SCONF_ACT_SMAXBWD && g_smb.rowv)), "sconf_gemm_bf16: bad act %d", act);

    GemmParams p;
    p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.resid = resid; p.ldr = ldr; p.aux = (const bf16*)aux; p.ldaux = ldaux;
    p.pre = (bf16*)pre; p.ldpre = ldpre; p.alpha = alpha; p.act = act; p.out_f32 = out_f32;
#ifdef SCONF_GEMM_PROBE
    { const char* d = getenv("SCONF_GEMM_DEBUG"); p.debug = d ? atoi(d) : 0; }
    { const char* e = getenv("SCONF_GEMM_STAGGER"); p.stagger = e ? atoi(e) : 0; const char* f = getenv("SCONF_GEMM_STAGGER_MODE"); p.stagger_mode = f ? atoi(f) : 0; }
    p.stamps = g_probe_stamps;
#endif
    { const char* e = getenv("SCONF_GEMM_GM"); p.gm = e ? atoi(e) : 0; }                  // tuning: L2 patch height of the 256-row kernels
    p.rot_cos = g_rot.cos; p.rot_sin = g_rot.sin; p.rot_n = g_rot.n; p.rot_cols = g_rot.cols;   // set only inside sconf_gemm_qkv_rotary
    p.rowv = g_smb.rowv; p.colslab = g_smb.colslab;                                              // set only inside sconf_gemm_softmax_bwd
    const int nkt = cdiv(K, BK);
    p.k_per_split = cdiv(nkt, split_k) * BK;
    const int splits = cdiv(K, p.k_per_split);
    p.splits = splits;
    p.split_stride = splits > 1 ? M * ldc : 0;
    if (splits > 1) SCONF_REQUIRE(!bias && !resid && act == SCONF_ACT_NONE && !pre, "sconf_gemm_bf16: split-K supports only the plain epilogue");

    if (p.rot_cos || act == SCONF_ACT_SMAXBWD) {      // only the 256-row NT kernels have these epilogues: the caller falls back otherwise
        if (getenv("SCONF_GEMM_NO_256") || !sconf_gemm256_eligible(p, layout)) return 2;
        return sconf_gemm256_launch(p, layout, stream);
    }
    // A/B switch (read per call so that one process can compare both kernels): keep everything on the 128x128 kernel
    if (!getenv("SCONF_GEMM_NO_256") && sconf_gemm256_eligible(p, layout)) return sconf_gemm256_launch(p, layout, stream);

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
    if (glds) {
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

SCONF_API int sconf_rotary_inplace(void* qkv, const float* cos_tab, const float* sin_tab, int64_t B, int64_t N, int64_t H, int64_t D, hipStream_t stream);

// qkv projection with the rotary rotation in the GEMM epilogue: C (M, 3, H, D) bf16 = [q | k | v] = x W^T (+ bias) with W in the
// REGROUPED row order (sconf_cast_shadows, R < 0) and q, k rotated by the NeoX rotary of position (row % seq_len) - attention.py:485,
// 498-507 + rotary_emb.py:61-73 with no pass over the activation.  cos / sin: (seq_len, D/2) f32.  head_dim 128 through the 256x256
// kernel; any other shape runs the plain GEMM followed by sconf_rotary_inplace (same result up to one bf16 rounding).
SCONF_API int sconf_gemm_qkv_rotary(const void* A, const void* W, void* C, int64_t M, int64_t K, int64_t H, int64_t D,
                                    int64_t lda, int64_t ldb, const float* bias, const float* cos_tab, const float* sin_tab,
                                    int64_t seq_len, hipStream_t stream) {
    SCONF_REQUIRE(cos_tab && sin_tab && seq_len > 0 && M % seq_len == 0, "sconf_gemm_qkv_rotary: tables / seq_len (M=%ld seq_len=%ld)", (long)M, (long)seq_len);
    const int64_t N = 3 * H * D;
    int rc = 2;
    if (D == 128 && !getenv("SCONF_QKV_ROT_EPILOGUE_OFF")) {
        g_rot.cos = cos_tab; g_rot.sin = sin_tab; g_rot.n = (int)seq_len; g_rot.cols = (int)(2 * H * D);
        rc = sconf_gemm_bf16(0, A, W, C, M, N, K, lda, ldb, N, bias, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, SCONF_ACT_NONE, 0, 1, stream);
        g_rot = RotSpec();
    }
    if (rc != 2) return rc;
    rc = sconf_gemm_bf16(0, A, W, C, M, N, K, lda, ldb, N, bias, nullptr, 0, nullptr, 0, nullptr, 0, 1.f, SCONF_ACT_NONE, 0, 1, stream);
    if (rc) return rc;
    return sconf_rotary_inplace(C, cos_tab, sin_tab, M / seq_len, seq_len, H, D, stream);
}

// Softmax backward inside the GEMM that produces its input gradient (the self-conditioning reprojection's dgrad, sconformer_xl.py:241-243
// backward): dl[m][v] = probs[m][v] * ((dy Wt^T)[m][v] - delta[m]) in bf16, where delta[m] = sum_v probs dp is handed in (it equals
// sum_c dy[m][c] * (reprojection output before bias / residual)[m][c]: sconf_rowdot on the forward's saved product) - dp = dy Wt^T is
// never written.  colslab (2 * M / 256, V) f32 receives per-(row panel, wave row) column sums of dl (the decoder bias gradient: add
// them up with sconf_colsum).  Returns 2 when the shape does not take the 256-row NT kernel (the caller then runs the plain GEMM +
// sconf_softmax_bwd); otherwise 0 / an error.
SCONF_API int sconf_gemm_softmax_bwd(const void* dy, const void* Wt, const void* probs, const float* delta, void* dl, float* colslab,
                                     int64_t M, int64_t V, int64_t K, int64_t lddy, int64_t ldw, int64_t ldp, hipStream_t stream) {
    SCONF_REQUIRE(delta && colslab && probs, "sconf_gemm_softmax_bwd: delta, colslab and probs are required");
    g_smb.rowv = delta; g_smb.colslab = colslab;
    const int rc = sconf_gemm_bf16(0, dy, Wt, dl, M, V, K, lddy, ldw, V, nullptr, nullptr, 0, probs, ldp, nullptr, 0, 1.f, SCONF_ACT_SMAXBWD, 0, 1, stream);
    g_smb = SmbSpec();
    return rc;
}

// Which kernel sconf_gemm_bf16 runs for a problem (diagnostics / benchmark bookkeeping; same decision code as the launch):
// 0 = gemm_kernel (128x128 tile, 4 waves), 1 = gemm256_kernel<NT, 256 wide>, 2 = gemm256_kernel<NT, 192 wide>,
// 3 = gemm256_kernel<TN>.  (2 is gemm192_kernel unless SCONF_GEMM_192_4PHASE asks for the 4-phase template.)
SCONF_API int sconf_gemm_variant(int layout, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int split_k, int act,
                                 int has_resid, int has_pre) {
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0 || split_k < 1) return -1;
    GemmParams p{};
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.lda = lda; p.ldb = ldb; p.act = act;
    p.resid = has_resid ? reinterpret_cast<const float*>(8) : nullptr;      // only tested for null
    p.pre = has_pre ? reinterpret_cast<bf16*>(8) : nullptr;
    p.k_per_split = cdiv(cdiv(K, BK), split_k) * BK;
    p.splits = cdiv(K, p.k_per_split);
    if (getenv("SCONF_GEMM_NO_256") || !sconf_gemm256_eligible(p, layout)) return 0;
    return layout == 2 ? 3 : (sconf_gemm256_width(p, layout) == 256 ? 1 : 2);
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
