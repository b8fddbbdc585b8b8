// Pieces shared by the bf16 GEMM kernels (gemm.hip: 128x128 tile / 4 waves; gemm256.hip: 256x256 tile / 8 waves):
// the launch parameter block, the LDS swizzles and the fused epilogues, written per 16-row block of one wave.
#pragma once
#include "common.h"

namespace gemm_tile {

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
    int gm;                                         // row panels per L2 patch of the 256-row kernels (0 = default)
    // NeoX rotary fused into the epilogue of the qkv projection (sconf_gemm_qkv_rotary; 256x256 NT kernel, head_dim 128): output
    // columns < rot_cols are (head, d) with d < 128; row r is position r % rot_n; tables (rot_n, 64) f32.  null = no rotation.
    const float* rot_cos; const float* rot_sin; int rot_n, rot_cols;
    // softmax backward in the epilogue (sconf_gemm_softmax_bwd; 256-row NT kernel): out = (acc - rowv[m]) * aux[m][n], and the column
    // sums of the output of every (256-row item, wave row) go to colslab[2 * m0 / 256 + wave row][N] (f32; the bias gradient).
    const float* rowv; float* colslab;
#ifdef SCONF_GEMM_PROBE
    int debug;                                      // probe builds only (make PROBE=1; SCONF_GEMM_DEBUG): 1 = skip epilogue stores, 2 = skip the epilogue
    long long* stamps;                              // probe: per workgroup [64 items][4] {realtime at epilogue start, cycles at start, at end, after the next item's first wait}
    int stagger, stagger_mode;                      // probe: start-up phase offsets between workgroups (units of 1024 cycles)
#endif
};

__device__ __forceinline__ int swz_strided(int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); }
// XOR swizzle (in 16-B chunks) of a K-contiguous tile row.  The A image is read 16 consecutive rows at a time; the B
// image is read with the PERMUTED row set {16p + 4j + e} (see tile_mma), so it needs a different conflict-free function.
template <bool BIMG> __device__ __forceinline__ int swz_kc(int row) {
    return BIMG ? (((row >> 1) & 1) | (((row >> 4) & 3) << 1)) : ((row >> 1) & 7);
}


// ---- fused epilogue, per run of W (4 or 8) consecutive output columns of one row ---------------------------------------------
//   out = resid + alpha * act(acc + bias)        (act may also save gelu' / the pre-activation, or multiply by aux)
// Split in two steps so that a caller can issue the aux / residual loads of all its runs before any math or store.
template <int W> __device__ __forceinline__ void loadv(const bf16* p, float (&v)[W]) { if constexpr (W == 8) load8(p, v); else load4(p, v); }
template <int W> __device__ __forceinline__ void loadv(const float* p, float (&v)[W]) { if constexpr (W == 8) load8(p, v); else load4(p, v); }
template <int W> __device__ __forceinline__ void storev(bf16* p, const float (&v)[W]) { if constexpr (W == 8) store8(p, v); else store4(p, v); }
template <int W> __device__ __forceinline__ void storev(float* p, const float (&v)[W]) { if constexpr (W == 8) store8(p, v); else store4(p, v); }

template <int W> struct EpiIn { float ax[W], rs[W]; };
__device__ __forceinline__ bool epi_uses_aux(const GemmParams& p) {
    return p.act == SCONF_ACT_DGELU || p.act == SCONF_ACT_DSILU || p.act == SCONF_ACT_MULAUX || p.act == SCONF_ACT_SMAXBWD;
}
template <int W> __device__ __forceinline__ void epi_load(const GemmParams& p, EpiIn<W>& in, int m, int n) {
#pragma unroll
    for (int e = 0; e < W; ++e) { in.ax[e] = 0.f; in.rs[e] = 0.f; }
    if (epi_uses_aux(p)) loadv<W>(p.aux + (long)m * p.ldaux + n, in.ax);
    if (p.resid) loadv<W>(p.resid + (long)m * p.ldr + n, in.rs);
}
// The math + stores of one run.  ACT >= 0 fixes the activation at compile time (the 256-row kernel dispatches once per
// work item, so that the instructions an item executes are contiguous); ACT = -1 reads p.act.  bs / ax / rs are the run's
// bias, aux and residual values (zeros where absent), loaded by the caller - ahead of any store where it matters: on gfx950
// loads and stores retire through one in-order counter, so a load issued behind a store waits for that store's completion.
// FL >= 0 also fixes, at compile time, whether a bias is added (bit 0), whether the output is f32 (bit 1) and whether the final
// alpha * v + residual step is the identity (bit 2: alpha == 1, no residual), whether acc + bias is saved to p.pre (bit 3), and
// promises that p.pre is otherwise only used by GELU_DSAVE: with those known the compiler emits no branch and - the point - no conservative
// `s_waitcnt vmcnt(0)` in front of each bias use (which drained the stores of the previous row every time).  coff / poff are
// the element offsets of the run in C and in pre (row offset computed once per row block by the caller).
template <int W, int ACT = -1, int FL = -1>
__device__ __forceinline__ void epi_math_store_at(const GemmParams& p, float (&v)[W], const float (&bs)[W], const float (&ax)[W],
                                                  const float (&rs)[W], long coff, long poff, int split) {
    const int act = ACT >= 0 ? ACT : p.act;
    const bool has_bias = FL >= 0 ? (FL & 1) != 0 : p.bias != nullptr;
    const bool f32o = FL >= 0 ? (FL & 2) != 0 : p.out_f32 != 0;
    const bool has_pre = FL >= 0 ? (FL & 8) != 0 : p.pre != nullptr;
    if (has_bias) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] += bs[e];
    }
#ifdef SCONF_GEMM_PROBE
    const bool st = p.debug != 1 || v[0] == 1.2345e-30f;
#else
    constexpr bool st = true;
#endif
    if (act == SCONF_ACT_GELU_DSAVE) {
        float dg[W];
#pragma unroll
        for (int e = 0; e < W; ++e) gelu_both(v[e], v[e], dg[e]);
        if (st) storev<W>(p.pre + poff, dg);
    } else if (has_pre) { if (st) storev<W>(p.pre + poff, v); }
    if (act == SCONF_ACT_MULAUX) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] *= ax[e];
    } else if (act == SCONF_ACT_SMAXBWD) {          // rs carries the row's scalar (the caller sets FL bit 2: no alpha / residual step)
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = (v[e] - rs[e]) * ax[e];
    } else if (act == SCONF_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = geluf_(v[e]);
    } else if (act == SCONF_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = siluf_(v[e]);
    } else if (act == SCONF_ACT_DGELU) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] *= dgeluf_(ax[e]);
    } else if (act == SCONF_ACT_DSILU) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] *= dsiluf_(ax[e]);
    }
    if (!(FL >= 0 && (FL & 4))) {                    // FL bit 2: alpha == 1 and no residual (epilogue_kind checks) - nothing to do
#pragma unroll
        for (int e = 0; e < W; ++e) v[e] = v[e] * p.alpha + rs[e];
    }
    if (!st) return;
    if (f32o) storev<W>(reinterpret_cast<float*>(p.C) + split * p.split_stride + coff, v);
    else      storev<W>(reinterpret_cast<bf16*>(p.C) + coff, v);
}
// Store-only step of a two-phase epilogue: v already holds resid + alpha * act(acc + bias).
template <int W, int FL>
__device__ __forceinline__ void epi_store_at(const GemmParams& p, const float (&v)[W], long coff, int split) {
#ifdef SCONF_GEMM_PROBE
    if (p.debug == 1 && v[0] != 1.2345e-30f) return;
#endif
    if ((FL & 2) != 0) storev<W>(reinterpret_cast<float*>(p.C) + split * p.split_stride + coff, v);
    else               storev<W>(reinterpret_cast<bf16*>(p.C) + coff, v);
}
template <int W, int ACT = -1>
__device__ __forceinline__ void epi_math_store(const GemmParams& p, float (&v)[W], const float (&bs)[W], const float (&ax)[W],
                                               const float (&rs)[W], int m, int n, int split) {
    epi_math_store_at<W, ACT, -1>(p, v, bs, ax, rs, (long)m * p.ldc + n, (long)m * p.ldpre + n, split);
}
template <int W> __device__ __forceinline__ void epi_apply(const GemmParams& p, float (&v)[W], const EpiIn<W>& in, int m, int n, int split) {
    float bs[W];
#pragma unroll
    for (int e = 0; e < W; ++e) bs[e] = 0.f;
    if (p.bias) loadv<W>(p.bias + n, bs);
    epi_math_store<W, -1>(p, v, bs, in.ax, in.rs, m, n, split);
}

// narrow row block (K-strided B: NN / TN): lane (r = lane&15, g = lane>>4) owns row m and, in each of four 16-column tiles
// j, the 4 consecutive columns ncol[j] .. +3 (ncol[j] already includes the lane's 4*g)
__device__ __forceinline__ void epi_narrow_row(const GemmParams& p, const f32x4 (&acc)[4], int m, const int (&ncol)[4], int split) {
    if (m >= p.M) return;
    EpiIn<4> in[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (ncol[j] < p.N) epi_load<4>(p, in[j], m, ncol[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (ncol[j] >= p.N) continue;
        float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        epi_apply<4>(p, v, in[j], m, ncol[j], split);
    }
}
// wide row block (NT): the lane owns row m and the 16 consecutive columns nrun .. nrun+15 (acc[j][e] = column 4j + e),
// handled as two 8-column runs (16 B of bf16 / 32 B of f32 per lane each)
__device__ __forceinline__ void epi_wide_row(const GemmParams& p, const f32x4 (&acc)[4], int m, int nrun, int split) {
    if (m >= p.M || nrun >= p.N) return;                  // N % 16 == 0 (host-checked): the run is all-in or all-out
    EpiIn<8> in[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) epi_load<8>(p, in[hh], m, nrun + 8 * hh);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[2 * hh + (e >> 2)][e & 3];
        epi_apply<8>(p, v, in[hh], m, nrun + 8 * hh, split);
    }
}

}  // namespace gemm_tile

// 256x256-tile kernel (gemm256.hip).  Returns false when the problem does not meet its shape requirements.
bool sconf_gemm256_eligible(const gemm_tile::GemmParams& p, int layout);
int  sconf_gemm256_width(const gemm_tile::GemmParams& p, int layout);      // 256 or 192 (NT), 0 = not eligible
int  sconf_gemm256_launch(const gemm_tile::GemmParams& p, int layout, hipStream_t stream);
