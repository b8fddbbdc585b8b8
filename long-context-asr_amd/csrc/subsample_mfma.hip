// Fused subsampler stage 0 -> 1 with conv0 on the matrix cores (gfx950).
//
// conv0 (Conv2d 1 -> C, 3x3, stride 2, subsampling.py:299-306) is a (positions x 9) . (9 x C) product: 9 multiply-adds per
// output on the VALU, repeated for 10.7 G outputs per 128 x 16384-frame batch, made the VALU-only kernels of subsample.hip
// issue-bound (stage01_fwd 9.0 ms, stage01_bwd 20.3 ms at 6-9 % of HBM peak, zero MFMA instructions).  Here
//   * the mel patch of one conv0 row is expanded in LDS to im2col form, one 32-byte row per position:
//         [ 9 taps | 1.0 | 1.0 | 0 x 5 ]  (bf16 - the operand precision of the reference's autocast conv),
//     so an MFMA A fragment is ONE ds_read_b128 and the bias rides in the product as two extra K columns
//     (bias = hi + lo in bf16: exact to 2^-17);
//   * pre0[32 positions][32 channels] = patch . W0^T is one v_mfma_f32_32x32x16_bf16 (32 cycles for 1024 outputs, against
//     576 VALU cycles); a wave owns one 32-channel block, its W0 fragment stays in 4 registers;
//   * SiLU, the depthwise 3x3 (forward) and the elementwise part of the backward stay on the VALU, with the lane = channel
//     layout the MFMA accumulator already has; the 3-row window of stage-0 activations lives in LDS as bf16 PAIRS of
//     adjacent frequency bins, so the depthwise conv is v_dot2c_f32_bf16 on whole pairs (no unpacking);
//   * backward: dW0^T[k][c] += patch^T . dP is again an MFMA (A = hardware-transposed read of the same im2col rows, B = the
//     packed dP accumulator), and because column 9 of the patch is 1.0 its row 9 IS the bias gradient.
// Taken when C % 32 == 0 and F/2 <= 64 positions (the paper configs: F = 80, C = 256 / 512); other shapes keep subsample.hip.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int PK = 16;                  // im2col columns per position (32 bytes)
constexpr int PPOS = 64;                // positions per conv0 row image (two 32-row MFMA blocks)
constexpr int PATCH_BYTES = (PPOS + 1) * PK * 2;   // + one overrun row: the transposed reads of the backward touch row 64

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2_t t = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ int acc_pos(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// One conv0 row t2 of the im2col image, entry (pos, column pair kp): fetched from global memory (clamped, unconditional loads;
// the zero padding and the rows outside [0,T2) are applied as selects) - fetched one row AHEAD into registers, stored later.
template <typename TX> struct PatchEntry {
    float v0, v1;
    __device__ __forceinline__ void fetch(const TX* __restrict__ xb, int F, int T, int F2, int T2, int t2, int idx) {
        const int pos = idx >> 3, kp = idx & 7;
        const bool row_ok = t2 >= 0 && t2 < T2 && pos < F2;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = 2 * kp + e, kc = min(k, 8);          // the load is unconditional (a branch around it costs a round trip per lane group)
            const int a = kc / 3, b = kc - 3 * a, t = 2 * t2 + a - 1, f = 2 * pos + b - 1;
            const float m = ld_f(xb + (long)min(max(f, 0), F - 1) * T + min(max(t, 0), T - 1));
            const float tap = (row_ok && t >= 0 && t < T && f >= 0 && f < F) ? m : 0.f;
            const float one = row_ok ? 1.f : 0.f;              // bias columns 9, 10: a padding position's pre-activation is exactly 0
            v[e] = k < 9 ? tap : (k < 11 ? one : 0.f);
        }
        v0 = v[0]; v1 = v[1];
    }
    __device__ __forceinline__ void store(char* patch, int idx) const {
        *reinterpret_cast<unsigned*>(patch + (idx >> 3) * (PK * 2) + (idx & 7) * 4) = pack2(v0, v1);
    }
};

// W0 as the MFMA B operand of channel block cb: B[k = 8 hh + j][col = c] = {w0[c][0..8], bias hi, bias lo, 0...}
__device__ __forceinline__ bf16x8 w0_frag(const float* __restrict__ w0g, const float* __restrict__ b0g, int c, int hh) {
    bf16x8 f;
    const float bias = b0g[c];
    const bf16 bh = (bf16)bias;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * hh + j;
        float v = 0.f;
        if (k < 9) v = w0g[c * 9 + min(k, 8)];
        else if (k == 9) v = (float)bh;
        else if (k == 10) v = bias - (float)bh;
        f[j] = (bf16)v;
    }
    return f;
}

// ================================================================================================================
// forward: d1[t4][f4][c] = bd[c] + sum_ij wd[c][i][j] * SiLU(conv0(x))[2t4+i-1][2f4+j-1][c]
// grid (ceil(T4 / rows_per_block), B), 512 threads: wave w owns the channel blocks w, w+8, ...
// LDS: patch[2] | act[3 slots][NPAIR][QS]: stage-0 activations of conv0 row t2 in slot t2 mod 3 as bf16 pairs (f2, f2+1) per
// channel, pair index q = (f2 + 2) >> 1 (pair 0 = the zero padding at f2 = -1), QS = 4 C + 128 bytes (the 128 spread the four
// 16-lane groups of the depthwise read over all banks).
// ================================================================================================================
template <typename TX, int NCB>      // NCB = channel blocks per wave (1: C <= 256, 2: C <= 512)
__global__ __launch_bounds__(512, (NCB == 1 ? 4 : 2)) void stage01_fwd_mfma_kernel(const TX* __restrict__ x, const float* __restrict__ w0g, const float* __restrict__ b0g,
                                                                  const float* __restrict__ wdg, const float* __restrict__ bdg, bf16* __restrict__ d1,
                                                                  int F, int T, int C, int T2, int F2, int T4, int F4, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NPAIR = F2 / 2 + 2, QS = 4 * C + 128, SLOT = NPAIR * QS;
    char* patch = smem;                                  // [2][PATCH_BYTES]
    char* act = smem + 2 * PATCH_BYTES;                  // [3][NPAIR][QS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, cl = lane & 31;
    const int b = blockIdx.y;
    const TX* xb = x + (long)b * F * T;
    const int ncb = C / 32;

    bf16x8 wf[NCB];
    unsigned wj0[NCB][3][2], wj12[NCB][3][2];            // depthwise taps as bf16 pairs: (0, w[i][0]) and (w[i][1], w[i][2]), 2 channels
    float bd2[NCB][2];
    const int cp = lane & 15, fg = lane >> 4;            // depthwise phase: channel pair and frequency group of this lane
#pragma unroll
    for (int u = 0; u < NCB; ++u) {
        const int cb = min(wave + 8 * u, ncb - 1);
        wf[u] = w0_frag(w0g, b0g, cb * 32 + cl, hh);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int c = cb * 32 + 2 * cp + ch;
            bd2[u][ch] = bdg[c];
#pragma unroll
            for (int i = 0; i < 3; ++i) { wj0[u][i][ch] = pack2(0.f, wdg[c * 9 + 3 * i]); wj12[u][i][ch] = pack2(wdg[c * 9 + 3 * i + 1], wdg[c * 9 + 3 * i + 2]); }
        }
    }
    // zero the padding pair (q = 0) of the three slots and the overrun rows of the patch images
    for (int i = tid; i < 3 * (QS / 4); i += 512) *reinterpret_cast<unsigned*>(act + (i / (QS / 4)) * SLOT + (i % (QS / 4)) * 4) = 0u;
    if (tid < 2 * PK / 2) *reinterpret_cast<unsigned*>(patch + (tid / (PK / 2)) * PATCH_BYTES + PPOS * PK * 2 + (tid % (PK / 2)) * 4) = 0u;

    const int r0 = blockIdx.x * rows_per_block, r1 = min(T4, r0 + rows_per_block);
    const int t2_first = 2 * r0 - 1, t2_last = 2 * (r1 - 1) + 1;
    // the mel taps of a row are requested TWO rows before the row is computed (one row of SiLU work does not cover an L2 miss)
    PatchEntry<TX> pe, pe2;
    pe.fetch(xb, F, T, F2, T2, t2_first, tid);
    pe.store(patch, tid);
    pe.fetch(xb, F, T, F2, T2, t2_first + 1, tid);
    __syncthreads();
    for (int t2 = t2_first; t2 <= t2_last; ++t2) {
        const int buf = (t2 - t2_first) & 1;
        const char* pcur = patch + buf * PATCH_BYTES;
        if (t2 + 1 < t2_last) pe2.fetch(xb, F, T, F2, T2, t2 + 2, tid);
        char* arow = act + ((t2 + 3) % 3) * SLOT;
#pragma unroll
        for (int u = 0; u < NCB; ++u) {
            if (wave + 8 * u >= ncb) break;
            const int c = (wave + 8 * u) * 32 + cl;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                if (blk * 32 >= F2) break;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(pcur + (blk * 32 + cl) * (PK * 2) + hh * 16);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[u], acc, 0, 0, 0);       // pre0[pos][c], bias included
#pragma unroll
                for (int g = 0; g < 8; ++g) {              // 8 pairs of adjacent positions per lane
                    const int f2 = blk * 32 + acc_pos(2 * g, hh), q = (f2 + 2) >> 1;
                    if (q < NPAIR)                          // positions >= F2 give exactly 0 (their patch rows are zero): the right padding
                        *reinterpret_cast<unsigned*>(arow + q * QS + c * 4) = pack2(siluf_(acc[2 * g]), siluf_(acc[2 * g + 1]));
                }
            }
        }
        if (t2 < t2_last) pe.store(patch + (buf ^ 1) * PATCH_BYTES, tid);
        pe = pe2;
        __syncthreads();                                   // row t2 of the window complete; next patch image complete
        if ((t2 & 1) == 0) continue;                       // depthwise rows are centred on even t2: run after row 2 t4 + 1
        const int t4 = (t2 - 1) >> 1;
        if (t4 < r0) continue;                             // the first row of the block only fills the window (uniform)
#pragma unroll
        for (int u = 0; u < NCB; ++u) {
            if (wave + 8 * u >= ncb) break;
            const int c0 = (wave + 8 * u) * 32 + 2 * cp;
#pragma unroll
            for (int fi = 0; fi < 8; ++fi) {               // <= 32 outputs per row (F2 <= 64): independent chains, unrolled
                const int f4 = fg + 4 * fi;
                if (f4 >= F4) break;
                float a0 = bd2[u][0], a1 = bd2[u][1];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const char* srow = act + ((2 * t4 + i - 1 + 3) % 3) * SLOT + c0 * 4;
                    const uint2 pa = *reinterpret_cast<const uint2*>(srow + f4 * QS);            // pair q = f4:     (2f4-2, 2f4-1)
                    const uint2 pb = *reinterpret_cast<const uint2*>(srow + (f4 + 1) * QS);      // pair q = f4 + 1: (2f4,   2f4+1)
                    a0 = dot2(pa.x, wj0[u][i][0], a0); a0 = dot2(pb.x, wj12[u][i][0], a0);
                    a1 = dot2(pa.y, wj0[u][i][1], a1); a1 = dot2(pb.y, wj12[u][i][1], a1);
                }
                *reinterpret_cast<unsigned*>(d1 + (((long)b * T4 + t4) * F4 + f4) * C + c0) = pack2(a0, a1);
            }
        }
        __syncthreads();                                   // the window slot of row 2 t4 - 1 is overwritten by row 2 t4 + 2
    }
}

// ================================================================================================================
// The later depthwise stage (conv[5]): y[to][fo][c] = bias[c] + sum_ij w[c][i][j] * SiLU(x[2to+i-1][2fo+j-1][c]) on a channels-last bf16
// pre-activation tensor x (B,Ti,Fi,C).  Same structure as the fused stage above minus conv0: an input row is read ONCE, its SiLU
// taken ONCE (the position-centric kernel of subsample.hip recomputes it for each of the 2.25 outputs an element feeds: 500 VALU
// instructions per 8 outputs) into the 3-row LDS window of bf16 pairs, and the taps are v_dot2c_f32_bf16.
// grid (ceil(To / rows_per_block), B), 512 threads; needs C % 32 == 0, C <= 512.
// ================================================================================================================
template <int NCB>
__global__ __launch_bounds__(512, 4) void dwconv_window_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ wdg, const float* __restrict__ bdg,
                                                                   bf16* __restrict__ y, int Ti, int Fi, int C, int To, int Fo, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NPAIR = Fi / 2 + 2, QS = 4 * C + 128, SLOT = NPAIR * QS;
    char* act = smem;                                    // [3][NPAIR][QS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, ncb = C / 32;
    const bf16* xb = x + (long)b * Ti * Fi * C;
    unsigned wj0[NCB][3][2], wj12[NCB][3][2];
    float bd2[NCB][2];
    const int cp = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int u = 0; u < NCB; ++u) {
        const int cb = min(wave + 8 * u, ncb - 1);
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int c = cb * 32 + 2 * cp + ch;
            bd2[u][ch] = bdg[c];
#pragma unroll
            for (int i = 0; i < 3; ++i) { wj0[u][i][ch] = pack2(0.f, wdg[c * 9 + 3 * i]); wj12[u][i][ch] = pack2(wdg[c * 9 + 3 * i + 1], wdg[c * 9 + 3 * i + 2]); }
        }
    }
    for (int i = tid; i < 3 * (QS / 4); i += 512) *reinterpret_cast<unsigned*>(act + (i / (QS / 4)) * SLOT + (i % (QS / 4)) * 4) = 0u;

    // producer: item = (pair q' of adjacent input bins (2q', 2q'+1), group of 8 channels): 2 x 16 B in, 8 SiLU pairs (32 B) out
    const int npq = (Fi + 1) / 2, items = npq * (C / 8);
    struct Row { bf16x8 a[2], b[2]; };
    auto fetch = [&](Row& r, int ti) {
        const bool row_ok = ti >= 0 && ti < Ti;
        const bf16* src = xb + (long)min(max(ti, 0), Ti - 1) * Fi * C;
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = min(tid + 512 * it, items - 1), q = idx / (C / 8), c8 = idx - q * (C / 8);
            const bf16x8 va = *reinterpret_cast<const bf16x8*>(src + (long)(2 * q) * C + c8 * 8);
            const bf16x8 vb = *reinterpret_cast<const bf16x8*>(src + (long)min(2 * q + 1, Fi - 1) * C + c8 * 8);
            r.a[it] = row_ok ? va : z;                       // SiLU(0) = 0: rows / bins outside the tensor are the zero padding
            r.b[it] = (row_ok && 2 * q + 1 < Fi) ? vb : z;
        }
    };
    auto produce = [&](const Row& r, int ti) {
        char* arow = act + ((ti + 3) % 3) * SLOT;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 512 * it;
            if (idx < items) {
                const int q = idx / (C / 8), c8 = idx - q * (C / 8);
                unsigned w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = pack2(siluf_((float)r.a[it][e]), siluf_((float)r.b[it][e]));
                uint4* d = reinterpret_cast<uint4*>(arow + (q + 1) * QS + c8 * 32);     // window pair index = (f + 2) >> 1 = q + 1
                d[0] = make_uint4(w[0], w[1], w[2], w[3]); d[1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
    };

    const int r0 = blockIdx.x * rows_per_block, r1 = min(To, r0 + rows_per_block);
    const int ti_first = 2 * r0 - 1, ti_last = 2 * (r1 - 1) + 1;
    Row row;
    fetch(row, ti_first);
    __syncthreads();                                         // the zero pairs
    for (int ti = ti_first; ti <= ti_last; ++ti) {
        produce(row, ti);
        if (ti < ti_last) fetch(row, ti + 1);               // in flight during the depthwise phase / the barrier
        __syncthreads();                                     // row ti of the window complete
        if ((ti & 1) == 0) continue;
        const int to = (ti - 1) >> 1;
        if (to < r0) continue;
#pragma unroll
        for (int u = 0; u < NCB; ++u) {
            if (wave + 8 * u >= ncb) break;
            const int c0 = (wave + 8 * u) * 32 + 2 * cp;
            for (int fo = fg; fo < Fo; fo += 4) {
                float a0 = bd2[u][0], a1 = bd2[u][1];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const char* srow = act + ((2 * to + i - 1 + 3) % 3) * SLOT + c0 * 4;
                    const uint2 pa = *reinterpret_cast<const uint2*>(srow + fo * QS);
                    const uint2 pb = *reinterpret_cast<const uint2*>(srow + (fo + 1) * QS);
                    a0 = dot2(pa.x, wj0[u][i][0], a0); a0 = dot2(pb.x, wj12[u][i][0], a0);
                    a1 = dot2(pa.y, wj0[u][i][1], a1); a1 = dot2(pb.y, wj12[u][i][1], a1);
                }
                *reinterpret_cast<unsigned*>(y + (((long)b * To + to) * Fo + fo) * C + c0) = pack2(a0, a1);
            }
        }
        __syncthreads();                                     // the slot of row 2 to - 1 is overwritten by row 2 to + 2
    }
}

// ================================================================================================================
// backward: parameter gradients of conv0 (dw0, db0) and of the first depthwise conv (dwd, dbd) from dd1 (B,T4,F4,C), one pass over
// the conv0 positions.  grid (ceil(T2 / rows_per_block), B), 512 threads, rows_per_block even; wave w owns channel blocks w, w+8.
// Per conv0 row t2 and channel block:   pre = patch . W0^T (MFMA)  ->  sg = sigmoid(pre), s = pre sg
//   for the (i, j) with 2 to + i - 1 = t2, 2 fo + j - 1 = f2:  g = dd1[to][fo];  gs += wd[i][j] g;  dwd[i][j] += g s;  dbd += g at (1,1)
//   dp = gs sg (1 + pre (1 - sg));   dW0^T[k][c] += sum_pos patch[pos][k] dp[pos][c]  (MFMA; k = 9 is the 1.0 column: db0)
// The accumulator layout gives a lane ONE channel and 16 positions, 4 consecutive f2 per register group: which taps an element
// has is known at compile time from the register index (f2 parity) and one uniform branch per row (t2 parity).
// LDS: patch[2] | gimg[2 slots = to & 1][F4][C] bf16 = the dd1 rows in use, copied by LDS-DMA.
// ================================================================================================================
template <typename TX, int NCB>
__global__ __launch_bounds__(512, (NCB == 1 ? 4 : 2)) void stage01_bwd_mfma_kernel(const TX* __restrict__ x, const float* __restrict__ w0g, const float* __restrict__ b0g,
                                                                  const float* __restrict__ wdg, const bf16* __restrict__ dd1,
                                                                  float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dwd, float* __restrict__ dbd,
                                                                  int F, int T, int C, int T2, int F2, int T4, int F4, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int GS = 2 * C, GSLOT = (F4 + 2) * GS;          // a dd1 row + two zero bins (fo = F4, F4 + 1: taps past the last output)
    char* patch = smem;                                  // [2][PATCH_BYTES]
    char* gimg = smem + 2 * PATCH_BYTES;                 // [3][F4 + 2][C] bf16: slots to & 1, slot 2 = zeros (rows outside [0, T4))
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, cl = lane & 31;
    const int b = blockIdx.y;
    const TX* xb = x + (long)b * F * T;
    const bf16* gb = dd1 + (long)b * T4 * F4 * C;
    const int ncb = C / 32;

    bf16x8 wf[NCB];
    float wd[NCB][9], gwd[NCB][9], gbd[NCB];
    f32x16 aw0[NCB];                                     // dW0^T[k][c] (row 9 = db0)
#pragma unroll
    for (int u = 0; u < NCB; ++u) {
        const int c = min(wave + 8 * u, ncb - 1) * 32 + cl;
        wf[u] = w0_frag(w0g, b0g, c, hh);
        gbd[u] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { wd[u][k] = wdg[c * 9 + k]; gwd[u][k] = 0.f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) aw0[u][r] = 0.f;
    }
    if (tid < 2 * PK / 2) *reinterpret_cast<unsigned*>(patch + (tid / (PK / 2)) * PATCH_BYTES + PPOS * PK * 2 + (tid % (PK / 2)) * 4) = 0u;

    // one dd1 row `to` ([F4][C] bf16, contiguous) -> LDS slot to & 1 by LDS-DMA issued from inline asm (no staging registers, no
    // ds_write, invisible to hipcc's vmcnt bookkeeping: waited for by hand before the barrier that publishes it).  Rows outside
    // [0, T4) are not copied: their users read the all-zero slot 2; bins >= F4 of every slot stay zero (set once below) - so the
    // tap reads need neither clamps nor selects.
    for (int i = tid; i < 3 * GSLOT / 4; i += 512) *reinterpret_cast<unsigned*>(gimg + i * 4) = 0u;
    __syncthreads();                                     // before any wave's DMA can land in a slot another wave is still zeroing
    const unsigned lds_g = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)gimg);
    const int gchunks = F4 * C / 8;                       // 16-byte pieces per row
    auto gissue = [&](int to) {
        if (to < 0 || to >= T4) return;                                        // uniform
        const bf16* src = gb + (long)to * F4 * C;
        const unsigned dst = lds_g + (unsigned)((to & 1) * GSLOT) + (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
        for (int i = 0; i * 512 < gchunks; ++i) {
            const int ch = tid + 512 * i;
            if (ch < gchunks) {                                                // lanes past the row stay out (the zero bins follow it)
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src + (long)ch * 8), "s"(dst + (unsigned)(i * 512 * 16)) : "memory");
            }
        }
    };

    const int r0 = blockIdx.x * rows_per_block, r1 = min(T2, r0 + rows_per_block);     // r0 even
    PatchEntry<TX> pe, pe2;
    pe.fetch(xb, F, T, F2, T2, r0, tid);
    gissue(r0 >> 1);
    pe.store(patch, tid);
    pe.fetch(xb, F, T, F2, T2, r0 + 1, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t2 = r0; t2 < r1; ++t2) {
        const int buf = (t2 - r0) & 1;
        const char* pcur = patch + buf * PATCH_BYTES;
        const bool odd = t2 & 1;
        if (t2 + 2 < r1) pe2.fetch(xb, F, T, F2, T2, t2 + 2, tid);      // mel taps two rows ahead
        if (!odd) gissue((t2 >> 1) + 1);                   // the row the next (odd) conv0 row needs on top of this one's
        // dd1 rows of this conv0 row: even t2: to = t2/2 (tap row 1); odd: to = (t2+1)/2 (tap row 0) and (t2-1)/2 (tap row 2)
        const int toX = (odd ? t2 + 1 : t2) >> 1, toZ = (t2 - 1) >> 1;
        const char* gX = gimg + (toX < T4 ? (toX & 1) : 2) * GSLOT;    // tap row 1 (even) / 0 (odd); past the tensor: the zero slot
        const char* gZ = gimg + (toZ & 1) * GSLOT;                       // tap row 2 (odd rows only; toZ >= 0 always)
#pragma unroll
        for (int u = 0; u < NCB; ++u) {
            if (wave + 8 * u >= ncb) break;
            const int c = (wave + 8 * u) * 32 + cl;
#pragma unroll 1
            for (int blk = 0; blk < 2; ++blk) {            // a real loop: unrolled, hipcc interleaves both blocks and needs 30 more registers
                if (blk * 32 >= F2) break;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(pcur + (blk * 32 + cl) * (PK * 2) + hh * 16);
                f32x16 pre;
#pragma unroll
                for (int r = 0; r < 16; ++r) pre[r] = 0.f;
                pre = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[u], pre, 0, 0, 0);
                unsigned dpk[8];                           // dP packed to bf16 pairs as it is produced (the MFMA B operand below)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {           // 4 consecutive positions base .. base + 3 per register group
                    const int base = blk * 32 + 8 * g4 + 4 * hh, p = base >> 2;
                    if (base >= F2) { dpk[2 * g4] = 0u; dpk[2 * g4 + 1] = 0u; continue; }
                    float sg[4], sv[4], gs[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = pre[4 * g4 + e]; sg[e] = sigmoidf_(v); sv[e] = v * sg[e]; gs[e] = 0.f; }
                    auto taps = [&](const char* grow, int i) {   // one dd1 row, tap row i: fo = 2p, 2p+1, 2p+2 serve f2 = base..base+3
                        const unsigned short* gp_ = reinterpret_cast<const unsigned short*>(grow + (2 * p) * GS + c * 2);
                        const float g0 = __builtin_bit_cast(float, (unsigned)gp_[0] << 16);
                        const float g1 = __builtin_bit_cast(float, (unsigned)gp_[C] << 16);
                        const float g2 = __builtin_bit_cast(float, (unsigned)gp_[2 * C] << 16);
                        const float* w = wd[u] + 3 * i;
                        float* gw = gwd[u] + 3 * i;
                        // f2 = base   (even): j = 1, fo = 2p        f2 = base+1 (odd): j = 0, fo = 2p+1;  j = 2, fo = 2p
                        // f2 = base+2 (even): j = 1, fo = 2p+1      f2 = base+3 (odd): j = 0, fo = 2p+2;  j = 2, fo = 2p+1
                        gs[0] += w[1] * g0;                 gw[1] += g0 * sv[0];
                        gs[1] += w[0] * g1 + w[2] * g0;     gw[0] += g1 * sv[1]; gw[2] += g0 * sv[1];
                        gs[2] += w[1] * g1;                 gw[1] += g1 * sv[2];
                        gs[3] += w[0] * g2 + w[2] * g1;     gw[0] += g2 * sv[3]; gw[2] += g1 * sv[3];
                        if (i == 1) gbd[u] += g0 + g1;      // the centre tap visits every dd1 element exactly once
                    };
                    if (!odd) taps(gX, 1);
                    else { taps(gX, 0); taps(gZ, 2); }
                    float d4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = pre[4 * g4 + e];
                        const float d = gs[e] * sg[e] * (1.f + v * (1.f - sg[e]));
                        d4[e] = (base + e < F2) ? d : 0.f;
                    }
                    dpk[2 * g4] = pack2(d4[0], d4[1]); dpk[2 * g4 + 1] = pack2(d4[2], d4[3]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the four groups' loads and temporaries from piling up
                }
                // dW0^T += patch^T . dP: A = transposed read of the im2col rows (k on the MFMA rows), B = the packed accumulator
                {
                    typedef __attribute__((address_space(3))) bf16x4* lds_p;
                    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, G = (lane >> 4) & 1;
                    const char* tb = pcur + (blk * 32 + 4 * hh + q4) * (PK * 2) + G * 32 + (p4 >> 1) * 16 + (p4 & 1) * 8;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if (blk * 32 + 16 * half >= F2) break;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(tb + (16 * half) * (PK * 2)));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(tb + (16 * half + 8) * (PK * 2)));
                        const bf16x8 at = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        const uint4 bw = make_uint4(dpk[4 * half], dpk[4 * half + 1], dpk[4 * half + 2], dpk[4 * half + 3]);
                        aw0[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at, __builtin_bit_cast(bf16x8, bw), aw0[u], 0, 0, 0);
                    }
                }
            }
        }
        if (t2 + 1 < r1) pe.store(patch + (buf ^ 1) * PATCH_BYTES, tid);
        pe = pe2;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dd1 row issued at the top of this (even) row has landed
        __syncthreads();
    }
    // ---- results: dW0^T rows k = acc_pos(r, hh): 0..3 / 8 / 9 (= db0) for hh = 0, 4..7 for hh = 1; depthwise sums over both halves
#pragma unroll
    for (int u = 0; u < NCB; ++u) {
        if (wave + 8 * u >= ncb) break;
        const int c = (wave + 8 * u) * 32 + cl;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(dw0 + c * 9 + 4 * hh + r, aw0[u][r]);
        if (hh == 0) { atomicAdd(dw0 + c * 9 + 8, aw0[u][4]); atomicAdd(db0 + c, aw0[u][5]); }
#pragma unroll
        for (int k = 0; k < 9; ++k) { const float v = gwd[u][k] + __shfl_xor(gwd[u][k], 32, 64); if (hh == 0) atomicAdd(dwd + c * 9 + k, v); }
        const float v = gbd[u] + __shfl_xor(gbd[u], 32, 64);
        if (hh == 0) atomicAdd(dbd + c, v);
    }
}

}  // namespace

// returns 1 if the MFMA kernels took the problem, 0 if the caller should use the VALU kernels of subsample.hip
int sconf_stage01_fwd_mfma(const void* x, int x_dtype, const float* w0, const float* b0, const float* wd, const float* bd, void* d1,
                           int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream) {
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1), T4 = (T2 - 1) / 2 + 1, F4 = (F2 - 1) / 2 + 1;
    if (C % 32 != 0 || C > 512 || F2 > PPOS) return 0;
    if (const char* e = getenv("SCONF_SUB_MFMA")) if (e[0] == '0') return 0;                   // A/B switch
    const size_t sh = 2 * (size_t)PATCH_BYTES + 3 * (size_t)(F2 / 2 + 2) * (4 * C + 128);
    if (sh > 160 * 1024) return 0;
    long target = 4096;
    if (const char* e = getenv("SCONF_SUB_FWD_BLOCKS")) target = atol(e);                       // tuning
    const int rpb = std::max(1, (int)cdiv((long)T4 * B, target));
    dim3 grid(cdiv(T4, rpb), (unsigned)B), block(512);
#define LF(TX, NCB_) do { \
        static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void*)stage01_fwd_mfma_kernel<TX, NCB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((stage01_fwd_mfma_kernel<TX, NCB_>), grid, block, sh, stream, (const TX*)x, w0, b0, wd, bd, (bf16*)d1, (int)F, (int)T, (int)C, T2, F2, T4, F4, rpb); } while (0)
    if (x_dtype == SCONF_F32) { if (C <= 256) LF(float, 1); else LF(float, 2); }
    else                      { if (C <= 256) LF(bf16, 1); else LF(bf16, 2); }
#undef LF
    return 1;
}

int sconf_stage01_bwd_mfma(const void* dd1, const void* x, int x_dtype, const float* w0, const float* b0, const float* wd,
                           float* dw0, float* db0, float* dwd, float* dbd, int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream) {
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1), T4 = (T2 - 1) / 2 + 1, F4 = (F2 - 1) / 2 + 1;
    if (C % 32 != 0 || C > 512 || F2 > PPOS) return 0;
    if (const char* e = getenv("SCONF_SUB_MFMA")) if (e[0] == '0') return 0;                   // A/B switch
    if ((F4 * C) % 8 != 0) return 0;
    const size_t gslot = (size_t)(F4 + 2) * 2 * C;
    const size_t sh = 2 * (size_t)PATCH_BYTES + 3 * gslot;
    if (sh > 80 * 1024) return 0;                                                               // two workgroups per CU
    long target = 2048;
    if (const char* e = getenv("SCONF_SUB_BWD_BLOCKS")) target = atol(e);                       // tuning
    int rpb = std::max(2, (int)cdiv((long)T2 * B, target));
    rpb += rpb & 1;                                                                              // even: a block starts on an even conv0 row
    dim3 grid(cdiv(T2, rpb), (unsigned)B), block(512);
#define LB(TX, NCB_) do { \
        if (sh > 64 * 1024) { static bool attr = false; if (!attr) { (void)hipFuncSetAttribute((const void*)stage01_bwd_mfma_kernel<TX, NCB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } } \
        hipLaunchKernelGGL((stage01_bwd_mfma_kernel<TX, NCB_>), grid, block, sh, stream, (const TX*)x, w0, b0, wd, (const bf16*)dd1, dw0, db0, dwd, dbd, (int)F, (int)T, (int)C, T2, F2, T4, F4, rpb); } while (0)
    if (x_dtype == SCONF_F32) { if (C <= 256) LB(float, 1); else LB(float, 2); }
    else                      { if (C <= 256) LB(bf16, 1); else LB(bf16, 2); }
#undef LB
    return 1;
}

int sconf_dwconv_window_fwd(const void* x, const float* w, const float* bias, void* y, int64_t B, int64_t Ti, int64_t Fi, int64_t C, hipStream_t stream) {
    const int To = (int)((Ti - 1) / 2 + 1), Fo = (int)((Fi - 1) / 2 + 1);
    if (C % 32 != 0 || C > 512 || ((Fi + 1) / 2) * (C / 8) > 1024) return 0;
    if (const char* e = getenv("SCONF_SUB_MFMA")) if (e[0] == '0') return 0;                   // A/B switch (shared with the fused stage)
    if (const char* e = getenv("SCONF_SUB_DW2")) if (e[0] == '0') return 0;                    // A/B switch of this kernel alone
    const size_t sh = 3 * (size_t)(Fi / 2 + 2) * (4 * C + 128);
    if (sh > 160 * 1024) return 0;
    long target = 4096;
    if (const char* e = getenv("SCONF_SUB_FWD_BLOCKS")) target = atol(e);
    const int rpb = std::max(1, (int)cdiv((long)To * B, target));
    dim3 grid(cdiv(To, rpb), (unsigned)B), block(512);
#define LW(NCB_) do { \
        static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void*)dwconv_window_fwd_kernel<NCB_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((dwconv_window_fwd_kernel<NCB_>), grid, block, sh, stream, (const bf16*)x, w, bias, (bf16*)y, (int)Ti, (int)Fi, (int)C, To, Fo, rpb); } while (0)
    if (C <= 256) LW(1); else LW(2);
#undef LW
    return 1;
}
