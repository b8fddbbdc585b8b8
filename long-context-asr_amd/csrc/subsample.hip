// Depthwise-striding x8 conv subsampler (ConvSubsampling 'dw_striding', subsampling.py:276-321, 384-428),
// channels-last so the 1x1 convs and the output Linear are plain GEMMs over [positions, C]:
//
//   mel (B,F,T) --conv0 3x3 s2--> pre0 (B,T/2,F/2,C) --[SiLU on load] dw 3x3 s2--> d1 (B,T/4,F/4,C)
//     --GEMM pw+bias--> pre1 --[SiLU on load] dw 3x3 s2--> d2 (B,N,F/8,C) --GEMM pw+bias--> pre2
//     --silu_transpose--> s (B,N,C*F/8) [index c*F8+f, the reference's transpose(1,2).reshape] --GEMM out--> (B,N,d)
//
// Only pre-activations are stored (bf16); SiLU is recomputed by the consumer on load (forward and
// backward), which halves the HBM traffic of the largest activations of the model.  H axis = time,
// W axis = frequency, exactly as the reference's Conv2d on (B,1,T,F).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int CG8 = 8;                               // channels per thread (16-B bf16 accesses)

// Thread geometry shared by the conv kernels: a thread OWNS one group of 8 channels for its whole life (its 72 filter
// taps + bias sit in registers) and walks output positions; a workgroup = cgs channel groups x PL position lanes, so
// each iteration touches PL x (C*2) contiguous bytes.  blockIdx.y = batch item; all index math is 32-bit.
template <int W> __device__ __forceinline__ void gemm_free_loadv(const bf16* p, float (&v)[W]) { if constexpr (W == 8) load8(p, v); else load4(p, v); }
template <int W> __device__ __forceinline__ void gemm_free_storev(bf16* p, const float (&v)[W]) { if constexpr (W == 8) store8(p, v); else store4(p, v); }

struct Geo {
    int cg, plane, c0;
    bool active;
    __device__ __forceinline__ Geo(int C, int PL) {
        const int cgs = C / CG8;
        cg = threadIdx.x % cgs; plane = threadIdx.x / cgs; c0 = cg * CG8;
        active = plane < PL;
    }
};

// ---- conv0: Conv2d(1 -> C, 3x3, stride 2, pad 1) + bias, pre-activation out -----------------
template <typename TX>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, bf16* __restrict__ y,
                                                        int F, int T, int C, int T2, int F2, int PL, int iters) {
    const Geo g(C, PL);
    if (!g.active) return;
    const int b = blockIdx.y, npos = T2 * F2;
    float wk[9][8], bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs[e] = bias[g.c0 + e];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k][e] = w[(g.c0 + e) * 9 + k]; }
    const TX* xb = x + (long)b * F * T;
    bf16* yb = y + (long)b * npos * C;
    for (int it = 0; it < iters; ++it) {
        const int p = (blockIdx.x * iters + it) * PL + g.plane;
        if (p >= npos) break;
        const int t2 = p / F2, f2 = p - t2 * F2;
        float in[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = 2 * t2 + i - 1, f = 2 * f2 + j - 1;
                in[i * 3 + j] = (t >= 0 && t < T && f >= 0 && f < F) ? ld_f(xb + (long)f * T + t) : 0.f;
            }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = bs[e];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += wk[k][e] * in[k];
            o[e] = acc;
        }
        store8(yb + (long)p * C + g.c0, o);
    }
}

// ---- depthwise Conv2d(C, 3x3, stride 2, pad 1, groups=C) + bias on SiLU(in) ------------------
__global__ __launch_bounds__(256) void dwconv2d_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, bf16* __restrict__ y,
                                                           int Ti, int Fi, int C, int To, int Fo, int PL, int iters) {
    const Geo g(C, PL);
    if (!g.active) return;
    const int b = blockIdx.y, npos = To * Fo;
    float wk[9][8], bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs[e] = bias[g.c0 + e];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k][e] = w[(g.c0 + e) * 9 + k]; }
    const bf16* xb = x + (long)b * Ti * Fi * C + g.c0;
    bf16* yb = y + (long)b * npos * C + g.c0;
    for (int it = 0; it < iters; ++it) {
        const int p = (blockIdx.x * iters + it) * PL + g.plane;
        if (p >= npos) break;
        const int to = p / Fo, fo = p - to * Fo;
        // All nine tap loads are issued unconditionally (address clamped into the image, contribution zeroed afterwards): a
        // load under a run-time bounds test makes hipcc branch around it and wait vmcnt(0) per tap - nine dependent round trips.
        bf16x8 raw[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = min(max(2 * to + i - 1, 0), Ti - 1), f = min(max(2 * fo + j - 1, 0), Fi - 1);
                raw[i * 3 + j] = *reinterpret_cast<const bf16x8*>(xb + ((long)t * Fi + f) * C);
            }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bs[e];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = 2 * to + i - 1, f = 2 * fo + j - 1;
                const float in = (t >= 0 && t < Ti && f >= 0 && f < Fi) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += wk[i * 3 + j][e] * (in * siluf_((float)raw[i * 3 + j][e]));
            }
        store8(yb + (long)p * C, acc);
    }
}

// ---- backward of the depthwise conv on SiLU(pre_in): input gradient AND weight / bias gradients in one pass -------------
// Walks the INPUT positions (ti, fi): each reads its pre-activation once (sigmoid shared by SiLU and SiLU'), gathers the 1, 2
// or 4 output gradients it feeds, writes dpre_in = SiLU'(pre) * sum w * g, and accumulates dw[i][j] += g * SiLU(pre) per
// (i, j) in registers (dbias += g at the centre tap, which visits every output exactly once).  Workgroup reduction over the
// position lanes through LDS, then one atomic per sum.
template <int CW>
__global__ __launch_bounds__(512) void dwconv2d_bwd_kernel(const bf16* __restrict__ dout, const float* __restrict__ w,
                                                           const bf16* __restrict__ pre_in, bf16* __restrict__ dpre_in,
                                                           float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dcolsum,
                                                           int Ti, int Fi, int C, int To, int Fo, int PL, int iters) {
    __shared__ float red[512];
    const int cgs = C / CW;
    struct { int cg, plane, c0; bool active; } g;
    g.cg = threadIdx.x % cgs; g.plane = threadIdx.x / cgs; g.c0 = g.cg * CW; g.active = g.plane < PL;
    const int b = blockIdx.y, npos = Ti * Fi;
    float wk[9][CW], gw[9][CW], gbs[CW], gcs[CW];       // gcs: column sums of the bf16 output (the bias gradient of the 1x1 conv before it)
#pragma unroll
    for (int e = 0; e < CW; ++e) {
        gbs[e] = 0.f; gcs[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { wk[k][e] = g.active ? w[(g.c0 + e) * 9 + k] : 0.f; gw[k][e] = 0.f; }
    }
    const bf16* gb = dout + (long)b * To * Fo * C + g.c0;
    const bf16* pb = pre_in + (long)b * npos * C + g.c0;
    bf16* ob = dpre_in + (long)b * npos * C + g.c0;
    if (g.active) {
        for (int it = 0; it < iters; ++it) {
            const int p = (blockIdx.x * iters + it) * PL + g.plane;
            if (p >= npos) break;
            const int ti = p / Fi, fi = p - ti * Fi;
            // The (up to) four output gradients this input position feeds sit at (to0 - a, fo0 - b), a, b in {0, 1}, with
            // to0 = (ti + 1) / 2, fo0 = (fi + 1) / 2 (a = 1 only for odd ti: tap row 2; a = 0 is tap row 0 or 1).  All four
            // are loaded up front with clamped addresses - a load under each tap's run-time test cost a round trip per tap.
            const int to0 = (ti + 1) >> 1, fo0 = (fi + 1) >> 1;
            float gq[2][2][CW];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int to = min(max(to0 - a, 0), To - 1), fo = min(max(fo0 - bb, 0), Fo - 1);
                    gemm_free_loadv<CW>(gb + ((long)to * Fo + fo) * C, gq[a][bb]);
                }
            float pv[CW], sg[CW], sv[CW]; gemm_free_loadv<CW>(pb + (long)p * C, pv);
#pragma unroll
            for (int e = 0; e < CW; ++e) { sg[e] = sigmoidf_(pv[e]); sv[e] = pv[e] * sg[e]; }
            float acc[CW];
#pragma unroll
            for (int e = 0; e < CW; ++e) acc[e] = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int tt = ti + 1 - i;                         // = 2*to
                if (tt < 0 || (tt & 1)) continue;
                const int to = tt >> 1;
                if (to >= To) continue;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ff = fi + 1 - j;
                    if (ff < 0 || (ff & 1)) continue;
                    const int fo = ff >> 1;
                    if (fo >= Fo) continue;
                    const float (&gv)[CW] = gq[i == 2][j == 2];            // (to, fo) = (to0 - (i == 2), fo0 - (j == 2))
#pragma unroll
                    for (int e = 0; e < CW; ++e) {
                        acc[e] += wk[i * 3 + j][e] * gv[e];
                        gw[i * 3 + j][e] += gv[e] * sv[e];
                        if (i == 1 && j == 1) gbs[e] += gv[e];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < CW; ++e) { acc[e] *= sg[e] * (1.f + pv[e] * (1.f - sg[e])); gcs[e] += (float)(bf16)acc[e]; }   // sums of what is stored
            gemm_free_storev<CW>(ob + (long)p * C, acc);
        }
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        if (k == 10 && !dcolsum) break;
#pragma unroll
        for (int e = 0; e < CW; ++e) {
            __syncthreads();
            red[threadIdx.x] = g.active ? (k < 9 ? gw[k][e] : (k == 9 ? gbs[e] : gcs[e])) : 0.f;
            __syncthreads();
            if (g.plane == 0) {
                float v = 0.f;
                for (int pl = 0; pl < PL; ++pl) v += red[pl * cgs + g.cg];
                if (k < 9) atomicAdd(dw + (g.c0 + e) * 9 + k, v); else if (k == 9) atomicAdd(dbias + g.c0 + e, v); else atomicAdd(dcolsum + g.c0 + e, v);
            }
        }
    }
}

// ---- weight / bias gradients of a 3x3 stride-2 conv whose output gradient is channels-last -----
// DEPTHWISE: input is SiLU(pre_in[.., c]); otherwise (conv0) the single-channel mel input (B,F,T).
// Each thread accumulates its channel group's 72+8 sums over its positions in registers; the workgroup then reduces
// across its position lanes through LDS and issues ONE atomic per sum (a few hundred workgroups in total), because
// per-thread atomics on the ~C*10 distinct addresses serialise.
template <bool DEPTHWISE, typename TX>
__global__ __launch_bounds__(256) void conv3x3s2_bwd_weight_kernel(const bf16* __restrict__ dout, const void* __restrict__ in_,
                                                                   float* __restrict__ dw, float* __restrict__ dbias,
                                                                   int Ti, int Fi, int C, int To, int Fo, int PL, int iters) {
    __shared__ float red[256];
    const Geo g(C, PL);
    const int cgs = C / CG8;
    const int b = blockIdx.y, npos = To * Fo;
    float gw[9][8], gbs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gbs[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) gw[k][e] = 0.f; }
    const bf16* gp = dout + (long)b * npos * C + g.c0;
    if (g.active) {
        for (int it = 0; it < iters; ++it) {
            const int p = (blockIdx.x * iters + it) * PL + g.plane;
            if (p >= npos) break;
            const int to = p / Fo, fo = p - to * Fo;
            float gv[8]; load8(gp + (long)p * C, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) gbs[e] += gv[e];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = 2 * to + i - 1, f = 2 * fo + j - 1;
                    if (t < 0 || t >= Ti || f < 0 || f >= Fi) continue;
                    if (DEPTHWISE) {
                        float v[8]; load8((const bf16*)in_ + (((long)b * Ti + t) * Fi + f) * C + g.c0, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) gw[i * 3 + j][e] += gv[e] * siluf_(v[e]);
                    } else {
                        const float v = ld_f((const TX*)in_ + ((long)b * Fi + f) * Ti + t);   // mel is (B,F,T)
#pragma unroll
                        for (int e = 0; e < 8; ++e) gw[i * 3 + j][e] += gv[e] * v;
                    }
                }
        }
    }
    // workgroup reduction over position lanes, one value at a time (80 values per channel group)
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __syncthreads();
            red[threadIdx.x] = g.active ? (k < 9 ? gw[k][e] : gbs[e]) : 0.f;
            __syncthreads();
            if (g.plane == 0) {
                float v = 0.f;
                for (int pl = 0; pl < PL; ++pl) v += red[pl * cgs + g.cg];
                if (k < 9) atomicAdd(dw + (g.c0 + e) * 9 + k, v); else atomicAdd(dbias + g.c0 + e, v);
            }
        }
}

// ---- SiLU + (F8, C) -> (C, F8) transpose of one token's features, through LDS ------------------
// fwd: pre (rows, F8, C) -> s (rows, C*F8) = SiLU(pre) at index c*F8+f
// bwd: ds (rows, C*F8), pre -> dpre (rows, F8, C) = ds^T * SiLU'(pre)
template <bool BWD>
__global__ __launch_bounds__(256) void silu_transpose_kernel(const bf16* __restrict__ pre, const bf16* __restrict__ ds,
                                                             bf16* __restrict__ out, long rows, int F8, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16* sh = reinterpret_cast<bf16*>(smem_);
    const int E = F8 * C;                               // elements per row
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const bf16* pr = pre + row * E;
        if (!BWD) {
            for (int i = threadIdx.x * 8; i < E; i += 256 * 8) {   // i = f*C + c
                float v[8]; load8(pr + i, v);
                const int f = i / C, c = i % C;
#pragma unroll
                for (int e = 0; e < 8; ++e) sh[(c + e) * F8 + f] = (bf16)siluf_(v[e]);
            }
            __syncthreads();
            for (int i = threadIdx.x * 8; i < E; i += 256 * 8)
                *reinterpret_cast<uint4*>(out + row * E + i) = *reinterpret_cast<const uint4*>(sh + i);
            __syncthreads();
        } else {
            for (int i = threadIdx.x * 8; i < E; i += 256 * 8)
                *reinterpret_cast<uint4*>(sh + i) = *reinterpret_cast<const uint4*>(ds + row * E + i);
            __syncthreads();
            for (int i = threadIdx.x * 8; i < E; i += 256 * 8) {
                float v[8]; load8(pr + i, v);
                const int f = i / C, c = i % C;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)sh[(c + e) * F8 + f] * dsiluf_(v[e]);
                store8(out + row * E + i, v);
            }
            __syncthreads();
        }
    }
}

// =================================================================================================================
// Fused stage 0 -> 1: conv0 (1->C, 3x3 s2) + SiLU + depthwise 3x3 s2, WITHOUT materialising the (B,T/2,F/2,C) tensor
// (2.7 GB bf16 per 16 x 16384-frame batch; 2.7 GB more for its gradient).  conv0 is 9 MACs per output on a single-channel
// input, so it is recomputed from a mel patch staged in LDS wherever it is needed (forward, dw-conv weight gradient,
// conv0 weight gradient): the three kernels below read the mel rows + the small stage-1 tensors only.
// Geometry: a workgroup walks time rows; per row it stages NT mel time-columns x F bins in LDS; a thread owns 4 channels
// (its 2 x 9 taps + biases in registers) and loops over the row's frequency positions.
// =================================================================================================================
constexpr int FC = 4;                                    // channels per thread in the fused kernels

template <typename TX, int NT>
__device__ __forceinline__ void stage_mel(const TX* __restrict__ xb, float* xs, int F, int T, int t_base) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < NT * F; idx += 256) {
        const int f = idx / NT, tt = idx - f * NT, t = t_base + tt;
        const float v = ld_f(xb + (long)f * T + min(max(t, 0), T - 1));        // unconditional (clamped) load: no branch, no
        xs[tt * F + f] = (t >= 0 && t < T) ? v : 0.f;                          // per-element vmcnt(0); zero padding applied after
    }
    __syncthreads();
}
// pre0 at conv0 position (t2, f2) for 4 channels, from the LDS mel patch whose row 0 is time t_base
__device__ __forceinline__ void conv0_at(const float* xs, int F, int tt0, int f2, const float (&w0)[9][FC], const float (&b0)[FC],
                                         float (&pre)[FC], float (&xin)[9]) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int f = 2 * f2 + b - 1;
            xin[a * 3 + b] = (f >= 0 && f < F) ? xs[(tt0 + a) * F + f] : 0.f;
        }
#pragma unroll
    for (int e = 0; e < FC; ++e) {
        float acc = b0[e];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += w0[k][e] * xin[k];
        pre[e] = acc;
    }
}

struct FGeo {                                            // thread -> (channel group, position lane)
    int cg, plane, c0, PL;
    bool active;
    __device__ __forceinline__ FGeo(int C, int nthreads = 256) {
        const int cgs = C / FC;
        PL = max(1, nthreads / cgs);
        cg = threadIdx.x % cgs; plane = threadIdx.x / cgs; c0 = cg * FC;
        active = plane < PL;
    }
};

// d1[t4][f4][c] = bd[c] + sum_{i,j} wd[c][i][j] * SiLU(pre0[2t4+i-1][2f4+j-1][c])      (zero padding of the SiLU output)
// Every stage-0 activation feeds 1, 2 or 4 outputs (2.25 on average), so it is computed ONCE into a rolling 3-row LDS window
// ([3][F2][C] bf16 - the precision the reference's autocast conv output has) and the depthwise conv reads the window:
// per time row t4: stage 5 mel columns -> compute stage-0 rows 2t4, 2t4+1 (row 2t4-1 is the previous step's 2t4+1) -> dw.
template <typename TX>
__global__ __launch_bounds__(512) void stage01_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ w0g, const float* __restrict__ b0g,
                                                          const float* __restrict__ wdg, const float* __restrict__ bdg, bf16* __restrict__ d1,
                                                          int F, int T, int C, int T2, int F2, int T4, int F4, int rows_per_block) {
    extern __shared__ float xs[];                        // [7][F] mel | [3][F2][C] bf16 stage-0 activations (slot = t2 mod 3)
    bf16* act = reinterpret_cast<bf16*>(xs + 7 * F);
    const FGeo g(C, blockDim.x);
    const int b = blockIdx.y;
    const int rowel = F2 * C;
    float w0[9][FC], b0[FC], wd[9][FC], bd[FC];
#pragma unroll
    for (int e = 0; e < FC; ++e) {
        const int c = g.active ? g.c0 + e : 0;
        b0[e] = b0g[c]; bd[e] = bdg[c];
#pragma unroll
        for (int k = 0; k < 9; ++k) { w0[k][e] = w0g[c * 9 + k]; wd[k][e] = wdg[c * 9 + k]; }
    }
    const TX* xb = x + (long)b * F * T;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(T4, r0 + rows_per_block);
    for (int t4 = r0; t4 < r1; ++t4) {
        const bool first = t4 == r0;                           // the window is empty: row 2t4-1 has to be computed as well
        const int tlo = first ? 2 * t4 - 1 : 2 * t4;            // stage-0 rows tlo .. 2t4+1 are new
        // mel columns 2*tlo-1 .. 2*(2t4+1)+1  (5, or 7 for the first step: staged in two calls of 5 / 3 columns -> use 7-wide buffer)
        __syncthreads();                                       // previous dw phase done with the window + mel patch
        for (int idx = threadIdx.x; idx < 7 * F; idx += blockDim.x) {
            const int f = idx / 7, tt = idx - f * 7, t = 4 * t4 - 3 + tt;
            const float v = ld_f(xb + (long)f * T + min(max(t, 0), T - 1));             // unconditional (clamped) load
            if (tt >= (first ? 0 : 2)) xs[tt * F + f] = (t >= 0 && t < T) ? v : 0.f;
        }
        __syncthreads();
        if (g.active) {
            for (int t2 = tlo; t2 <= 2 * t4 + 1; ++t2) {
                bf16* arow = act + ((t2 + 3) % 3) * rowel;
                const bool inside = t2 >= 0 && t2 < T2;
                for (int f2 = g.plane; f2 < F2; f2 += g.PL) {
                    float a[FC] = {0.f, 0.f, 0.f, 0.f};
                    if (inside) {
                        float pre[FC], xin[9];
                        conv0_at(xs, F, 2 * (t2 - (2 * t4 - 1)), f2, w0, b0, pre, xin);   // mel column of tap a=0: 2t2-1 = (4t4-3) + 2(t2-2t4+1)
#pragma unroll
                        for (int e = 0; e < FC; ++e) a[e] = siluf_(pre[e]);
                    }
                    store4(arow + f2 * C + g.c0, a);           // rows outside [0,T2) are the conv's zero padding
                }
            }
        }
        __syncthreads();
        if (!g.active) continue;
        for (int f4 = g.plane; f4 < F4; f4 += g.PL) {
            float acc[FC];
#pragma unroll
            for (int e = 0; e < FC; ++e) acc[e] = bd[e];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bf16* arow = act + ((2 * t4 + i - 1 + 3) % 3) * rowel;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f2 = 2 * f4 + j - 1;
                    if (f2 < 0 || f2 >= F2) continue;
                    float v[FC]; load4(arow + f2 * C + g.c0, v);
#pragma unroll
                    for (int e = 0; e < FC; ++e) acc[e] += wd[i * 3 + j][e] * v[e];
                }
            }
            store4(d1 + (((long)b * T4 + t4) * F4 + f4) * C + g.c0, acc);
        }
    }
}

// workgroup reduction of NV per-thread sums (4 channels each) over position lanes, then one atomic per value
template <int NV>
__device__ __forceinline__ void fused_reduce(const float (&v)[NV][FC], const FGeo& g, int C, float* red, float* out, int out_stride) {
    const int cgs = C / FC;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < FC; ++e) {
            __syncthreads();
            red[threadIdx.x] = g.active ? v[k][e] : 0.f;
            __syncthreads();
            if (g.plane == 0) {
                float a = 0.f;
                for (int pl = 0; pl < g.PL; ++pl) a += red[pl * cgs + g.cg];
                atomicAdd(out + (long)(g.c0 + e) * out_stride + k, a);
            }
        }
}

// Both parameter gradients of the fused stage in ONE pass over the conv0 positions (t2, f2) - a walk over the
// depthwise OUTPUTS would recompute conv0 + the sigmoid 2.25x per position (every stage-0 activation feeds 1, 2 or 4 outputs),
// and separate kernels for the two convolutions' gradients would each recompute them again (measured: 4.6 ms -> 2.6 ms).  Per position and channel:
//   pre = conv0(x), sg = sigmoid(pre), s = pre * sg, s' = sg * (1 + pre * (1 - sg))
//   for the (i, j) with 2to + i - 1 = t2, 2fo + j - 1 = f2:   g = dd1[to][fo];  gs += wd[i][j] * g;  dwd[i][j] += g * s
//   dp = gs * s';  db0 += dp;  dw0[a][b] += dp * x[2t2 + a - 1][2f2 + b - 1];   dbd += g at the centre tap (counts each g once)
// RW conv0 rows share one staging step (2 RW + 1 mel columns, RW/2 + 2 dd1 rows in LDS, two barriers).
template <typename TX, int RW>
__global__ __launch_bounds__(256) void stage01_bwd_kernel(const TX* __restrict__ x, const float* __restrict__ w0g, const float* __restrict__ b0g,
                                                          const float* __restrict__ wdg, const bf16* __restrict__ dd1,
                                                          float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dwd,
                                                          float* __restrict__ dbd, int F, int T, int C, int T2, int F2, int T4, int F4,
                                                          int rows_per_block) {
    constexpr int NT = 2 * RW + 1, NG = RW / 2 + 2;       // mel time-columns / dd1 rows per staging step
    extern __shared__ float xs[];                        // [NT][F] mel | 256 floats reduction scratch | [NG][F4][C] bf16 dd1 rows
    float* red = xs + NT * F;
    bf16* grow = reinterpret_cast<bf16*>(red + 256);
    const FGeo g(C);
    const int b = blockIdx.y;
    float w0[9][FC], b0[FC], wd[9][FC], gw0[9][FC], gwd[9][FC], gb0[1][FC], gbd[1][FC];
#pragma unroll
    for (int e = 0; e < FC; ++e) {
        const int c = g.active ? g.c0 + e : 0;
        b0[e] = b0g[c]; gb0[0][e] = 0.f; gbd[0][e] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { w0[k][e] = w0g[c * 9 + k]; wd[k][e] = wdg[c * 9 + k]; gw0[k][e] = 0.f; gwd[k][e] = 0.f; }
    }
    const TX* xb = x + (long)b * F * T;
    const bf16* gp = dd1 + (long)b * T4 * F4 * C;
    const int rowel = F4 * C;                             // elements of one dd1 row
    const int r0 = blockIdx.x * rows_per_block, r1 = min(T2, r0 + rows_per_block);
    for (int tb = r0; tb < r1; tb += RW) {
        stage_mel<TX, NT>(xb, xs, F, T, 2 * tb - 1);
        const int to0 = (tb - 1) >> 1;                     // first dd1 row any of the rows tb .. tb+RW-1 can feed (may be -1)
        for (int idx = threadIdx.x * 8; idx < NG * rowel; idx += 256 * 8) {
            const int slot = idx / rowel, to = to0 + slot;
            uint4 v = *reinterpret_cast<const uint4*>(gp + (long)min(max(to, 0), T4 - 1) * rowel + (idx - slot * rowel));   // clamped,
            if (to < 0 || to >= T4) v = make_uint4(0, 0, 0, 0);                                                             // unconditional
            *reinterpret_cast<uint4*>(grow + idx) = v;
        }
        __syncthreads();
        if (!g.active) continue;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int t2 = tb + rr;
            if (t2 >= r1) break;
            for (int f2 = g.plane; f2 < F2; f2 += g.PL) {
                float pre[FC], xin[9], sg[FC], sv[FC];
                conv0_at(xs, F, 2 * rr, f2, w0, b0, pre, xin);
#pragma unroll
                for (int e = 0; e < FC; ++e) { sg[e] = sigmoidf_(pre[e]); sv[e] = pre[e] * sg[e]; }
                float gs[FC] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int tt = t2 + 1 - i;                 // = 2 * to
                    if (tt < 0 || (tt & 1)) continue;
                    const int to = tt >> 1;
                    if (to >= T4) continue;
                    const bf16* grw = grow + (to - to0) * rowel;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int ff = f2 + 1 - j;
                        if (ff < 0 || (ff & 1)) continue;
                        const int fo = ff >> 1;
                        if (fo >= F4) continue;
                        float gv[FC]; load4(grw + fo * C + g.c0, gv);
#pragma unroll
                        for (int e = 0; e < FC; ++e) {
                            gs[e] += wd[i * 3 + j][e] * gv[e];
                            gwd[i * 3 + j][e] += gv[e] * sv[e];
                            if (i == 1 && j == 1) gbd[0][e] += gv[e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < FC; ++e) {
                    const float dp = gs[e] * sg[e] * (1.f + pre[e] * (1.f - sg[e]));
                    gb0[0][e] += dp;
#pragma unroll
                    for (int k = 0; k < 9; ++k) gw0[k][e] += dp * xin[k];
                }
            }
        }
    }
    fused_reduce<9>(gw0, g, C, red, dw0, 9);
    fused_reduce<1>(gb0, g, C, red, db0, 1);
    fused_reduce<9>(gwd, g, C, red, dwd, 9);
    fused_reduce<1>(gbd, g, C, red, dbd, 1);
}

}  // namespace
// subsample_mfma.hip: conv0 on the matrix cores; return 1 when they took the problem
int sconf_stage01_fwd_mfma(const void* x, int x_dtype, const float* w0, const float* b0, const float* wd, const float* bd, void* d1,
                           int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream);
int sconf_dwconv_window_fwd(const void* x, const float* w, const float* bias, void* y, int64_t B, int64_t Ti, int64_t Fi, int64_t C, hipStream_t stream);
int sconf_stage01_bwd_mfma(const void* dd1, const void* x, int x_dtype, const float* w0, const float* b0, const float* wd,
                           float* dw0, float* db0, float* dwd, float* dbd, int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream);
namespace {

struct LaunchGeo { int PL, iters, threads; dim3 grid; };
// npos positions per batch item; aim for ~target workgroups in total.
inline LaunchGeo geo_for(int64_t C, int64_t B, long npos, long target_blocks) {
    LaunchGeo g;
    const int cgs = (int)(C / 8);
    g.PL = std::max(1, 256 / cgs);
    g.threads = (cgs * g.PL + 63) / 64 * 64;
    const long per_b = std::max<long>(1, target_blocks / B);
    g.iters = (int)std::max<long>(1, cdiv(npos, (long)g.PL * per_b));
    g.grid = dim3(cdiv(npos, (long)g.PL * g.iters), (unsigned)B);
    return g;
}
#define SUB_REQ(fn) SCONF_REQUIRE(C % 8 == 0 && C / 8 <= 256, fn ": C must be a multiple of 8 and <= 2048"); \
                    SCONF_REQUIRE(B <= 65535, fn ": B must be <= 65535")

}  // namespace

// Conv2d(1->C,3,s2,p1)+bias on the (B,F,T) mel input; pre-activation out (B,T2,F2,C) bf16.
// Replaces subsampling.py:299-306 (self.conv[0]); SiLU (conv[1]) is applied by the consumer.
SCONF_API int sconf_sub_conv0_fwd(const void* x, int x_dtype, const float* w, const float* bias, void* y,
                                  int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream) {
    SUB_REQ("sconf_sub_conv0_fwd");
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1);
    if (B * T2 * F2 == 0) return 0;
    const LaunchGeo g = geo_for(C, B, (long)T2 * F2, 16384);
    if (x_dtype == SCONF_F32) hipLaunchKernelGGL((conv0_fwd_kernel<float>), g.grid, dim3(g.threads), 0, stream, (const float*)x, w, bias, (bf16*)y, (int)F, (int)T, (int)C, T2, F2, g.PL, g.iters);
    else hipLaunchKernelGGL((conv0_fwd_kernel<bf16>), g.grid, dim3(g.threads), 0, stream, (const bf16*)x, w, bias, (bf16*)y, (int)F, (int)T, (int)C, T2, F2, g.PL, g.iters);
    SCONF_LAUNCH_OK("sconf_sub_conv0_fwd");
    return 0;
}

// y = dwConv2d(SiLU(x)) + bias, channels-last.  Replaces subsampling.py:309-330 (conv[2], conv[5]) fused with
// the preceding activation (conv[1], conv[4]).
SCONF_API int sconf_sub_dwconv_fwd(const void* x, const float* w, const float* bias, void* y,
                                   int64_t B, int64_t Ti, int64_t Fi, int64_t C, hipStream_t stream) {
    SUB_REQ("sconf_sub_dwconv_fwd");
    const int To = (int)((Ti - 1) / 2 + 1), Fo = (int)((Fi - 1) / 2 + 1);
    if (B * To * Fo == 0) return 0;
    if (sconf_dwconv_window_fwd(x, w, bias, y, B, Ti, Fi, C, stream)) { SCONF_LAUNCH_OK("sconf_sub_dwconv_fwd"); return 0; }
    const LaunchGeo g = geo_for(C, B, (long)To * Fo, 16384);
    hipLaunchKernelGGL(dwconv2d_fwd_kernel, g.grid, dim3(g.threads), 0, stream, (const bf16*)x, w, bias, (bf16*)y, (int)Ti, (int)Fi, (int)C, To, Fo, g.PL, g.iters);
    SCONF_LAUNCH_OK("sconf_sub_dwconv_fwd");
    return 0;
}

// dpre_in = SiLU'(pre_in) * dwconv^T(dout);  dw/dbias ACCUMULATED (+=).  dpre_colsum (nullable, f32 [C], ACCUMULATED): the column sums of
// dpre_in as stored (bf16) = the bias gradient of the 1x1 conv that produced pre_in, from the same pass (a separate column-sum
// kernel re-read the 5.4 GB tensor: 0.95 ms per step at B = 128).
SCONF_API int sconf_sub_dwconv_bwd(const void* dout, const float* w, const void* pre_in, void* dpre_in, float* dw, float* dbias,
                                   float* dpre_colsum, int64_t B, int64_t Ti, int64_t Fi, int64_t C, hipStream_t stream) {
    SUB_REQ("sconf_sub_dwconv_bwd");
    const int To = (int)((Ti - 1) / 2 + 1), Fo = (int)((Fi - 1) / 2 + 1);
    if (B * Ti * Fi == 0) return 0;
    long target = 512; int cw = 8;                             // few workgroups (each ends with 10 x C atomics), 16-B accesses
    if (const char* e = getenv("SCONF_SUB_DWBWD_CFG")) { int a = 0; long t = 0; if (sscanf(e, "%d,%ld", &a, &t) == 2) { cw = a; target = t; } }   // tuning
    const int cgs = (int)(C / cw);
    SCONF_REQUIRE(cgs <= 256, "sconf_sub_dwconv_bwd: C too large");
    int nthr = 256;                                            // measured with all tap loads in flight: 3.3 -> 1.9 ms at B = 64
    if (const char* e = getenv("SCONF_SUB_DWBWD_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512) nthr = v; }     // tuning
    const int PL = std::max(1, nthr / cgs), threads = (cgs * PL + 63) / 64 * 64;
    const long npos = (long)Ti * Fi, per_b = std::max<long>(1, target / B);
    const int iters = (int)std::max<long>(1, cdiv(npos, (long)PL * per_b));
    dim3 grid(cdiv(npos, (long)PL * iters), (unsigned)B);
    if (cw == 8) hipLaunchKernelGGL((dwconv2d_bwd_kernel<8>), grid, dim3(threads), 0, stream, (const bf16*)dout, w, (const bf16*)pre_in, (bf16*)dpre_in, dw, dbias, dpre_colsum, (int)Ti, (int)Fi, (int)C, To, Fo, PL, iters);
    else         hipLaunchKernelGGL((dwconv2d_bwd_kernel<4>), grid, dim3(threads), 0, stream, (const bf16*)dout, w, (const bf16*)pre_in, (bf16*)dpre_in, dw, dbias, dpre_colsum, (int)Ti, (int)Fi, (int)C, To, Fo, PL, iters);
    SCONF_LAUNCH_OK("sconf_sub_dwconv_bwd");
    return 0;
}

// conv0 parameter gradients (the mel input needs no gradient).  dw [C][9], dbias [C] ACCUMULATED (+=).
SCONF_API int sconf_sub_conv0_bwd(const void* dpre0, const void* x, int x_dtype, float* dw, float* dbias,
                                  int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream) {
    SUB_REQ("sconf_sub_conv0_bwd");
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1);
    if (B * T2 * F2 == 0) return 0;
    const LaunchGeo g = geo_for(C, B, (long)T2 * F2, 1024);
    if (x_dtype == SCONF_F32) hipLaunchKernelGGL((conv3x3s2_bwd_weight_kernel<false, float>), g.grid, dim3(g.threads), 0, stream, (const bf16*)dpre0, x, dw, dbias, (int)T, (int)F, (int)C, T2, F2, g.PL, g.iters);
    else hipLaunchKernelGGL((conv3x3s2_bwd_weight_kernel<false, bf16>), g.grid, dim3(g.threads), 0, stream, (const bf16*)dpre0, x, dw, dbias, (int)T, (int)F, (int)C, T2, F2, g.PL, g.iters);
    SCONF_LAUNCH_OK("sconf_sub_conv0_bwd");
    return 0;
}

// bwd == 0: out (rows, C*F8) = SiLU(pre (rows,F8,C)) transposed to the reference's c*F8+f feature order
//           (subsampling.py:422-423).  bwd != 0: out (rows,F8,C) = ds^T * SiLU'(pre).
SCONF_API int sconf_sub_silu_transpose(int bwd, const void* pre, const void* ds, void* out, int64_t rows, int64_t F8, int64_t C,
                                       hipStream_t stream) {
    SCONF_REQUIRE(C % 8 == 0, "sconf_sub_silu_transpose: C must be a multiple of 8");
    SCONF_REQUIRE(F8 * C * 2 <= 64 * 1024, "sconf_sub_silu_transpose: row of %ld elements does not fit LDS", (long)(F8 * C));
    if (rows == 0) return 0;
    const int blocks = (int)std::min<long>(rows, 8192);
    const size_t sh = (size_t)F8 * C * 2;
    if (!bwd) hipLaunchKernelGGL((silu_transpose_kernel<false>), dim3(blocks), dim3(256), sh, stream, (const bf16*)pre, (const bf16*)ds, (bf16*)out, (long)rows, (int)F8, (int)C);
    else      hipLaunchKernelGGL((silu_transpose_kernel<true>), dim3(blocks), dim3(256), sh, stream, (const bf16*)pre, (const bf16*)ds, (bf16*)out, (long)rows, (int)F8, (int)C);
    SCONF_LAUNCH_OK("sconf_sub_silu_transpose");
    return 0;
}

// Fused subsampler stage 0->1 forward: d1 (B,T4,F4,C) bf16 = dwConv(SiLU(conv0(x))) + bias, no (B,T/2,F/2,C) tensor.
// Replaces subsampling.py:299-318 (conv[0..2]).
SCONF_API int sconf_sub_stage01_fwd(const void* x, int x_dtype, const float* w0, const float* b0, const float* wd, const float* bd,
                                    void* d1, int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 4 == 0 && C / 4 <= 256, "sconf_sub_stage01_fwd: C must be a multiple of 4 and <= 1024");
    SCONF_REQUIRE(B <= 65535 && F <= 1024, "sconf_sub_stage01_fwd: B <= 65535 and F <= 1024");
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1), T4 = (T2 - 1) / 2 + 1, F4 = (F2 - 1) / 2 + 1;
    if (B * T4 * F4 == 0) return 0;
    if (sconf_stage01_fwd_mfma(x, x_dtype, w0, b0, wd, bd, d1, B, F, T, C, stream)) { SCONF_LAUNCH_OK("sconf_sub_stage01_fwd"); return 0; }
    long target = 4096;
    if (const char* e = getenv("SCONF_SUB_FWD_BLOCKS")) target = atol(e);                       // tuning
    const int rpb = std::max(1, (int)cdiv((long)T4 * B, target));
    dim3 grid(cdiv(T4, rpb), (unsigned)B), block(512);           // 8 waves: two workgroups (LDS-limited) fill a CU's 16 wave slots
    const size_t sh = (size_t)7 * F * 4 + (size_t)3 * F2 * C * 2;
    SCONF_REQUIRE(sh <= 150 * 1024, "sconf_sub_stage01_fwd: the 3-row activation window (%ld B) does not fit LDS", (long)sh);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stage01_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)stage01_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    if (x_dtype == SCONF_F32) hipLaunchKernelGGL((stage01_fwd_kernel<float>), grid, block, sh, stream, (const float*)x, w0, b0, wd, bd, (bf16*)d1, (int)F, (int)T, (int)C, T2, F2, T4, F4, rpb);
    else hipLaunchKernelGGL((stage01_fwd_kernel<bf16>), grid, block, sh, stream, (const bf16*)x, w0, b0, wd, bd, (bf16*)d1, (int)F, (int)T, (int)C, T2, F2, T4, F4, rpb);
    SCONF_LAUNCH_OK("sconf_sub_stage01_fwd");
    return 0;
}

// Backward of the fused stage: parameter gradients of conv0 and of the first depthwise conv from dd1 (B,T4,F4,C) bf16;
// all four outputs ACCUMULATED (+=).  (The mel input needs no gradient.)
SCONF_API int sconf_sub_stage01_bwd(const void* dd1, const void* x, int x_dtype, const float* w0, const float* b0, const float* wd,
                                    float* dw0, float* db0, float* dwd, float* dbd, int64_t B, int64_t F, int64_t T, int64_t C,
                                    hipStream_t stream) {
    SCONF_REQUIRE(C % 4 == 0 && C / 4 <= 256, "sconf_sub_stage01_bwd: C must be a multiple of 4 and <= 1024");
    SCONF_REQUIRE(B <= 65535 && F <= 1024, "sconf_sub_stage01_bwd: B <= 65535 and F <= 1024");
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1), T4 = (T2 - 1) / 2 + 1, F4 = (F2 - 1) / 2 + 1;
    if (B * T4 * F4 == 0) return 0;
    if (sconf_stage01_bwd_mfma(dd1, x, x_dtype, w0, b0, wd, dw0, db0, dwd, dbd, B, F, T, C, stream)) { SCONF_LAUNCH_OK("sconf_sub_stage01_bwd"); return 0; }
    int rw = 4; long target = 512;                             // measured best of {1,2,4} x {256..2048} at config 3
    if (const char* e = getenv("SCONF_SUB_BWD_CFG")) { int a = 0; long t = 0; if (sscanf(e, "%d,%ld", &a, &t) == 2) { rw = a; target = t; } }   // tuning
    auto lds_bytes = [&](int r) { return (size_t)((2 * r + 1) * F + 256) * 4 + (size_t)(r / 2 + 2) * F4 * C * 2; };
    while (rw > 1 && lds_bytes(rw) > 64 * 1024) rw /= 2;       // wide stages (C = 512): fewer dd1 rows per staging step
    int rpb2 = std::max(1, (int)cdiv((long)T2 * B, target));
    rpb2 = (rpb2 + rw - 1) / rw * rw;                          // whole staging steps per workgroup
    dim3 g2(cdiv(T2, rpb2), (unsigned)B), block(256);
    const size_t sh = lds_bytes(rw);
    SCONF_REQUIRE(sh <= 64 * 1024 && (F4 * C) % 8 == 0, "sconf_sub_stage01_bwd: dd1 rows do not fit LDS");
#define LB(TX, RW_) hipLaunchKernelGGL((stage01_bwd_kernel<TX, RW_>), g2, block, sh, stream, (const TX*)x, w0, b0, wd, (const bf16*)dd1, dw0, db0, dwd, dbd, (int)F, (int)T, (int)C, T2, F2, T4, F4, rpb2)
    if (x_dtype == SCONF_F32) { if (rw == 1) LB(float, 1); else if (rw == 4) LB(float, 4); else LB(float, 2); }
    else                      { if (rw == 1) LB(bf16, 1); else if (rw == 4) LB(bf16, 4); else LB(bf16, 2); }
#undef LB
    SCONF_LAUNCH_OK("sconf_sub_stage01_bwd");
    return 0;
}
