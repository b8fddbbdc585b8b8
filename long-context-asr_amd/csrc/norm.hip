// Fused row normalisation, forward + backward (HBM-bound; one wave64 per row, row kept in registers).
//
// mode 0: LayerNorm  (torch.nn.LayerNorm / apex FusedLayerNorm, eps inside the sqrt; sconformer_xl.py:14-17)
// mode 1: RMSNorm    (lcasr/components/normalisation.py:6-47:  scale * x / (||x||_2 * d^-1/2 + eps))
// mode 2: RMSNorm    (apex FusedRMSNorm convention:            weight * x * rsqrt(mean(x^2) + eps))
//
// The backward optionally adds a residual-stream gradient (`dres`) so that the pre-norm pattern
//   x -> x + f(norm(x))     (wrappers.py:5-28, sconformer_xl.py:355-369)
// needs a single pass: dx = dres + norm_bwd(dy).  Parameter gradients are reduced per lane over a
// grid-stride loop of rows, reduced across the workgroup's waves in LDS and flushed with one f32 atomic per
// column per workgroup.
#include "common.h"
#include <stdlib.h>

SCONF_API int sconf_num_cus(void);

namespace {

constexpr int MAXD = 2048;                   // d <= 8 * 256; kernels are instantiated for NIT = ceil(d/256) in {1,2,3,4,8}
// so a d=768 row costs 3 (not 8) register iterations: occupancy, not bandwidth, was the limit at small d.

template <typename TI, typename TO, int MODE, int MAXIT>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, TO* __restrict__ y,
                                                       float* __restrict__ stat_mean, float* __restrict__ stat_rstd,
                                                       int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const TI* xr = x + (long)row * d;
    float v[MAXIT][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) { load4(xr + c, v[it]); s += v[it][0] + v[it][1] + v[it][2] + v[it][3]; }
        else { v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f; }
    }
    float mean = 0.f, rstd;
    if (MODE == 0) {
        mean = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float t = v[it][e] - mean; q += t * t; }
            }
        }
        rstd = rsqrtf(wave_sum(q) / d + eps);
    } else {
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) q += v[it][e] * v[it][e];
        q = wave_sum(q) / d;
        rstd = (MODE == 1) ? 1.f / (sqrtf(q) + eps) : rsqrtf(q + eps);
    }
    if (lane == 0) { if (stat_mean) stat_mean[row] = mean; if (stat_rstd) stat_rstd[row] = rstd; }
    TO* yr = y + (long)row * d;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            float wv[4], o[4]; load4(w + c, wv);
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 0 && b) load4(b + c, bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mean) * rstd * wv[e] + bv[e];
            store4(yr + c, o);
        }
    }
}

// raw 4-element vectors: a row's loads are issued one row ahead and converted only when consumed
template <typename T> struct Raw4;
template <> struct Raw4<float> { float4 v; __device__ __forceinline__ void get(float (&o)[4]) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; } };
template <> struct Raw4<bf16>  { bf16x4 v; __device__ __forceinline__ void get(float (&o)[4]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (float)v[e]; } };
template <typename T> __device__ __forceinline__ void raw_load(Raw4<T>& r, const T* p) { r.v = *reinterpret_cast<const decltype(r.v)*>(p); }

template <typename TI, typename TG, int MAXIT> struct BwdRow {
    Raw4<TI> x[MAXIT]; Raw4<TG> g[MAXIT]; Raw4<float> r[MAXIT]; float mean, rstd;
    template <int MODE> __device__ __forceinline__ void load(const TG* dy, const TI* xp, const float* stat_mean, const float* stat_rstd,
                                                             const float* dres, int row, int d, int lane, int coff = 0) {
        mean = (MODE == 0) ? stat_mean[row] : 0.f;
        rstd = stat_rstd[row];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = coff + it * 256 + lane * 4;
            if (c < d) {
                raw_load(x[it], xp + (long)row * d + c); raw_load(g[it], dy + (long)row * d + c);
                if (dres) raw_load(r[it], dres + (long)row * d + c);
            }
        }
    }
};

// One wave per row, rows strided over the grid; the NEXT row's loads (x, dy, the incoming residual gradient, the row
// statistics) are issued before the current row's reductions, so the HBM latency overlaps the dependent wave_sum chains.
// SLAB: the workgroup's column sums go to ws[blockIdx][dw | db][d] (plain stores; norm_bwd_reduce_kernel adds them up) instead
// of atomics: all workgroups finish together and 512 x 2d atomics on 2d addresses serialise at the end of the kernel.
// CS = 2 (rows wider than 1024): a PAIR of waves per row, each MAXIT * 256 columns of it - with one wave per 2048-wide row the four
// per-column accumulators alone are 128 registers and the kernel spilled 146-217 VGPRs (2.6 TB/s); the two row sums are exchanged
// through LDS (one barrier per row; every wave runs the same number of trips).
template <typename TI, typename TG, typename TO, int MODE, int MAXIT, int NB_WAVES, bool AHEAD, bool SLAB, int CS = 1>
__global__ __launch_bounds__(64 * NB_WAVES) void norm_bwd_kernel(const TG* __restrict__ dy, const TI* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ stat_mean,
                                                       const float* __restrict__ stat_rstd, const float* __restrict__ dres,
                                                       TO* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                       float* __restrict__ ws, bf16* __restrict__ dx16, int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int wv_ = threadIdx.x >> 6, half = CS == 2 ? (wv_ & 1) : 0, coff = half * (MAXIT * 256);
    const int wid = (blockIdx.x * NB_WAVES + wv_) / CS, nw = gridDim.x * NB_WAVES / CS;
    __shared__ float xs[2][NB_WAVES][2];               // CS == 2: the two partial row sums of a wave, double-buffered by trip
    // dx16 (SLAB only): a bf16 copy of dx for the GEMMs of the block that receives dx as its output gradient, and the column
    // sums of that copy (that block's output-projection bias gradient) - saves it a cast pass and a column-sum pass over dx.
    float aw[MAXIT][4], ab[MAXIT][4], wv[MAXIT][4], ac[MAXIT][4];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = coff + it * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { aw[it][e] = 0.f; ab[it][e] = 0.f; wv[it][e] = 0.f; ac[it][e] = 0.f; }
        if (c < d) load4(w + c, wv[it]);
    }
    BwdRow<TI, TG, MAXIT> nxt;
    if (AHEAD && wid < M) nxt.template load<MODE>(dy, x, stat_mean, stat_rstd, dres, wid, d, lane, coff);
    const int trips = CS == 2 ? (M + nw - 1) / nw : 0;  // CS == 2: uniform over the workgroup (a barrier per trip); rows past M are clamped and not stored
    for (int row = wid, tr = 0; CS == 2 ? tr < trips : row < M; row += nw, ++tr) {
        const bool live = row < M;
        const int rowc = CS == 2 ? min(row, M - 1) : row;
        BwdRow<TI, TG, MAXIT> cur;
        if (AHEAD) {
            cur = nxt;
            if (row + nw < M) nxt.template load<MODE>(dy, x, stat_mean, stat_rstd, dres, row + nw, d, lane, coff);
        } else cur.template load<MODE>(dy, x, stat_mean, stat_rstd, dres, rowc, d, lane, coff);
        const float mean = cur.mean, rstd = cur.rstd;
        float xh[MAXIT][4], g[MAXIT][4];
        float s1 = 0.f, s2 = 0.f;                    // sum(g*w), sum(g*w*xhat)  (xhat = raw x for RMS modes)
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = coff + it * 256 + lane * 4;
            if (c < d) {
                cur.x[it].get(xh[it]); cur.g[it].get(g[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xn = (MODE == 0) ? (xh[it][e] - mean) * rstd : xh[it][e];
                    xh[it][e] = xn;
                    const float gw = g[it][e] * wv[it][e];
                    s1 += gw; s2 += gw * xn;
                    if (CS == 1 || live) {
                        aw[it][e] += g[it][e] * ((MODE == 0) ? xn : xn * rstd);
                        ab[it][e] += g[it][e];
                    }
                }
            }
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (CS == 2) {                                  // the partner wave holds the other half of the row
            if (lane == 0) { xs[tr & 1][wv_][0] = s1; xs[tr & 1][wv_][1] = s2; }
            __syncthreads();
            s1 += xs[tr & 1][wv_ ^ 1][0]; s2 += xs[tr & 1][wv_ ^ 1][1];
        }
        float c1, c2;                                 // dx = rstd*gw - c1 - xn*c2
        if (MODE == 0) { c1 = rstd * s1 / d; c2 = rstd * s2 / d; }
        else if (MODE == 1) {                         // y = w x / (rms+eps);  rstd = 1/(rms+eps)
            const float rms = 1.f / rstd - eps;
            c1 = 0.f; c2 = (rms > 0.f) ? s2 * rstd * rstd / (d * rms) : 0.f;
        } else { c1 = 0.f; c2 = s2 * rstd * rstd * rstd / d; }
        TO* dxr = dx + (long)rowc * d;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = coff + it * 256 + lane * 4;
            if (c < d && (CS == 1 || live)) {
                float o[4];
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                if (dres) cur.r[it].get(r);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = r[e] + rstd * g[it][e] * wv[it][e] - c1 - xh[it][e] * c2;
                store4(dxr + c, o);
                if (SLAB && dx16) {
                    store4(dx16 + (long)rowc * d + c, o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ac[it][e] += (float)(bf16)o[e];
                }
            }
        }
    }
    // cross-wave reduction in LDS, then ONE atomic per column per workgroup (<= 512 workgroups): column atomics from
    // every wave of every workgroup serialise on the few hundred addresses of dw/db.
    __shared__ float red[NB_WAVES][256];
    const int wvi = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if (it * 256 >= d) break;                                   // uniform
        const int c = coff + it * 256 + lane * 4;
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            if (pass == 1 && !(MODE == 0 && db)) continue;          // uniform
            if (pass == 2 && !(SLAB && dx16)) continue;             // uniform
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wvi][lane * 4 + e] = pass == 0 ? aw[it][e] : (pass == 1 ? ab[it][e] : ac[it][e]);
            __syncthreads();
            if (wvi < CS && c < d) {                                // wave 0 (and, CS == 2, wave 1 for the upper half of the columns)
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = 0.f;
#pragma unroll
                    for (int k = wvi; k < NB_WAVES; k += CS) v[e] += red[k][lane * 4 + e];
                }
                if (SLAB) store4(ws + ((long)blockIdx.x * 3 + pass) * d + c, v);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd((pass == 0 ? dw : db) + c + e, v[e]);
                }
            }
        }
    }
}

template <int MODE, int NIT>
int launch_fwd(const void* x, int xdt, const float* w, const float* b, void* y, int ydt, float* mean, float* rstd,
               int M, int d, float eps, hipStream_t st) {
    dim3 grid(cdiv(M, 4)), block(256);
#define L(TI, TO) hipLaunchKernelGGL((norm_fwd_kernel<TI, TO, MODE, NIT>), grid, block, 0, st, (const TI*)x, w, b, (TO*)y, mean, rstd, M, d, eps)
    if (xdt == SCONF_F32 && ydt == SCONF_F32) L(float, float);
    else if (xdt == SCONF_F32 && ydt == SCONF_BF16) L(float, bf16);
    else if (xdt == SCONF_BF16 && ydt == SCONF_BF16) L(bf16, bf16);
    else L(bf16, float);
#undef L
    return 0;
}

// dw[c] += sum_b ws[b][0][c], db[c] += sum_b ws[b][1][c], cs[c] = sum_b ws[b][2][c]: one lane per column, 16 waves stride over
// the workgroups' slabs (fixed summation order)
__global__ __launch_bounds__(1024) void norm_bwd_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db,
                                                               float* __restrict__ cs, int nblocks, int d) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;                          // in [0, 3d): dw columns, db columns, cs columns
    const int pass = col / d, c = col - pass * d;
    float* const outp = pass == 0 ? dw : (pass == 1 ? db : cs);
    const bool live = col < 3 * d && outp != nullptr;
    float a = 0.f;
    if (live) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        int b = wv;
        for (; b + 48 < nblocks; b += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] += ws[((long)(b + 16 * u) * 3 + pass) * d + c];
        }
        for (; b < nblocks; b += 16) t[0] += ws[((long)b * 3 + pass) * d + c];
        a = (t[0] + t[1]) + (t[2] + t[3]);
    }
    red[wv][lane] = a;
    __syncthreads();
    if (wv == 0 && live) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][lane];
        if (pass == 2) outp[c] = v; else outp[c] += v;
    }
}

// Launch geometry of the backward: 8 waves per workgroup, <= one workgroup per CU (each wave walks rows wid, wid + nw, ...); the next
// row's loads are issued a row ahead when the row fits the register file (d <= 1024).  Both were run-time template choices in round 1
// (x 4 instantiations of every (mode, width, dtype) combination: a 10 MB object for a LayerNorm); the measured-best pair is now fixed.
constexpr int NBW = 8;
static int bwd_maxgrid() {
    static int cus = 0;
    if (!cus) { const int n = sconf_num_cus(); cus = n > 0 ? n : 256; }
    if (const char* e = getenv("SCONF_NORM_BWD_GRID")) { const int v = atoi(e); if (v > 0) return v; }      // benchmarking
    return cus;
}

template <int MODE, int NIT>
int launch_bwd(const void* dy, int gdt, const void* x, int xdt, const float* w, const float* mean, const float* rstd,
               const float* dres, void* dx, int odt, float* dw, float* db, float* ws, long ws_floats, bf16* dx16, float* dx_colsum,
               int M, int d, float eps, hipStream_t st) {
    constexpr int CS = NIT > 4 ? 2 : 1, WIT = NIT / CS;          // rows wider than 1024: a pair of waves per row (WIT * 256 columns each)
    static_assert(NIT <= 4 || NIT % 2 == 0, "wide rows are split in two equal column halves");
    constexpr bool AHEAD = true;
    dim3 grid(min(cdiv(M * CS, NBW), bwd_maxgrid())), block(64 * NBW);
    const bool slab = ws && ws_floats >= (long)grid.x * 3 * d;
    if (!slab) { dx16 = nullptr; dx_colsum = nullptr; }
#define L4(TI, TG, TO, SL) hipLaunchKernelGGL((norm_bwd_kernel<TI, TG, TO, MODE, WIT, NBW, AHEAD, SL, CS>), grid, block, 0, st, (const TG*)dy, (const TI*)x, w, mean, rstd, dres, (TO*)dx, dw, db, ws, dx16, M, d, eps)
#define L(TI, TG, TO) do { if (slab) L4(TI, TG, TO, true); else L4(TI, TG, TO, false); } while (0)
    if (xdt == SCONF_F32) {
        if (gdt == SCONF_F32) { if (odt == SCONF_F32) L(float, float, float); else L(float, float, bf16); }
        else                  { if (odt == SCONF_F32) L(float, bf16, float);  else L(float, bf16, bf16); }
    } else {
        if (gdt == SCONF_F32) { if (odt == SCONF_F32) L(bf16, float, float); else L(bf16, float, bf16); }
        else                  { if (odt == SCONF_F32) L(bf16, bf16, float);  else L(bf16, bf16, bf16); }
    }
#undef L
#undef L4
    if (slab) hipLaunchKernelGGL(norm_bwd_reduce_kernel, dim3(cdiv(3 * d, 64)), dim3(1024), 0, st, ws, dw, (MODE == 0 ? db : nullptr),
                                 dx16 ? dx_colsum : nullptr, (int)grid.x, d);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two LayerNorms back to back, y1 = LN(x; w1, b1) (f32, kept: it is the layer's output and the residual of what follows) and
// h2 = LN(y1; w2, b2) (bf16, the operand of the next projection): `norm_out` of a ConformerLayer followed by the decoder norm of the
// self-conditioning step or of the head (sconformer_xl.py:371, 241-247; decoder.py:23).  Both kernels are HBM-bound and the two
// single norms run at the HBM roofline already, so what the fusion buys is bytes: forward 10 instead of 14 B per element (y1 is
// not re-read), backward 16 instead of 28 (the gradient of y1 never exists in memory, and y1 itself is recomputed from x and the
// saved row statistics instead of being read).
// TWICE: the second norm is applied two times, h = LN(LN(y1; w2, b2); w2, b2) - the reference's legacy double norm of the head
// (sconformer_xl.py:246-247 with legasee_double_norm) after the last layer's norm_out; the value in between stays in registers.
template <int MAXIT, bool TWICE>
__global__ __launch_bounds__(256) void norm2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ y1, bf16* __restrict__ h2, float* __restrict__ mean1,
                                                        float* __restrict__ rstd1, float* __restrict__ mean2, float* __restrict__ rstd2,
                                                        float* __restrict__ mean3, float* __restrict__ rstd3,
                                                        int M, int d, float eps1, float eps2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long)row * d;
    float v[MAXIT][4];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) { load4(xr + c, v[it]); s += v[it][0] + v[it][1] + v[it][2] + v[it][3]; }
        else { v[it][0] = v[it][1] = v[it][2] = v[it][3] = 0.f; }
    }
    // the same two-pass statistics as norm_fwd_kernel
    auto stats = [&](float sum, float eps, float& mean, float& rstd) {
        mean = wave_sum(sum) / d;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float t = v[it][e] - mean; q += t * t; }
            }
        }
        rstd = rsqrtf(wave_sum(q) / d + eps);
    };
    float m1, r1, m2, r2;
    stats(s, eps1, m1, r1);
    float* yr = y1 + (long)row * d;
    s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            float wv[4], bv[4]; load4(w1 + c, wv); load4(b1 + c, bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[it][e] = (v[it][e] - m1) * r1 * wv[e] + bv[e];
            store4(yr + c, v[it]);
            s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
        }
    }
    stats(s, eps2, m2, r2);
    if (lane == 0) { mean1[row] = m1; rstd1[row] = r1; mean2[row] = m2; rstd2[row] = r2; }
    if (TWICE) {
        s = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                float wv[4], bv[4]; load4(w2 + c, wv); load4(b2 + c, bv);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[it][e] = (v[it][e] - m2) * r2 * wv[e] + bv[e];
                s += v[it][0] + v[it][1] + v[it][2] + v[it][3];
            }
        }
        stats(s, eps2, m2, r2);                           // (m2, r2 now hold the statistics of the third norm's input)
        if (lane == 0) { mean3[row] = m2; rstd3[row] = r2; }
    }
    bf16* hr = h2 + (long)row * d;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
        if (c < d) {
            float wv[4], bv[4], o[4]; load4(w2 + c, wv); load4(b2 + c, bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - m2) * r2 * wv[e] + bv[e];
            store4(hr + c, o);
        }
    }
}

// Backward of the pair.  dh2: gradient of h2 (bf16); dres: gradient that reaches y1 directly (f32, nullable).  Per row:
//   y1 = xhat1 w1 + b1 (recomputed), xhat2 = (y1 - mean2) rstd2,  dy1 = dres + LN'(dh2 at y1),  dx = LN'(dy1 at x).
// TWICE: one more LN'(. at y2 = xhat2 w2 + b2) in front, with the same w2 (its gradient collects both applications).
// Parameter gradients of both norms and the column sums of the bf16 twin go to the workgroup's slab [5][d]
// (dw1 | db1 | dw2 | db2 | colsum), added up in a fixed order by norm2_bwd_reduce_kernel.
template <int MAXIT, int NB_WAVES, bool TWICE>
__global__ __launch_bounds__(64 * NB_WAVES) void norm2_bwd_kernel(const bf16* __restrict__ dh2, const float* __restrict__ x,
                                                          const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2,
                                                          const float* __restrict__ mean1, const float* __restrict__ rstd1,
                                                          const float* __restrict__ mean2, const float* __restrict__ rstd2,
                                                          const float* __restrict__ mean3, const float* __restrict__ rstd3,
                                                          const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ ws,
                                                          bf16* __restrict__ dx16, int M, int d) {
    constexpr bool AHEAD = !TWICE;                          // the three-norm form has no registers left for the row-ahead loads (one launch per step)
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * NB_WAVES + (threadIdx.x >> 6), nw = gridDim.x * NB_WAVES;
    float aw1[MAXIT][4], ab1[MAXIT][4], aw2[MAXIT][4], ab2[MAXIT][4], ac[MAXIT][4], w1v[MAXIT][4], w2v[MAXIT][4];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { aw1[it][e] = ab1[it][e] = aw2[it][e] = ab2[it][e] = ac[it][e] = 0.f; w1v[it][e] = w2v[it][e] = 0.f; }
        if (c < d) { load4(w1 + c, w1v[it]); load4(w2 + c, w2v[it]); }
    }
    struct Row { Raw4<float> x[MAXIT]; Raw4<bf16> g[MAXIT]; Raw4<float> r[MAXIT]; float m1, r1, m2, r2, m3, r3; };
    auto load_row = [&](Row& R, int row) {
        R.m1 = mean1[row]; R.r1 = rstd1[row]; R.m2 = mean2[row]; R.r2 = rstd2[row];
        if (TWICE) { R.m3 = mean3[row]; R.r3 = rstd3[row]; }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                raw_load(R.x[it], x + (long)row * d + c); raw_load(R.g[it], dh2 + (long)row * d + c);
                if (dres) raw_load(R.r[it], dres + (long)row * d + c);
            }
        }
    };
    Row nxt;
    if (AHEAD && wid < M) load_row(nxt, wid);
    for (int row = wid; row < M; row += nw) {
        Row cur;
        if (AHEAD) { cur = nxt; if (row + nw < M) load_row(nxt, row + nw); }
        else load_row(cur, row);
        float xh1[MAXIT][4], xh2[TWICE ? MAXIT : 1][4], t[MAXIT][4];     // t: the innermost xhat, then the gradient flowing outwards
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                float xv[4], g[4], bv[4];
                cur.x[it].get(xv); cur.g[it].get(g); load4(b1 + c, bv);
                float b2v[4] = {0.f, 0.f, 0.f, 0.f};
                if (TWICE) load4(b2 + c, b2v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x1 = (xv[e] - cur.m1) * cur.r1;
                    xh1[it][e] = x1;
                    float xin = (x1 * w1v[it][e] + bv[e] - cur.m2) * cur.r2;              // xhat2
                    if (TWICE) { xh2[it][e] = xin; xin = (xin * w2v[it][e] + b2v[e] - cur.m3) * cur.r3; }   // xhat3
                    const float gw = g[e] * w2v[it][e];
                    s1 += gw; s2 += gw * xin;
                    aw2[it][e] += g[e] * xin; ab2[it][e] += g[e];
                    t[it][e] = xin;
                }
            }
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        const float rin = TWICE ? cur.r3 : cur.r2;
        float c1 = rin * s1 / d, c2 = rin * s2 / d;
        if (TWICE) {                                        // through the inner application of (w2, b2): dy2, then the same sums with xhat2
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int c = it * 256 + lane * 4;
                if (c < d) {
                    float g[4];
                    cur.g[it].get(g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dy2 = cur.r3 * g[e] * w2v[it][e] - c1 - t[it][e] * c2;
                        t[it][e] = dy2;
                        const float gw = dy2 * w2v[it][e];
                        p1 += gw; p2 += gw * xh2[it][e];
                        aw2[it][e] += dy2 * xh2[it][e]; ab2[it][e] += dy2;
                    }
                }
            }
            p1 = wave_sum(p1); p2 = wave_sum(p2);
            c1 = cur.r2 * p1 / d; c2 = cur.r2 * p2 / d;
        }
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                float g[4], r[4] = {0.f, 0.f, 0.f, 0.f};
                if (!TWICE) cur.g[it].get(g);
                if (dres) cur.r[it].get(r);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // gradient entering the (outer) application of (w2, b2) at y1, and xhat2
                    const float gin = TWICE ? t[it][e] : g[e], x2 = TWICE ? xh2[it][e] : t[it][e];
                    const float dy1 = r[e] + cur.r2 * gin * w2v[it][e] - c1 - x2 * c2;
                    t[it][e] = dy1;
                    const float gw = dy1 * w1v[it][e];
                    u1 += gw; u2 += gw * xh1[it][e];
                    aw1[it][e] += dy1 * xh1[it][e]; ab1[it][e] += dy1;
                }
            }
        }
        u1 = wave_sum(u1); u2 = wave_sum(u2);
        const float e1 = cur.r1 * u1 / d, e2 = cur.r1 * u2 / d;
        float* dxr = dx + (long)row * d;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = cur.r1 * t[it][e] * w1v[it][e] - e1 - xh1[it][e] * e2;
                store4(dxr + c, o);
                if (dx16) {
                    store4(dx16 + (long)row * d + c, o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ac[it][e] += (float)(bf16)o[e];
                }
            }
        }
    }
    __shared__ float red[NB_WAVES][256];
    const int wvi = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if (it * 256 >= d) break;                                   // uniform
        const int c = it * 256 + lane * 4;
#pragma unroll
        for (int pass = 0; pass < 5; ++pass) {
            if (pass == 4 && !dx16) continue;                       // uniform
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e)
                red[wvi][lane * 4 + e] = pass == 0 ? aw1[it][e] : pass == 1 ? ab1[it][e] : pass == 2 ? aw2[it][e] : pass == 3 ? ab2[it][e] : ac[it][e];
            __syncthreads();
            if (wvi == 0 && c < d) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = 0.f;
#pragma unroll
                    for (int k = 0; k < NB_WAVES; ++k) v[e] += red[k][lane * 4 + e];
                }
                store4(ws + ((long)blockIdx.x * 5 + pass) * d + c, v);
            }
        }
    }
}

// outs[pass][c] (+)= sum_b ws[b][pass][c] for the five slabs of norm2_bwd_kernel (the colsum slab overwrites, the rest accumulate)
struct Norm2Outs { float* p[5]; };
__global__ __launch_bounds__(1024) void norm2_bwd_reduce_kernel(const float* __restrict__ ws, Norm2Outs outs, int nblocks, int d) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;                          // in [0, 5d)
    const int pass = col / d, c = col - pass * d;
    float* const outp = pass < 5 ? outs.p[pass] : nullptr;
    const bool live = col < 5 * d && outp != nullptr;
    float a = 0.f;
    if (live) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        int b = wv;
        for (; b + 48 < nblocks; b += 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] += ws[((long)(b + 16 * u) * 5 + pass) * d + c];
        }
        for (; b < nblocks; b += 16) t[0] += ws[((long)b * 5 + pass) * d + c];
        a = (t[0] + t[1]) + (t[2] + t[3]);
    }
    red[wv][lane] = a;
    __syncthreads();
    if (wv == 0 && live) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][lane];
        if (pass == 4) outp[c] = v; else outp[c] += v;
    }
}

template <int NIT> int launch_norm2_fwd(bool twice, const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y1, bf16* h2,
                                        float* mean1, float* rstd1, float* mean2, float* rstd2, float* mean3, float* rstd3, int M, int d, float eps1, float eps2, hipStream_t st) {
    if (twice) hipLaunchKernelGGL((norm2_fwd_kernel<NIT, true>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, w1, b1, w2, b2, y1, h2, mean1, rstd1, mean2, rstd2, mean3, rstd3, M, d, eps1, eps2);
    else hipLaunchKernelGGL((norm2_fwd_kernel<NIT, false>), dim3(cdiv(M, 4)), dim3(256), 0, st, x, w1, b1, w2, b2, y1, h2, mean1, rstd1, mean2, rstd2, mean3, rstd3, M, d, eps1, eps2);
    return 0;
}
template <int NIT> int launch_norm2_bwd(bool twice, const bf16* dh2, const float* x, const float* w1, const float* b1, const float* w2, const float* b2, const float* mean1,
                                        const float* rstd1, const float* mean2, const float* rstd2, const float* mean3, const float* rstd3, const float* dres, float* dx, float* ws,
                                        bf16* dx16, int grid, int M, int d, hipStream_t st) {
    if (twice) hipLaunchKernelGGL((norm2_bwd_kernel<NIT, NBW, true>), dim3(grid), dim3(64 * NBW), 0, st, dh2, x, w1, b1, w2, b2, mean1, rstd1, mean2, rstd2, mean3, rstd3, dres, dx, ws, dx16, M, d);
    else hipLaunchKernelGGL((norm2_bwd_kernel<NIT, NBW, false>), dim3(grid), dim3(64 * NBW), 0, st, dh2, x, w1, b1, w2, b2, mean1, rstd1, mean2, rstd2, mean3, rstd3, dres, dx, ws, dx16, M, d);
    return 0;
}
#define NIT_DISPATCH1(FN, ...) do { const int nit_ = (int)((d + 255) / 256); \
    if (nit_ <= 1) FN<1>(__VA_ARGS__); else if (nit_ <= 2) FN<2>(__VA_ARGS__); else FN<3>(__VA_ARGS__); } while (0)   /* d <= 768: wider rows spill in the backward */

#define NIT_DISPATCH(FN, MODE_, ...) do { const int nit_ = (int)((d + 255) / 256); \
    if (nit_ <= 1) FN<MODE_, 1>(__VA_ARGS__); else if (nit_ <= 2) FN<MODE_, 2>(__VA_ARGS__); else if (nit_ <= 3) FN<MODE_, 3>(__VA_ARGS__); \
    else if (nit_ <= 4) FN<MODE_, 4>(__VA_ARGS__); else FN<MODE_, 8>(__VA_ARGS__); } while (0)

}  // namespace

// Replaces torch.nn.LayerNorm / apex FusedLayerNorm / RMSNorm forward (sconformer_xl.py:14-17, normalisation.py:34-47).
SCONF_API int sconf_norm_fwd(int mode, const void* x, int x_dtype, const float* weight, const float* bias,
                             void* y, int y_dtype, float* mean, float* rstd, int64_t M, int64_t d, float eps,
                             hipStream_t stream) {
    SCONF_REQUIRE(mode >= 0 && mode <= 2, "sconf_norm_fwd: bad mode %d", mode);
    SCONF_REQUIRE(d % 4 == 0 && d <= MAXD && d > 0, "sconf_norm_fwd: d=%ld must be a multiple of 4 and <= 2048", (long)d);
    SCONF_REQUIRE(M < (1L << 31), "sconf_norm_fwd: too many rows");
    if (M == 0) return 0;
    if (mode == 0) NIT_DISPATCH(launch_fwd, 0, x, x_dtype, weight, bias, y, y_dtype, mean, rstd, (int)M, (int)d, eps, stream);
    else if (mode == 1) NIT_DISPATCH(launch_fwd, 1, x, x_dtype, weight, bias, y, y_dtype, mean, rstd, (int)M, (int)d, eps, stream);
    else NIT_DISPATCH(launch_fwd, 2, x, x_dtype, weight, bias, y, y_dtype, mean, rstd, (int)M, (int)d, eps, stream);
    SCONF_LAUNCH_OK("sconf_norm_fwd");
    return 0;
}

// dx = (dres ? dres : 0) + d(norm)/dx . dy ;  dweight/dbias are ACCUMULATED (+=).  `workspace` (optional, >=
// sconf_norm_bwd_workspace(M, d) floats, contents irrelevant) receives per-workgroup column sums that a second kernel adds
// into dweight/dbias in a fixed order; without it the workgroups fall back to f32 atomics (slower, order not fixed).
SCONF_API int64_t sconf_norm_bwd_workspace(int64_t M, int64_t d) {
    return (int64_t)min(cdiv(M * (d > 1024 ? 2 : 1), NBW), bwd_maxgrid()) * 3 * d;        // (rows wider than 1024: two waves per row)
}
SCONF_API int sconf_norm_bwd(int mode, const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight,
                             const float* mean, const float* rstd, const float* dres, void* dx, int dx_dtype,
                             float* dweight, float* dbias, float* workspace, int64_t workspace_floats,
                             void* dx_bf16, float* dx_colsum, int64_t M, int64_t d, float eps, hipStream_t stream) {
    SCONF_REQUIRE(mode >= 0 && mode <= 2, "sconf_norm_bwd: bad mode %d", mode);
    SCONF_REQUIRE(d % 4 == 0 && d <= MAXD && d > 0, "sconf_norm_bwd: d=%ld must be a multiple of 4 and <= 2048", (long)d);
    SCONF_REQUIRE(M < (1L << 31), "sconf_norm_bwd: too many rows");
    SCONF_REQUIRE(!dx_bf16 || (dx_dtype == SCONF_F32 && dx_colsum && workspace), "sconf_norm_bwd: the bf16 twin needs an f32 dx, dx_colsum and a workspace");
    if (M == 0) return 0;
    if (mode == 0) NIT_DISPATCH(launch_bwd, 0, dy, dy_dtype, x, x_dtype, weight, mean, rstd, dres, dx, dx_dtype, dweight, dbias, workspace, (long)workspace_floats, (bf16*)dx_bf16, dx_colsum, (int)M, (int)d, eps, stream);
    else if (mode == 1) NIT_DISPATCH(launch_bwd, 1, dy, dy_dtype, x, x_dtype, weight, mean, rstd, dres, dx, dx_dtype, dweight, dbias, workspace, (long)workspace_floats, (bf16*)dx_bf16, dx_colsum, (int)M, (int)d, eps, stream);
    else NIT_DISPATCH(launch_bwd, 2, dy, dy_dtype, x, x_dtype, weight, mean, rstd, dres, dx, dx_dtype, dweight, dbias, workspace, (long)workspace_floats, (bf16*)dx_bf16, dx_colsum, (int)M, (int)d, eps, stream);
    SCONF_LAUNCH_OK("sconf_norm_bwd");
    return 0;
}

// y1 = LayerNorm(x; w1, b1) (f32) and h2 = LayerNorm(y1; w2, b2) (bf16) in one pass over x; the row statistics are saved for
// sconf_norm2_bwd.  Replaces `norm_out` + the decoder norm that follows it (sconformer_xl.py:371 then 241-247 / decoder.py:23).
// twice != 0: h2 = LN(LN(y1; w2, b2); w2, b2), the legacy double norm of the head (mean3 / rstd3: statistics of the inner result).
SCONF_API int sconf_norm2_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y1, void* h2_bf16,
                              float* mean1, float* rstd1, float* mean2, float* rstd2, float* mean3, float* rstd3, int twice,
                              int64_t M, int64_t d, float eps1, float eps2, hipStream_t stream) {
    SCONF_REQUIRE(d % 4 == 0 && d <= 768 && d > 0, "sconf_norm2_fwd: d=%ld must be a multiple of 4 and <= 768", (long)d);
    SCONF_REQUIRE(M < (1L << 31), "sconf_norm2_fwd: too many rows");
    SCONF_REQUIRE(!twice || (mean3 && rstd3), "sconf_norm2_fwd: twice needs mean3 / rstd3");
    if (M == 0) return 0;
    NIT_DISPATCH1(launch_norm2_fwd, twice != 0, x, w1, b1, w2, b2, y1, (bf16*)h2_bf16, mean1, rstd1, mean2, rstd2, mean3, rstd3, (int)M, (int)d, eps1, eps2, stream);
    SCONF_LAUNCH_OK("sconf_norm2_fwd");
    return 0;
}

// dx = d(LN1)/dx . (dres + d(LN2)/dy1 . dh2)  [twice: one more d(LN2) in front];  dw1/db1/dw2/db2 ACCUMULATED;  dx_bf16 / dx_colsum (both or
// neither): a bf16 copy of dx and its column sums (overwritten).  workspace: sconf_norm2_bwd_workspace(M, d) floats (required).
SCONF_API int64_t sconf_norm2_bwd_workspace(int64_t M, int64_t d) { return (int64_t)min(cdiv(M, NBW), bwd_maxgrid()) * 5 * d; }
SCONF_API int sconf_norm2_bwd(const void* dh2_bf16, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                              const float* mean1, const float* rstd1, const float* mean2, const float* rstd2,
                              const float* mean3, const float* rstd3, int twice, const float* dres,
                              float* dx, float* dw1, float* db1, float* dw2, float* db2, float* workspace, int64_t workspace_floats,
                              void* dx_bf16, float* dx_colsum, int64_t M, int64_t d, hipStream_t stream) {
    SCONF_REQUIRE(d % 4 == 0 && d <= 768 && d > 0, "sconf_norm2_bwd: d=%ld must be a multiple of 4 and <= 768", (long)d);
    SCONF_REQUIRE(M < (1L << 31), "sconf_norm2_bwd: too many rows");
    SCONF_REQUIRE((dx_bf16 == nullptr) == (dx_colsum == nullptr), "sconf_norm2_bwd: dx_bf16 and dx_colsum go together");
    SCONF_REQUIRE(!twice || (mean3 && rstd3 && b2), "sconf_norm2_bwd: twice needs mean3 / rstd3 / b2");
    if (M == 0) return 0;
    const int grid = min(cdiv(M, NBW), bwd_maxgrid());
    SCONF_REQUIRE(workspace && workspace_floats >= (int64_t)grid * 5 * d, "sconf_norm2_bwd: workspace of sconf_norm2_bwd_workspace(M, d) floats required");
    NIT_DISPATCH1(launch_norm2_bwd, twice != 0, (const bf16*)dh2_bf16, x, w1, b1, w2, b2, mean1, rstd1, mean2, rstd2, mean3, rstd3, dres, dx, workspace, (bf16*)dx_bf16, grid, (int)M, (int)d, stream);
    Norm2Outs outs = {{dw1, db1, dw2, db2, dx_colsum}};
    hipLaunchKernelGGL(norm2_bwd_reduce_kernel, dim3(cdiv(5 * d, 64)), dim3(1024), 0, stream, workspace, outs, grid, (int)d);
    SCONF_LAUNCH_OK("sconf_norm2_bwd");
    return 0;
}
