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

template <typename TI, typename TG, typename TO, int MODE, int MAXIT>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const TG* __restrict__ dy, const TI* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ stat_mean,
                                                       const float* __restrict__ stat_rstd, const float* __restrict__ dres,
                                                       TO* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                       int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    float aw[MAXIT][4], ab[MAXIT][4], wv[MAXIT][4];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = it * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { aw[it][e] = 0.f; ab[it][e] = 0.f; wv[it][e] = 0.f; }
        if (c < d) load4(w + c, wv[it]);
    }
    for (int row = wid; row < M; row += nw) {
        const TI* xr = x + (long)row * d;
        const TG* gr = dy + (long)row * d;
        const float mean = (MODE == 0) ? stat_mean[row] : 0.f;
        const float rstd = stat_rstd[row];
        float xh[MAXIT][4], g[MAXIT][4];
        float s1 = 0.f, s2 = 0.f;                    // sum(g*w), sum(g*w*xhat)  (xhat = raw x for RMS modes)
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                load4(xr + c, xh[it]); load4(gr + c, g[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xn = (MODE == 0) ? (xh[it][e] - mean) * rstd : xh[it][e];
                    xh[it][e] = xn;
                    const float gw = g[it][e] * wv[it][e];
                    s1 += gw; s2 += gw * xn;
                    aw[it][e] += g[it][e] * ((MODE == 0) ? xn : xn * rstd);
                    ab[it][e] += g[it][e];
                }
            }
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        float c1, c2;                                 // dx = rstd*gw - c1 - xn*c2
        if (MODE == 0) { c1 = rstd * s1 / d; c2 = rstd * s2 / d; }
        else if (MODE == 1) {                         // y = w x / (rms+eps);  rstd = 1/(rms+eps)
            const float rms = 1.f / rstd - eps;
            c1 = 0.f; c2 = (rms > 0.f) ? s2 * rstd * rstd / (d * rms) : 0.f;
        } else { c1 = 0.f; c2 = s2 * rstd * rstd * rstd / d; }
        TO* dxr = dx + (long)row * d;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int c = it * 256 + lane * 4;
            if (c < d) {
                float o[4];
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                if (dres) load4(dres + (long)row * d + c, r);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = r[e] + rstd * g[it][e] * wv[it][e] - c1 - xh[it][e] * c2;
                store4(dxr + c, o);
            }
        }
    }
    // cross-wave reduction in LDS, then ONE atomic per column per workgroup (<= 512 workgroups): column atomics from
    // every wave of every workgroup serialise on the few hundred addresses of dw/db.
    __shared__ float red[4][256];
    const int wvi = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if (it * 256 >= d) break;                                   // uniform
        const int c = it * 256 + lane * 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && !(MODE == 0 && db)) break;             // uniform
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wvi][lane * 4 + e] = pass == 0 ? aw[it][e] : ab[it][e];
            __syncthreads();
            if (wvi == 0 && c < d) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = red[0][lane * 4 + e] + red[1][lane * 4 + e] + red[2][lane * 4 + e] + red[3][lane * 4 + e];
                    atomicAdd((pass == 0 ? dw : db) + c + e, v);
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

template <int MODE, int NIT>
int launch_bwd(const void* dy, int gdt, const void* x, int xdt, const float* w, const float* mean, const float* rstd,
               const float* dres, void* dx, int odt, float* dw, float* db, int M, int d, float eps, hipStream_t st) {
    dim3 grid(min(cdiv(M, 4), 512)), block(256);
#define L(TI, TG, TO) hipLaunchKernelGGL((norm_bwd_kernel<TI, TG, TO, MODE, NIT>), grid, block, 0, st, (const TG*)dy, (const TI*)x, w, mean, rstd, dres, (TO*)dx, dw, db, M, d, eps)
    if (xdt == SCONF_F32) {
        if (gdt == SCONF_F32) { if (odt == SCONF_F32) L(float, float, float); else L(float, float, bf16); }
        else                  { if (odt == SCONF_F32) L(float, bf16, float);  else L(float, bf16, bf16); }
    } else {
        if (gdt == SCONF_F32) { if (odt == SCONF_F32) L(bf16, float, float); else L(bf16, float, bf16); }
        else                  { if (odt == SCONF_F32) L(bf16, bf16, float);  else L(bf16, bf16, bf16); }
    }
#undef L
    return 0;
}

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

// dx = (dres ? dres : 0) + d(norm)/dx . dy ;  dweight/dbias are ACCUMULATED (+=) with f32 atomics.
SCONF_API int sconf_norm_bwd(int mode, const void* dy, int dy_dtype, const void* x, int x_dtype, const float* weight,
                             const float* mean, const float* rstd, const float* dres, void* dx, int dx_dtype,
                             float* dweight, float* dbias, int64_t M, int64_t d, float eps, hipStream_t stream) {
    SCONF_REQUIRE(mode >= 0 && mode <= 2, "sconf_norm_bwd: bad mode %d", mode);
    SCONF_REQUIRE(d % 4 == 0 && d <= MAXD && d > 0, "sconf_norm_bwd: d=%ld must be a multiple of 4 and <= 2048", (long)d);
    SCONF_REQUIRE(M < (1L << 31), "sconf_norm_bwd: too many rows");
    if (M == 0) return 0;
    if (mode == 0) NIT_DISPATCH(launch_bwd, 0, dy, dy_dtype, x, x_dtype, weight, mean, rstd, dres, dx, dx_dtype, dweight, dbias, (int)M, (int)d, eps, stream);
    else if (mode == 1) NIT_DISPATCH(launch_bwd, 1, dy, dy_dtype, x, x_dtype, weight, mean, rstd, dres, dx, dx_dtype, dweight, dbias, (int)M, (int)d, eps, stream);
    else NIT_DISPATCH(launch_bwd, 2, dy, dy_dtype, x, x_dtype, weight, mean, rstd, dres, dx, dx_dtype, dweight, dbias, (int)M, (int)d, eps, stream);
    SCONF_LAUNCH_OK("sconf_norm_bwd");
    return 0;
}
