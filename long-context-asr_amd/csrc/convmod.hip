// Conformer convolution module, token-major (B, N, C) — no (B,C,N) transposes as in the reference.
//
//   pw1 (GEMM) -> [GLU -> pad-mask -> depthwise conv k -> batch statistics]   glu_dwconv_fwd
//              -> [BatchRenorm finalize: r, d, running-stat EMA]              brn_finalize
//              -> [affine + SiLU]                                             affine_silu_fwd
//              -> pw2 (GEMM)
// Reference: ConformerConvolution.forward convolution.py:103-124, BatchRenorm.forward
// batchrenorm.py:52-92 (batch statistics over ALL B*N positions, std + eps, r/d detached).
//
// All kernels are HBM-bound: one thread owns 4 adjacent channels (8-B bf16 accesses, coalesced
// across the wave) and slides a register window along time, so every activation byte is read once.
// Per-channel reductions (batch statistics, parameter gradients) are accumulated in registers over
// a time slab and flushed with one atomic per channel per thread (f64 for the statistics).
#include "common.h"

namespace {

constexpr int CV = 4;                               // channels per thread

// Workgroup geometry of the sliding-window kernels: 256 threads = CPB channel groups x SPB time slabs, blockIdx.y picks
// the block of CPB channel groups.  Keeping many slabs of the SAME channels in one workgroup lets per-channel sums be
// reduced through LDS and flushed as ONE well-shaped atomic per channel per workgroup (64 lanes -> 64 consecutive
// floats): per-thread atomics on a few thousand addresses, 64 lanes in 64 different lines, ran at ~0.08 TB/s and
// dominated these kernels.
struct SlabGeo {
    int cg, sl, cpb, spb;
    bool active;
    __device__ __forceinline__ SlabGeo(int CG) {
        cpb = min(64, CG); spb = 256 / cpb;
        const int cgl = threadIdx.x % cpb;
        sl = threadIdx.x / cpb;
        cg = blockIdx.y * cpb + cgl;
        active = cg < CG && sl < spb;
    }
};
// sum v[CV] (this thread's channels) over the workgroup's slabs; thread (sl==0) ends up owning nothing special: instead
// lane i of the first cpb*CV threads adds channel (blockIdx.y*cpb*CV + i).  `sh` needs 1024 floats.
template <typename AT>
__device__ __forceinline__ void slab_reduce_add(const float (&v)[CV], const SlabGeo& g, float* sh, AT* out, int d) {
    __syncthreads();
    if (threadIdx.x < g.cpb * g.spb) {
#pragma unroll
        for (int e = 0; e < CV; ++e) sh[g.sl * (g.cpb * CV) + (threadIdx.x % g.cpb) * CV + e] = g.active ? v[e] : 0.f;
    }
    __syncthreads();
    const int nch = g.cpb * CV;
    if ((int)threadIdx.x < nch) {
        const int c = blockIdx.y * nch + threadIdx.x;
        if (c < d) {
            float a = 0.f;
            for (int s_ = 0; s_ < g.spb; ++s_) a += sh[s_ * nch + threadIdx.x];
            atomicAdd(out + c, (AT)a);
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void glu_dwconv_fwd_kernel(const bf16* __restrict__ g, const int* __restrict__ len,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             bf16* __restrict__ h, double* __restrict__ stats,
                                                             int B, int N, int d, int TN) {
    constexpr int P = (K - 1) / 2;
    __shared__ float red[1024];
    const int CG = d / CV, spb = (N + TN - 1) / TN;
    const SlabGeo geo(CG);
    const long slab = (long)blockIdx.x * geo.spb + geo.sl;
    const bool live = geo.active && slab < (long)B * spb;
    const int b = live ? (int)(slab / spb) : 0, n0 = live ? (int)(slab % spb) * TN : 0;
    const int c0 = live ? geo.cg * CV : 0;
    const int L = len ? len[b] : N;
    float wk[K][CV], win[K][CV], bs[CV];
#pragma unroll
    for (int e = 0; e < CV; ++e) {
        bs[e] = bias[c0 + e];
#pragma unroll
        for (int j = 0; j < K; ++j) { wk[j][e] = w[(c0 + e) * K + j]; win[j][e] = 0.f; }
    }
    float s1[CV] = {0.f, 0.f, 0.f, 0.f}, s2[CV] = {0.f, 0.f, 0.f, 0.f};
    const bf16* gb = g + (long)b * N * 2 * d;
    const int nend = live ? min(N, n0 + TN) : n0 - 2 * P;        // dead threads run an empty loop, then join the reductions
    const bf16x4 z4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    auto fetch = [&](int t, bf16x4& va, bf16x4& ga) {
        va = z4; ga = z4;
        if (t >= 0 && t < N && t < L) {
            va = *reinterpret_cast<const bf16x4*>(gb + (long)t * 2 * d + c0);
            ga = *reinterpret_cast<const bf16x4*>(gb + (long)t * 2 * d + d + c0);
        }
    };
    bf16x4 cva, cga;
    fetch(n0 - P, cva, cga);
    for (int t = n0 - P; t < nend + P; ++t) {
        bf16x4 nva, nga;
        fetch(t + 1 < nend + P ? t + 1 : -1000000, nva, nga);          // next step's loads fly during this step
#pragma unroll
        for (int j = 0; j < K - 1; ++j)
#pragma unroll
            for (int e = 0; e < CV; ++e) win[j][e] = win[j + 1][e];
#pragma unroll
        for (int e = 0; e < CV; ++e) win[K - 1][e] = (float)cva[e] * sigmoidf_((float)cga[e]);   // zeros outside [0, min(N,L))
        const int n = t - P;                         // window now holds a[n-P .. n+P]
        if (n >= n0 && n < nend) {
            float o[CV];
#pragma unroll
            for (int e = 0; e < CV; ++e) {
                float acc = bs[e];
#pragma unroll
                for (int j = 0; j < K; ++j) acc += wk[j][e] * win[j][e];
                // statistics are taken from the bf16-ROUNDED value that is stored, so that the normalisation the
                // next kernel applies to the stored tensor is exactly centred (dw-conv bias gradient == 0).
                const float rr = (float)(bf16)acc;
                o[e] = rr; s1[e] += rr; s2[e] += rr * rr;
            }
            store4(h + ((long)b * N + n) * d + c0, o);
        }
        cva = nva; cga = nga;
    }
    if (stats) {
        slab_reduce_add<double>(s1, geo, red, stats, d);
        slab_reduce_add<double>(s2, geo, red, stats + d, d);
    }
}

// coef rows: 0 mean, 1 s(=sigma+eps), 2 r, 3 dshift, 4 A, 5 Bc   (y = silu(h*A + Bc))
__global__ void brn_finalize_kernel(const double* __restrict__ stats, double count, float* __restrict__ running_mean,
                                    float* __restrict__ running_std, long* __restrict__ nbt, const float* __restrict__ weight,
                                    const float* __restrict__ bias, float* __restrict__ coef, int d, int training,
                                    float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const float rm = running_mean[c], rs = running_std[c], w = weight[c], bb = bias[c];
    if (training) {
        const double nb = (double)(*nbt);
        const float rmax = fminf(fmaxf((float)(2.0 / 35000.0 * nb + 25.0 / 35.0), 1.f), 3.f);   // batchrenorm.py:40-44
        const float dmax = fminf(fmaxf((float)(5.0 / 20000.0 * nb - 25.0 / 20.0), 0.f), 5.f);   // batchrenorm.py:46-50
        const double mean = stats[c] / count;
        double var = stats[d + c] / count - mean * mean;
        if (var < 0) var = 0;
        const float sigma = (float)sqrt(var), s = sigma + eps;
        const float r = fminf(fmaxf(s / rs, 1.f / rmax), rmax);
        const float dd = fminf(fmaxf(((float)mean - rm) / rs, -dmax), dmax);
        coef[c] = (float)mean; coef[d + c] = s; coef[2 * d + c] = r; coef[3 * d + c] = dd;
        coef[4 * d + c] = w * r / s;
        coef[5 * d + c] = w * (dd - (float)mean * r / s) + bb;
        running_mean[c] = rm + momentum * ((float)mean - rm);
        running_std[c] = rs + momentum * (s - rs);
    } else {
        coef[c] = rm; coef[d + c] = rs; coef[2 * d + c] = 1.f; coef[3 * d + c] = 0.f;
        coef[4 * d + c] = w / rs;
        coef[5 * d + c] = bb - w * rm / rs;
    }
}
__global__ void brn_bump_kernel(long* nbt) { if (threadIdx.x == 0 && blockIdx.x == 0) *nbt += 1; }

__global__ void affine_silu_fwd_kernel(const bf16* __restrict__ h, const float* __restrict__ coef, bf16* __restrict__ y,
                                       long M, int d) {
    const long total = M * (d / 8);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (d / 8)) * 8;
        float v[8], a[8], b[8];
        load8(h + idx * 8, v); load8(coef + 4 * d + c, a); load8(coef + 5 * d + c, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = siluf_(v[e] * a[e] + b[e]);
        store8(y + idx * 8, v);
    }
}

// pass 1 of the backward: per-channel sum(dz), sum(dz * xhat0), dz = dy * silu'(h*A+Bc), xhat0 = (h-mean)/s
__global__ __launch_bounds__(256) void brn_bwd_reduce_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h,
                                                             const float* __restrict__ coef, double* __restrict__ red,
                                                             long M, int d, int rows_per_thread) {
    __shared__ float redb[1024];
    const int CG = d / CV;
    const SlabGeo geo(CG);
    const long nslab = (M + rows_per_thread - 1) / rows_per_thread;
    const long slab = (long)blockIdx.x * geo.spb + geo.sl;
    const bool live = geo.active && slab < nslab;
    const int c0 = live ? geo.cg * CV : 0;
    const long r0 = live ? slab * rows_per_thread : 0, r1 = live ? min(M, r0 + rows_per_thread) : 0;
    float mean[CV], is[CV], A[CV], Bc[CV];
    load4(coef + c0, mean); load4(coef + d + c0, is); load4(coef + 4 * d + c0, A); load4(coef + 5 * d + c0, Bc);
#pragma unroll
    for (int e = 0; e < CV; ++e) is[e] = 1.f / is[e];
    float s1[CV] = {0.f, 0.f, 0.f, 0.f}, s2[CV] = {0.f, 0.f, 0.f, 0.f};
    for (long r = r0; r < r1; r += 4) {                      // 4 rows of loads in flight per thread
        float g[4][CV], x[4][CV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {            // unconditional loads (row clamped, gradient zeroed past the end): a load under a
            const long rr = min(r + u, r1 - 1);  // run-time test is branched around and waited for on its own (vmcnt(0) per row)
            load4(dy + rr * d + c0, g[u]); load4(h + rr * d + c0, x[u]);
            if (r + u >= r1) {
#pragma unroll
                for (int e = 0; e < CV; ++e) g[u][e] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < CV; ++e) {
                const float dz = g[u][e] * dsiluf_(x[u][e] * A[e] + Bc[e]);
                s1[e] += dz; s2[e] += dz * (x[u][e] - mean[e]) * is[e];
            }
    }
    slab_reduce_add<double>(s1, geo, redb, red, d);
    slab_reduce_add<double>(s2, geo, redb, red + d, d);
}

// bcoef rows: 0 k0, 1 k1, 2 k2  with  dh = k0*dz - k1 - xhat0*k2 ; accumulates d(weight), d(bias) of BatchRenorm
__global__ void brn_bwd_finalize_kernel(const double* __restrict__ red, double count, const float* __restrict__ coef,
                                        const float* __restrict__ weight, float* __restrict__ bcoef,
                                        float* __restrict__ dweight, float* __restrict__ dbias, int d, int training, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const float s = coef[d + c], r = coef[2 * d + c], dd = coef[3 * d + c], w = weight[c];
    const float S1 = (float)red[c], S2 = (float)red[d + c];
    const float k0 = coef[4 * d + c];                      // w*r/s (training) or w/running_std (eval)
    if (training) {
        const float sigma = s - eps;
        bcoef[c] = k0;
        bcoef[d + c] = k0 * (float)(red[c] / count);
        bcoef[2 * d + c] = sigma > 0.f ? k0 * (s / sigma) * (float)(red[d + c] / count) : 0.f;
    } else { bcoef[c] = k0; bcoef[d + c] = 0.f; bcoef[2 * d + c] = 0.f; }
    atomicAdd(dweight + c, r * S2 + dd * S1);              // sum dz * xhat,  xhat = xhat0*r + d
    atomicAdd(dbias + c, S1);
}

// pass 2: dh on the fly -> depthwise-conv input gradient -> GLU backward; dw-conv weight/bias gradients.
template <int K>
__global__ __launch_bounds__(256) void dwconv_glu_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h,
                                                             const bf16* __restrict__ g, const int* __restrict__ len,
                                                             const float* __restrict__ w, const float* __restrict__ coef,
                                                             const float* __restrict__ bcoef, bf16* __restrict__ dg,
                                                             float* __restrict__ dwt, float* __restrict__ dbias,
                                                             float* __restrict__ dgcs, int B, int N, int d, int TN) {
    constexpr int P = (K - 1) / 2;
    __shared__ float redw[1024];
    const int CG = d / CV, spb = (N + TN - 1) / TN;
    const SlabGeo geo(CG);
    const long slab = (long)blockIdx.x * geo.spb + geo.sl;
    const bool live = geo.active && slab < (long)B * spb;
    const int b = live ? (int)(slab / spb) : 0, n0 = live ? (int)(slab % spb) * TN : 0;
    const int c0 = live ? geo.cg * CV : 0;
    const int L = len ? len[b] : N;
    float wk[K][CV], dhw[K][CV], aw[K][CV], gw[K][CV], gb[CV];
    float sv[CV] = {0.f, 0.f, 0.f, 0.f}, sg2[CV] = {0.f, 0.f, 0.f, 0.f};   // column sums of dg (the pointwise_conv1 bias gradient)
    float mean[CV], is[CV], A[CV], Bc[CV], k0[CV], k1[CV], k2[CV];
    load4(coef + c0, mean); load4(coef + d + c0, is); load4(coef + 4 * d + c0, A); load4(coef + 5 * d + c0, Bc);
    load4(bcoef + c0, k0); load4(bcoef + d + c0, k1); load4(bcoef + 2 * d + c0, k2);
#pragma unroll
    for (int e = 0; e < CV; ++e) {
        is[e] = 1.f / is[e]; gb[e] = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) { wk[j][e] = w[(c0 + e) * K + j]; dhw[j][e] = 0.f; aw[j][e] = 0.f; gw[j][e] = 0.f; }
    }
    const bf16* gbp = g + (long)b * N * 2 * d;
    const int nend = live ? min(N, n0 + TN) : n0 - 2 * P;        // dead threads: empty loop, then join the reductions
    // The dw-conv weight gradient  dW[j] = sum_n dh[n] * a[n + j - P]  is accumulated for window CENTRES in
    // [n0, nend) only, so slabs partition the sum exactly.
    // Software pipeline: the raw loads of step t+1 (leading edge dy/h/g and the centre's g) are issued before step t is
    // processed — the loop is a serial chain per thread and was latency-bound, not bandwidth-bound.
    struct Raw { bf16x4 gy, hx, va, ga, cva, cga; };
    const bf16x4 z4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    auto fetch = [&](int t) {
        Raw r; r.gy = z4; r.hx = z4; r.va = z4; r.ga = z4; r.cva = z4; r.cga = z4;
        if (t >= 0 && t < N) {
            const long row = (long)b * N + t;
            r.gy = *reinterpret_cast<const bf16x4*>(dy + row * d + c0);
            r.hx = *reinterpret_cast<const bf16x4*>(h + row * d + c0);
            if (t < L) {
                r.va = *reinterpret_cast<const bf16x4*>(gbp + (long)t * 2 * d + c0);
                r.ga = *reinterpret_cast<const bf16x4*>(gbp + (long)t * 2 * d + d + c0);
            }
        }
        const int n = t - P;
        if (n >= n0 && n < nend && n < L) {
            r.cva = *reinterpret_cast<const bf16x4*>(gbp + (long)n * 2 * d + c0);
            r.cga = *reinterpret_cast<const bf16x4*>(gbp + (long)n * 2 * d + d + c0);
        }
        return r;
    };
    Raw cur = fetch(n0 - P);
    for (int t = n0 - P; t < nend + P; ++t) {
        const Raw nxt = fetch(t + 1 < nend + P ? t + 1 : -1000000);
#pragma unroll
        for (int j = 0; j < K - 1; ++j)
#pragma unroll
            for (int e = 0; e < CV; ++e) { dhw[j][e] = dhw[j + 1][e]; aw[j][e] = aw[j + 1][e]; }
        if (t >= 0 && t < N) {
#pragma unroll
            for (int e = 0; e < CV; ++e) {
                const float hx = (float)cur.hx[e];
                const float dz = (float)cur.gy[e] * dsiluf_(hx * A[e] + Bc[e]);
                dhw[K - 1][e] = k0[e] * dz - k1[e] - (hx - mean[e]) * is[e] * k2[e];
                aw[K - 1][e] = (t < L) ? (float)cur.va[e] * sigmoidf_((float)cur.ga[e]) : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < CV; ++e) { dhw[K - 1][e] = 0.f; aw[K - 1][e] = 0.f; }
        }
        const int n = t - P;                         // windows hold dh[n-P..n+P], a[n-P..n+P]
        if (n >= n0 && n < nend) {
            float dv[CV], dgt[CV];
#pragma unroll
            for (int e = 0; e < CV; ++e) {
                float acc = 0.f;
                const float dhc = dhw[P][e];
#pragma unroll
                for (int j = 0; j < K; ++j) { acc += wk[j][e] * dhw[K - 1 - j][e]; gw[j][e] += dhc * aw[j][e]; }
                gb[e] += dhc;
                if (n < L) {
                    const float sg = sigmoidf_((float)cur.cga[e]);
                    dv[e] = acc * sg; dgt[e] = acc * (float)cur.cva[e] * sg * (1.f - sg);
                } else { dv[e] = 0.f; dgt[e] = 0.f; }
                sv[e] += dv[e]; sg2[e] += dgt[e];
            }
            bf16* o = dg + ((long)b * N + n) * 2 * d;
            store4(o + c0, dv); store4(o + d + c0, dgt);
        }
        cur = nxt;
    }
    // gradients of the K taps are accumulated TRANSPOSED (dwt[j][c]) so a wave's atomics hit consecutive floats
    slab_reduce_add<float>(gb, geo, redw, dbias, d);
#pragma unroll
    for (int j = 0; j < K; ++j) slab_reduce_add<float>(gw[j], geo, redw, dwt + (long)j * d, d);
    if (dgcs) { slab_reduce_add<float>(sv, geo, redw, dgcs, d); slab_reduce_add<float>(sg2, geo, redw, dgcs + d, d); }
}

int pick_tn(long B, long N, long CG) {
    int TN = 64;
    while (TN > 8 && B * cdiv(N, TN) * CG < 131072) TN >>= 1;
    return TN;
}

}  // namespace

// GLU(dim=channels) -> zero padded frames -> depthwise Conv1d(k, padding=(k-1)/2, groups=d, bias) and the batch
// statistics BatchRenorm needs.  g: (B,N,2d) bf16 = pw1 output; h: (B,N,d) bf16; stats: f64 [2][d], PRE-ZEROED.
// Replaces convolution.py:106-113 (+ flashfftconv conv1dFunc seam, convolution.py:6-22).
SCONF_API int sconf_glu_dwconv_fwd(const void* g, const int32_t* lengths, const float* w, const float* bias, void* h,
                                   double* stats, int64_t B, int64_t N, int64_t d, int64_t ksize, hipStream_t stream) {
    SCONF_REQUIRE(d % CV == 0, "sconf_glu_dwconv_fwd: d must be a multiple of 4");
    if (B * N == 0) return 0;
    const int TN = pick_tn(B, N, d / CV);
    const int cpb = (int)std::min<long>(64, d / CV), spbk = 256 / cpb;
    dim3 grid(cdiv(B * cdiv(N, TN), spbk), cdiv(d / CV, cpb)), block(256);
#define L(KK) hipLaunchKernelGGL((glu_dwconv_fwd_kernel<KK>), grid, block, 0, stream, (const bf16*)g, lengths, w, bias, (bf16*)h, stats, (int)B, (int)N, (int)d, TN)
    switch (ksize) { case 3: L(3); break; case 5: L(5); break; case 7: L(7); break; case 9: L(9); break;
        default: return sconf_set_error("sconf_glu_dwconv_fwd: unsupported kernel size %ld (3,5,7,9)", (long)ksize); }
#undef L
    SCONF_LAUNCH_OK("sconf_glu_dwconv_fwd");
    return 0;
}

// BatchRenorm1d statistics -> per-channel coefficients (coef: f32 [6][d]); training: EMA update of the running
// buffers and num_batches_tracked += 1, in place (batchrenorm.py:71-85).
SCONF_API int sconf_brn_finalize(const double* stats, int64_t count, float* running_mean, float* running_std,
                                 int64_t* num_batches_tracked, const float* weight, const float* bias, float* coef,
                                 int64_t d, int training, float eps, float momentum, hipStream_t stream) {
    hipLaunchKernelGGL(brn_finalize_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, stats, (double)count, running_mean,
                       running_std, (long*)num_batches_tracked, weight, bias, coef, (int)d, training, eps, momentum);
    if (training) hipLaunchKernelGGL(brn_bump_kernel, dim3(1), dim3(64), 0, stream, (long*)num_batches_tracked);
    SCONF_LAUNCH_OK("sconf_brn_finalize");
    return 0;
}

// y = SiLU(h * A[c] + Bc[c])  (BatchRenorm normalise + affine + activation, convolution.py:119-121)
SCONF_API int sconf_affine_silu_fwd(const void* h, const float* coef, void* y, int64_t M, int64_t d, hipStream_t stream) {
    SCONF_REQUIRE(d % 8 == 0, "sconf_affine_silu_fwd: d must be a multiple of 8");
    if (M == 0) return 0;
    const int blocks = (int)std::min<long>(cdiv(M * (d / 8), 256), 8192);
    hipLaunchKernelGGL(affine_silu_fwd_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16*)h, coef, (bf16*)y, (long)M, (int)d);
    SCONF_LAUNCH_OK("sconf_affine_silu_fwd");
    return 0;
}

// Backward of [GLU -> mask -> dwconv -> BatchRenorm -> SiLU].  dy: grad wrt the SiLU output (B,N,d) bf16.
// red: f64 [2][d] scratch PRE-ZEROED; bcoef: f32 [3][d] scratch.  Writes dg (B,N,2d) bf16; ACCUMULATES (+=)
// d(dw weight) TRANSPOSED as [k][d] (the caller transposes the tiny tensor back), d(dw bias) [d], d(brn weight) [d],
// d(brn bias) [d]; dg_colsum (optional, [2d]) += column sums of dg = the pointwise_conv1 bias gradient (convolution.py:100).
SCONF_API int sconf_convmod_bwd(const void* dy, const void* h, const void* g, const int32_t* lengths, const float* w,
                                const float* brn_weight, const float* coef, double* red, float* bcoef, void* dg,
                                float* dw, float* dbias, float* dbrn_weight, float* dbrn_bias, float* dg_colsum,
                                int64_t B, int64_t N, int64_t d, int64_t ksize, int training, float eps, hipStream_t stream) {
    SCONF_REQUIRE(d % CV == 0, "sconf_convmod_bwd: d must be a multiple of 4");
    const long M = B * N;
    if (M == 0) return 0;
    const int CG = (int)(d / CV);
    int rpt = 64;
    while (rpt > 8 && cdiv(M, rpt) * CG < 131072) rpt >>= 1;
    const int cpb = std::min(64, CG), spbk = 256 / cpb;
    hipLaunchKernelGGL(brn_bwd_reduce_kernel, dim3(cdiv(cdiv(M, rpt), spbk), cdiv(CG, cpb)), dim3(256), 0, stream,
                       (const bf16*)dy, (const bf16*)h, coef, red, M, (int)d, rpt);
    hipLaunchKernelGGL(brn_bwd_finalize_kernel, dim3(cdiv(d, 256)), dim3(256), 0, stream, red, (double)M, coef, brn_weight,
                       bcoef, dbrn_weight, dbrn_bias, (int)d, training, eps);
    const int TN = pick_tn(B, N, CG);
    dim3 grid(cdiv(B * cdiv(N, TN), spbk), cdiv(CG, cpb)), block(256);
#define L(KK) hipLaunchKernelGGL((dwconv_glu_bwd_kernel<KK>), grid, block, 0, stream, (const bf16*)dy, (const bf16*)h, (const bf16*)g, lengths, w, coef, bcoef, (bf16*)dg, dw, dbias, dg_colsum, (int)B, (int)N, (int)d, TN)
    switch (ksize) { case 3: L(3); break; case 5: L(5); break; case 7: L(7); break; case 9: L(9); break;
        default: return sconf_set_error("sconf_convmod_bwd: unsupported kernel size %ld (3,5,7,9)", (long)ksize); }
#undef L
    SCONF_LAUNCH_OK("sconf_convmod_bwd");
    return 0;
}
