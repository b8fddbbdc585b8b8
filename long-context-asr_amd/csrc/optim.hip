// Fused MADGRAD step + global-norm gradient clip over FLAT f32 buffers (one launch for the whole model).
// Replaces the per-tensor Python loop of lcasr/optim/madgrad.py:81-212 (dense, momentum != 0 branch) and
// torch.nn.utils.clip_grad_norm_ + GradScaler's inf check in exp/train.py:46-61.  Optionally refreshes a
// bf16 shadow copy of the parameters for the next forward's GEMMs in the same pass.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
    __shared__ float sh[16];
    float acc = 0.f;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i + 4 <= n; i += stride) { float v[4]; load4(g + i, v); acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]; }
    if (i < n) for (long j = i; j < n && j < i + 4; ++j) acc += g[j] * g[j];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(out, (double)acc);
}

// p_{k+1}: gss += lamb g^2 ; rms = cbrt(gss)+eps ; s += lamb g ; z = x0 - s/rms ; p = (1-ck) p + ck z,  lamb = lr sqrt(k+1)
// The step counter k comes from the host (k_host) or, when k_dev is given, from device memory: it only advances on APPLIED
// steps (sconf_madgrad_advance), like the reference where GradScaler skips optimizer.step() on inf/nan (exp/train.py:54-57).
// x0 is created at the first applied step (k == 0: x0 := p, madgrad.py:121-125), so weights loaded after the optimiser was
// constructed are the anchor, not the values the parameters had at construction.
__global__ __launch_bounds__(256) void madgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ gss,
                                                      float* __restrict__ s, float* __restrict__ x0, bf16* __restrict__ shadow,
                                                      long n, const double* __restrict__ sumsq, float max_norm, float grad_scale,
                                                      float lr, float ck, float eps, float weight_decay, long k_host,
                                                      const long* __restrict__ k_dev) {
    const long k = k_dev ? *k_dev : k_host;
    const float lamb = lr * sqrtf((float)(k + 1));
    const bool first = k == 0;
    float coef = grad_scale;
    if (sumsq) {
        const double tot = sqrt(*sumsq) * (double)fabsf(grad_scale);
        if (!(tot < INFINITY)) return;                         // inf/nan gradients: skip the step (GradScaler semantics)
        if (max_norm > 0.f) coef *= fminf(1.f, max_norm / ((float)tot + 1e-6f));
    }
    // 4 elements per thread and instruction (16-B loads / stores); the flat buffers are 16-B aligned and padded to 4
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long stride = (long)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        float pv[4], gv[4], qv[4], sv[4], xv[4];
        load4(p + i, pv); load4(g + i, gv); load4(gss + i, qv); load4(s + i, sv);
        if (first) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = pv[e];
            store4(x0 + i, xv);
        } else load4(x0 + i, xv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gg = gv[e] * coef;
            if (weight_decay != 0.f) gg += weight_decay * pv[e];
            qv[e] += lamb * gg * gg;
            sv[e] += lamb * gg;
            const float rms = cbrtf(qv[e]) + eps;
            const float z = xv[e] - sv[e] / rms;
            pv[e] = pv[e] * (1.f - ck) + ck * z;
        }
        store4(gss + i, qv); store4(s + i, sv); store4(p + i, pv);
        if (shadow) store4(shadow + i, pv);
    }
    if (i < n) {                                               // tail (n % 4 elements), first thread past the vector part only
        for (long j = i; j < n; ++j) {
            float pv = p[j];
            float gv = g[j] * coef;
            if (weight_decay != 0.f) gv += weight_decay * pv;
            const float q = gss[j] + lamb * gv * gv;
            const float sv = s[j] + lamb * gv;
            const float rms = cbrtf(q) + eps;
            if (first) x0[j] = pv;
            const float z = x0[j] - sv / rms;
            pv = pv * (1.f - ck) + ck * z;
            gss[j] = q; s[j] = sv; p[j] = pv;
            if (shadow) shadow[j] = (bf16)pv;
        }
    }
}

__global__ void madgrad_advance_kernel(long* __restrict__ k_dev, const double* __restrict__ sumsq, float grad_scale) {
    if (sumsq && !(sqrt(*sumsq) * (double)fabsf(grad_scale) < INFINITY)) return;    // the step was skipped: k does not move
    *k_dev += 1;
}

}  // namespace

// out (f64 scalar, PRE-ZEROED) += sum(g^2)
SCONF_API int sconf_sumsq(const float* g, int64_t n, double* out, hipStream_t stream) {
    if (n == 0) return 0;
    const int blocks = (int)std::min<long>(cdiv(n, 1024), 2048);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, stream, g, (long)n, out);
    SCONF_LAUNCH_OK("sconf_sumsq");
    return 0;
}

// One MADGRAD step over flat buffers.  sumsq (nullable): global sum of squared (unscaled-by-grad_scale) gradients;
// when given, gradients are clipped to max_norm (if > 0) and the step is skipped when the norm is not finite.
// grad_scale multiplies every gradient first (1/world_size for averaged DDP, 1/loss_scale for a GradScaler).
// k: the number of steps applied so far, from the host, or read from k_dev (device int64, nullable) when that is given.
SCONF_API int sconf_madgrad_step(float* p, const float* g, float* grad_sum_sq, float* s, float* x0, void* bf16_shadow,
                                 int64_t n, const double* sumsq, float max_norm, float grad_scale, float lr, float momentum,
                                 float eps, float weight_decay, int64_t k, const int64_t* k_dev, hipStream_t stream) {
    if (n == 0) return 0;
    SCONF_REQUIRE(momentum > 0.f && momentum < 1.f, "sconf_madgrad_step: momentum must be in (0,1)");
    if (lr != 0.f) lr = lr + eps;                              // madgrad.py:100-101
    const float ck = 1.f - momentum;
    SCONF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)grad_sum_sq | (uintptr_t)s | (uintptr_t)x0) & 15) == 0 && ((uintptr_t)bf16_shadow & 7) == 0,
                  "sconf_madgrad_step: buffers must be 16-byte aligned");
    const int blocks = (int)std::min<long>(cdiv(n, 1024), 4096);
    hipLaunchKernelGGL(madgrad_kernel, dim3(blocks), dim3(256), 0, stream, p, g, grad_sum_sq, s, x0, (bf16*)bf16_shadow, (long)n,
                       sumsq, max_norm, grad_scale, lr, ck, eps, weight_decay, (long)k, (const long*)k_dev);
    SCONF_LAUNCH_OK("sconf_madgrad_step");
    return 0;
}

// After the sconf_madgrad_step launches of one optimiser step (one per parameter group): *k_dev += 1 unless the step was
// skipped (non-finite gradient norm).  The reference keeps one global counter state['k'] (madgrad.py:95-98).
SCONF_API int sconf_madgrad_advance(int64_t* k_dev, const double* sumsq, float grad_scale, hipStream_t stream) {
    SCONF_REQUIRE(k_dev != nullptr, "sconf_madgrad_advance: k_dev is null");
    hipLaunchKernelGGL(madgrad_advance_kernel, dim3(1), dim3(1), 0, stream, (long*)k_dev, sumsq, grad_scale);
    SCONF_LAUNCH_OK("sconf_madgrad_advance");
    return 0;
}
