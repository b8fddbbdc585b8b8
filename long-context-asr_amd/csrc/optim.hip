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

// p_{k+1}: gss += lamb g^2 ; rms = cbrt(gss)+eps ; s += lamb g ; z = x0 - s/rms ; p = (1-ck) p + ck z
__global__ __launch_bounds__(256) void madgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ gss,
                                                      float* __restrict__ s, const float* __restrict__ x0, bf16* __restrict__ shadow,
                                                      long n, const double* __restrict__ sumsq, float max_norm, float grad_scale,
                                                      float lamb, float ck, float eps, float weight_decay) {
    float coef = grad_scale;
    if (sumsq) {
        const double tot = sqrt(*sumsq) * (double)fabsf(grad_scale);
        if (!(tot < INFINITY)) return;                         // inf/nan gradients: skip the step (GradScaler semantics)
        if (max_norm > 0.f) coef *= fminf(1.f, max_norm / ((float)tot + 1e-6f));
    }
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) {
        float pv = p[i];
        float gv = g[i] * coef;
        if (weight_decay != 0.f) gv += weight_decay * pv;
        const float q = gss[i] + lamb * gv * gv;
        const float sv = s[i] + lamb * gv;
        const float rms = cbrtf(q) + eps;
        const float z = x0[i] - sv / rms;
        pv = pv * (1.f - ck) + ck * z;
        gss[i] = q; s[i] = sv; p[i] = pv;
        if (shadow) shadow[i] = (bf16)pv;
    }
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
SCONF_API int sconf_madgrad_step(float* p, const float* g, float* grad_sum_sq, float* s, const float* x0, void* bf16_shadow,
                                 int64_t n, const double* sumsq, float max_norm, float grad_scale, float lr, float momentum,
                                 float eps, float weight_decay, int64_t k, hipStream_t stream) {
    if (n == 0) return 0;
    SCONF_REQUIRE(momentum > 0.f && momentum < 1.f, "sconf_madgrad_step: momentum must be in (0,1)");
    if (lr != 0.f) lr = lr + eps;                              // madgrad.py:100-101
    const float ck = 1.f - momentum;
    const float lamb = lr * sqrtf((float)(k + 1));
    const int blocks = (int)std::min<long>(cdiv(n, 256), 4096);
    hipLaunchKernelGGL(madgrad_kernel, dim3(blocks), dim3(256), 0, stream, p, g, grad_sum_sq, s, x0, (bf16*)bf16_shadow, (long)n,
                       sumsq, max_norm, grad_scale, lamb, ck, eps, weight_decay);
    SCONF_LAUNCH_OK("sconf_madgrad_step");
    return 0;
}
