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

namespace {

constexpr int CG8 = 8;                               // channels per thread (16-B bf16 accesses)

// ---- conv0: Conv2d(1 -> C, 3x3, stride 2, pad 1) + bias, pre-activation out -----------------
template <typename TX>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, bf16* __restrict__ y,
                                                        int B, int F, int T, int C, int T2, int F2) {
    const int cgs = C / CG8;
    const long total = (long)B * T2 * F2 * cgs;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c0 = (int)(idx % cgs) * CG8;
        const long pos = idx / cgs;
        const int f2 = (int)(pos % F2), t2 = (int)((pos / F2) % T2), b = (int)(pos / ((long)F2 * T2));
        float in[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = 2 * t2 + i - 1, f = 2 * f2 + j - 1;
                in[i * 3 + j] = (t >= 0 && t < T && f >= 0 && f < F) ? ld_f(x + ((long)b * F + f) * T + t) : 0.f;
            }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = bias[c0 + e];
#pragma unroll
            for (int k = 0; k < 9; ++k) acc += w[(c0 + e) * 9 + k] * in[k];
            o[e] = acc;
        }
        store8(y + pos * C + c0, o);
    }
}

// ---- depthwise Conv2d(C, 3x3, stride 2, pad 1, groups=C) + bias on SiLU(in) ------------------
__global__ __launch_bounds__(256) void dwconv2d_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, bf16* __restrict__ y,
                                                           int B, int Ti, int Fi, int C, int To, int Fo) {
    const int cgs = C / CG8;
    const long total = (long)B * To * Fo * cgs;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c0 = (int)(idx % cgs) * CG8;
        const long pos = idx / cgs;
        const int fo = (int)(pos % Fo), to = (int)((pos / Fo) % To), b = (int)(pos / ((long)Fo * To));
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias[c0 + e];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = 2 * to + i - 1, f = 2 * fo + j - 1;
                if (t >= 0 && t < Ti && f >= 0 && f < Fi) {
                    float v[8]; load8(x + (((long)b * Ti + t) * Fi + f) * C + c0, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += w[(c0 + e) * 9 + i * 3 + j] * siluf_(v[e]);
                }
            }
        store8(y + pos * C + c0, acc);
    }
}

// ---- input gradient of the depthwise conv, times SiLU'(pre_in) ------------------------------
__global__ __launch_bounds__(256) void dwconv2d_bwd_input_kernel(const bf16* __restrict__ dout, const float* __restrict__ w,
                                                                 const bf16* __restrict__ pre_in, bf16* __restrict__ dpre_in,
                                                                 int B, int Ti, int Fi, int C, int To, int Fo) {
    const int cgs = C / CG8;
    const long total = (long)B * Ti * Fi * cgs;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c0 = (int)(idx % cgs) * CG8;
        const long pos = idx / cgs;
        const int fi = (int)(pos % Fi), ti = (int)((pos / Fi) % Ti), b = (int)(pos / ((long)Fi * Ti));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
                float g[8]; load8(dout + (((long)b * To + to) * Fo + fo) * C + c0, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += w[(c0 + e) * 9 + i * 3 + j] * g[e];
            }
        }
        float p[8]; load8(pre_in + pos * C + c0, p);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= dsiluf_(p[e]);
        store8(dpre_in + pos * C + c0, acc);
    }
}

// ---- weight / bias gradients of a 3x3 stride-2 conv whose output gradient is channels-last -----
// DEPTHWISE: input is SiLU(pre_in[.., c]); otherwise (conv0) the single-channel mel input (B,F,T).
template <bool DEPTHWISE, typename TX>
__global__ __launch_bounds__(256) void conv3x3s2_bwd_weight_kernel(const bf16* __restrict__ dout, const void* __restrict__ in_,
                                                                   float* __restrict__ dw, float* __restrict__ dbias,
                                                                   int B, int Ti, int Fi, int C, int To, int Fo, int strip) {
    const int cgs = C / CG8;
    const long npos = (long)B * To * Fo;
    const long nstrips = (npos + strip - 1) / strip;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nstrips * cgs) return;
    const int c0 = (int)(idx % cgs) * CG8;
    const long p0 = (idx / cgs) * strip, p1 = min(npos, p0 + strip);
    float gw[9][8], gb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gb[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) gw[k][e] = 0.f; }
    for (long pos = p0; pos < p1; ++pos) {
        const int fo = (int)(pos % Fo), to = (int)((pos / Fo) % To), b = (int)(pos / ((long)Fo * To));
        float g[8]; load8(dout + pos * C + c0, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) gb[e] += g[e];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = 2 * to + i - 1, f = 2 * fo + j - 1;
                if (t < 0 || t >= Ti || f < 0 || f >= Fi) continue;
                if (DEPTHWISE) {
                    float v[8]; load8((const bf16*)in_ + (((long)b * Ti + t) * Fi + f) * C + c0, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gw[i * 3 + j][e] += g[e] * siluf_(v[e]);
                } else {
                    const float v = ld_f((const TX*)in_ + ((long)b * Fi + f) * Ti + t);   // mel is (B,F,T)
#pragma unroll
                    for (int e = 0; e < 8; ++e) gw[i * 3 + j][e] += g[e] * v;
                }
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(dbias + c0 + e, gb[e]);
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(dw + (c0 + e) * 9 + k, gw[k][e]);
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

inline int grid_for(long total_threads) { return (int)std::min<long>(cdiv(total_threads, 256), 16384); }

}  // namespace

// Conv2d(1->C,3,s2,p1)+bias on the (B,F,T) mel input; pre-activation out (B,T2,F2,C) bf16.
// Replaces subsampling.py:299-306 (self.conv[0]); SiLU (conv[1]) is applied by the consumer.
SCONF_API int sconf_sub_conv0_fwd(const void* x, int x_dtype, const float* w, const float* bias, void* y,
                                  int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 8 == 0, "sconf_sub_conv0_fwd: C must be a multiple of 8");
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1);
    const long total = B * T2 * F2 * (C / 8);
    if (total == 0) return 0;
    if (x_dtype == SCONF_F32) hipLaunchKernelGGL((conv0_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, stream, (const float*)x, w, bias, (bf16*)y, (int)B, (int)F, (int)T, (int)C, T2, F2);
    else hipLaunchKernelGGL((conv0_fwd_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, (int)B, (int)F, (int)T, (int)C, T2, F2);
    SCONF_LAUNCH_OK("sconf_sub_conv0_fwd");
    return 0;
}

// y = dwConv2d(SiLU(x)) + bias, channels-last.  Replaces subsampling.py:309-330 (conv[2], conv[5]) fused with
// the preceding activation (conv[1], conv[4]).
SCONF_API int sconf_sub_dwconv_fwd(const void* x, const float* w, const float* bias, void* y,
                                   int64_t B, int64_t Ti, int64_t Fi, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 8 == 0, "sconf_sub_dwconv_fwd: C must be a multiple of 8");
    const int To = (int)((Ti - 1) / 2 + 1), Fo = (int)((Fi - 1) / 2 + 1);
    const long total = B * To * Fo * (C / 8);
    if (total == 0) return 0;
    hipLaunchKernelGGL(dwconv2d_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, (int)B, (int)Ti, (int)Fi, (int)C, To, Fo);
    SCONF_LAUNCH_OK("sconf_sub_dwconv_fwd");
    return 0;
}

// dpre_in = SiLU'(pre_in) * dwconv^T(dout);  dw/dbias ACCUMULATED (+=).
SCONF_API int sconf_sub_dwconv_bwd(const void* dout, const float* w, const void* pre_in, void* dpre_in, float* dw, float* dbias,
                                   int64_t B, int64_t Ti, int64_t Fi, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 8 == 0, "sconf_sub_dwconv_bwd: C must be a multiple of 8");
    const int To = (int)((Ti - 1) / 2 + 1), Fo = (int)((Fi - 1) / 2 + 1);
    const long total = B * Ti * Fi * (C / 8);
    if (total == 0) return 0;
    hipLaunchKernelGGL(dwconv2d_bwd_input_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)dout, w, (const bf16*)pre_in, (bf16*)dpre_in, (int)B, (int)Ti, (int)Fi, (int)C, To, Fo);
    const long npos = B * To * Fo;
    int strip = 64;
    while (strip > 4 && cdiv(npos, strip) * (C / 8) < 65536) strip >>= 1;
    const long threads = cdiv(npos, strip) * (C / 8);
    hipLaunchKernelGGL((conv3x3s2_bwd_weight_kernel<true, float>), dim3(cdiv(threads, 256)), dim3(256), 0, stream, (const bf16*)dout, pre_in, dw, dbias, (int)B, (int)Ti, (int)Fi, (int)C, To, Fo, strip);
    SCONF_LAUNCH_OK("sconf_sub_dwconv_bwd");
    return 0;
}

// conv0 parameter gradients (the mel input needs no gradient).  dw [C][9], dbias [C] ACCUMULATED (+=).
SCONF_API int sconf_sub_conv0_bwd(const void* dpre0, const void* x, int x_dtype, float* dw, float* dbias,
                                  int64_t B, int64_t F, int64_t T, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 8 == 0, "sconf_sub_conv0_bwd: C must be a multiple of 8");
    const int T2 = (int)((T - 1) / 2 + 1), F2 = (int)((F - 1) / 2 + 1);
    const long npos = B * T2 * F2;
    if (npos == 0) return 0;
    int strip = 64;
    while (strip > 4 && cdiv(npos, strip) * (C / 8) < 65536) strip >>= 1;
    const long threads = cdiv(npos, strip) * (C / 8);
    if (x_dtype == SCONF_F32) hipLaunchKernelGGL((conv3x3s2_bwd_weight_kernel<false, float>), dim3(cdiv(threads, 256)), dim3(256), 0, stream, (const bf16*)dpre0, x, dw, dbias, (int)B, (int)T, (int)F, (int)C, T2, F2, strip);
    else hipLaunchKernelGGL((conv3x3s2_bwd_weight_kernel<false, bf16>), dim3(cdiv(threads, 256)), dim3(256), 0, stream, (const bf16*)dpre0, x, dw, dbias, (int)B, (int)T, (int)F, (int)C, T2, F2, strip);
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
