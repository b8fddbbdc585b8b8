// Shared device/host helpers for the sconf HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define SCONF_API extern "C" __attribute__((visibility("default")))

// ---- error reporting (C ABI: non-zero return + sconf_last_error()) ---------------------------
int sconf_set_error(const char* fmt, ...);
#define SCONF_REQUIRE(cond, ...) do { if (!(cond)) return sconf_set_error(__VA_ARGS__); } while (0)
#define SCONF_LAUNCH_OK(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) return sconf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)

enum SconfDtype { SCONF_F32 = 0, SCONF_BF16 = 1 };
enum SconfAct { SCONF_ACT_NONE = 0, SCONF_ACT_GELU = 1, SCONF_ACT_SILU = 2, SCONF_ACT_DGELU = 3, SCONF_ACT_DSILU = 4,
                SCONF_ACT_GELU_DSAVE = 5,   // out = gelu(v); `pre` receives gelu'(v) (what the backward multiplies by)
                SCONF_ACT_MULAUX = 6,       // out = v * aux
                SCONF_ACT_SMAXBWD = 7 };    // out = (v - rowv[m]) * aux (softmax backward; sconf_gemm_softmax_bwd only)


// ---- device helpers --------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }   // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

template <typename T> __device__ __forceinline__ float ld_f(const T* p);
template <> __device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f<bf16>(const bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st_f(T* p, float v);
template <> __device__ __forceinline__ void st_f<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f<bf16>(bf16* p, float v) { *p = (bf16)v; }

// 8-element vector load/store as floats (16 B for bf16, 2x16 B for f32)
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load4(const bf16* p, float (&v)[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void store4(bf16* p, const float (&v)[4]) {
    bf16x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x4*>(p) = t;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block reductions over blockDim.x (multiple of 64, <= 1024); `sh` needs 16 floats; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// Activations on the fast hardware ops (v_exp_f32 = 2^x, v_rcp_f32).  exp2 of a large positive argument gives +inf and
// rcp(inf) = 0, so the saturated tails are exact without branches.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float dsiluf_(float x) { float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f * sigmoidf_(2.f * x) - 1.f; }
// GELU, tanh approximation (reference: F.gelu(approximate='tanh'), fused_dense.py:466).  0.5(1 + tanh(y)) == sigmoid(2y),
// so  gelu(x) = x * sigmoid(2c(x + 0.044715 x^3)),  c = sqrt(2/pi):  one exp2 + one rcp per element.
__device__ __forceinline__ float geluf_(float x) {
    const float k = -2.f * 0.7978845608028654f * 1.4426950408889634f;         // -2c * log2(e)
    const float z = x * (1.f + 0.044715f * x * x);
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(k * z));
}
// gelu(x) and gelu'(x) sharing the one sigmoid (forward epilogue that saves the derivative for the backward)
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
    const float c2 = 2.f * 0.7978845608028654f;
    const float x2 = x * x;
    const float z = x * (1.f + 0.044715f * x2);
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-c2 * 1.4426950408889634f * z));
    g = x * s;
    dg = s + g * (1.f - s) * c2 * (1.f + 3.f * 0.044715f * x2);
}
__device__ __forceinline__ float dgeluf_(float x) {
    const float c2 = 2.f * 0.7978845608028654f;
    const float x2 = x * x;
    const float z = x * (1.f + 0.044715f * x2);
    const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-c2 * 1.4426950408889634f * z));
    return s + x * s * (1.f - s) * c2 * (1.f + 3.f * 0.044715f * x2);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
