// HBM-bound helpers of the SConformerXL path: casts, qkv de-interleave + rotary, row softmax /
// log-softmax (forward + backward), bias-gradient column sums, padded-row masking.
#include "common.h"

namespace {
// raw 4-element vectors: loads issued early, converted when consumed
template <typename T> struct Raw4;
template <> struct Raw4<float> { float4 v; __device__ __forceinline__ void get(float (&o)[4]) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; } };
template <> struct Raw4<bf16>  { bf16x4 v; __device__ __forceinline__ void get(float (&o)[4]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (float)v[e]; } };

// ---------------------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= n; i += stride) { float v[8]; load8(s + i, v); store8(d + i, v); }
    if (i < n) for (long j = i; j < n && j < i + 8; ++j) d[j] = (bf16)s[j];
}

__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ s, float* __restrict__ d, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= n; i += stride) { float v[8]; load8(s + i, v); store8(d + i, v); }
    if (i < n) for (long j = i; j < n && j < i + 8; ++j) d[j] = (float)s[j];
}

// dst[c][r] = bf16(src[r][c]): transposed bf16 shadow of an f32 weight (so every dgrad is an NT GEMM whose B operand
// is K-contiguous); 32x32 tiles through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < R) dst[(long)c * R + r] = (bf16)tile[tx][ty + 8 * k];
    }
}

// All bf16 weight shadows of a model in ONE launch (the per-step refresh after the optimiser has moved the f32 masters).
// table[e] = {src f32 (R,C), dst bf16 (R,C) or 0, dstT bf16 (C,R) or 0, R (negative: regroup, below), C, first 32x32-tile index}; entry n is a
// sentinel holding the total tile count.  Block b finds its entry by binary search over the tile prefix.
struct ShadowEntry { const float* src; bf16* dst; bf16* dstT; long R, C, tile0; };
__global__ __launch_bounds__(256) void cast_shadows_kernel(const ShadowEntry* __restrict__ table, int n) {
    __shared__ float tile[32][33];
    const long b = blockIdx.x;
    int lo = 0, hi = n;                                    // table[lo].tile0 <= b < table[hi].tile0
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (table[mid].tile0 <= b) lo = mid; else hi = mid; }
    const ShadowEntry e = table[lo];
    // R < 0: the source rows are the reference's "(h d qkv)" interleave of the fused qkv projection (attention.py:485); the shadow is
    // written with rows regrouped as [q | k | v] (row 3j + w -> w * R/3 + j), so that the GEMM's output columns are three
    // contiguous (h, d) blocks and no de-interleave pass over the activations is needed.
    const bool regroup = e.R < 0;
    const int R = (int)(regroup ? -e.R : e.R), C = (int)e.C, tiles_c = (C + 31) >> 5, R3 = R / 3;
    const int t = (int)(b - e.tile0), r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        const float v = (r < R && c < C) ? e.src[(long)r * C + c] : 0.f;
        tile[ty + 8 * k][tx] = v;
        const int rd = regroup ? (r % 3) * R3 + r / 3 : r;
        if (e.dst && r < R && c < C) e.dst[(long)rd * C + c] = (bf16)v;
    }
    if (!e.dstT) return;                                   // uniform per block
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        const int rd = regroup ? (r % 3) * R3 + r / 3 : r;
        if (c < C && r < R) e.dstT[(long)c * R + rd] = (bf16)tile[tx][ty + 8 * k];
    }
}

// NeoX rotary applied IN PLACE to the q and k blocks of a regrouped qkv activation (M, 3, H, D) bf16 (v untouched): one thread
// rotates 8 pairs (d = i..i+7 and d + D/2) of one (row, q-or-k, head).  cos/sin: (N, D/2) f32 (rotary_emb.py:44-57, 61-73).
__global__ void rotary_inplace_kernel(bf16* __restrict__ qkv, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                      long M, int N, int H, int D) {
    const int half = D / 2, gpr = half / 8;
    const long total = M * 2 * H * gpr;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int gi = (int)(idx % gpr);
        const int h = (int)((idx / gpr) % H);
        const int w = (int)((idx / ((long)gpr * H)) % 2);
        const long m = idx / ((long)gpr * H * 2);
        const int n = (int)(m % N), i0 = gi * 8;
        bf16* p1 = qkv + ((m * 3 + w) * H + h) * D + i0;
        bf16* p2 = p1 + half;
        float c[8], s[8], a[8], b[8], o1[8], o2[8];
        load8(cosT + (long)n * half + i0, c); load8(sinT + (long)n * half + i0, s);
        load8(p1, a); load8(p2, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) { o1[e] = a[e] * c[e] - b[e] * s[e]; o2[e] = b[e] * c[e] + a[e] * s[e]; }   // x*cos + rotate_half(x)*sin
        store8(p1, o1); store8(p2, o2);
    }
}

// ---------------------------------------------------------------------------------------------
// qkv de-interleave + NeoX rotary.  Reference layout of the qkv projection output is
// "b n (h d qkv)" (attention.py:485): column (h*D + d)*3 + {0:q, 1:k, 2:v}.  One thread handles
// 8 rotary pairs (d = i..i+7 and d + D/2) of one (row, head): 2 x 48 contiguous input bytes,
// 16-B output stores.  cos/sin: (N, D/2) f32 tables built exactly like rotary_emb.py:44-57.
template <bool BWD>
__global__ void rotary_qkv_kernel(bf16* __restrict__ qkv, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                  bf16* __restrict__ q, bf16* __restrict__ k, bf16* __restrict__ v,
                                  long M, int N, int H, int D, int use_rot) {
    const int half = D / 2, gpr = half / 8;                     // groups of 8 pairs per (row, head)
    const long total = M * H * gpr;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int gi = (int)(idx % gpr);
        const int h = (int)((idx / gpr) % H);
        const long m = idx / ((long)gpr * H);
        const int n = (int)(m % N);
        const int i0 = gi * 8;
        bf16* src1 = qkv + (m * H * D + (long)h * D + i0) * 3;         // 24 contiguous bf16: (q,k,v) x 8 for d = i0..
        bf16* src2 = src1 + (long)half * 3;                              // same for d + D/2
        const long o1 = (m * H + h) * D + i0, o2 = o1 + half;
        float c[8], s[8];
        if (use_rot) { load8(cosT + (long)n * half + i0, c); load8(sinT + (long)n * half + i0, s); }
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { c[e] = 1.f; s[e] = 0.f; }
        }
        if (!BWD) {
            float a[24], b[24];
            load8(src1, *(float(*)[8])&a[0]); load8(src1 + 8, *(float(*)[8])&a[8]); load8(src1 + 16, *(float(*)[8])&a[16]);
            load8(src2, *(float(*)[8])&b[0]); load8(src2 + 8, *(float(*)[8])&b[8]); load8(src2 + 16, *(float(*)[8])&b[16]);
            float q1[8], q2[8], k1[8], k2[8], v1[8], v2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float qa = a[3 * e], ka = a[3 * e + 1], qb = b[3 * e], kb = b[3 * e + 1];
                q1[e] = qa * c[e] - qb * s[e]; q2[e] = qb * c[e] + qa * s[e];   // x*cos + rotate_half(x)*sin
                k1[e] = ka * c[e] - kb * s[e]; k2[e] = kb * c[e] + ka * s[e];
                v1[e] = a[3 * e + 2]; v2[e] = b[3 * e + 2];
            }
            store8(q + o1, q1); store8(q + o2, q2); store8(k + o1, k1); store8(k + o2, k2);
            store8(v + o1, v1); store8(v + o2, v2);
        } else {
            float q1[8], q2[8], k1[8], k2[8], v1[8], v2[8];
            load8(q + o1, q1); load8(q + o2, q2); load8(k + o1, k1); load8(k + o2, k2); load8(v + o1, v1); load8(v + o2, v2);
            float a[24], b[24];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                                 // transpose of the rotation
                a[3 * e] = q1[e] * c[e] + q2[e] * s[e];     b[3 * e] = q2[e] * c[e] - q1[e] * s[e];
                a[3 * e + 1] = k1[e] * c[e] + k2[e] * s[e]; b[3 * e + 1] = k2[e] * c[e] - k1[e] * s[e];
                a[3 * e + 2] = v1[e];                        b[3 * e + 2] = v2[e];
            }
            store8(src1, *(float(*)[8])&a[0]); store8(src1 + 8, *(float(*)[8])&a[8]); store8(src1 + 16, *(float(*)[8])&a[16]);
            store8(src2, *(float(*)[8])&b[0]); store8(src2 + 8, *(float(*)[8])&b[8]); store8(src2 + 16, *(float(*)[8])&b[16]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row softmax family over C <= 8192 classes, one 256-thread block per row, row kept in registers.
constexpr int SM_IT = 8;

// MODE 0: softmax -> TO;  MODE 1: log_softmax -> TO
template <typename TI, typename TO, int MODE>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C) {
    __shared__ float sh[16];
    const long row = blockIdx.x;
    const TI* xr = x + row * C;
    float v[SM_IT][4];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < SM_IT; ++it) {
        const int c = it * 1024 + threadIdx.x * 4;
        if (c < C) { load4(xr + c, v[it]); mx = fmaxf(mx, fmaxf(fmaxf(v[it][0], v[it][1]), fmaxf(v[it][2], v[it][3]))); }
    }
    mx = block_max(mx, sh);
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < SM_IT; ++it) {
        const int c = it * 1024 + threadIdx.x * 4;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[it][e] -= mx; s += __expf(v[it][e]); }
        }
    }
    s = block_sum(s, sh);
    const float inv = 1.f / s, ls = __logf(s);
    TO* yr = y + row * C;
#pragma unroll
    for (int it = 0; it < SM_IT; ++it) {
        const int c = it * 1024 + threadIdx.x * 4;
        if (c < C) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (MODE == 0) ? __expf(v[it][e]) * inv : v[it][e] - ls;
            store4(yr + c, o);
        }
    }
}

// MODE 0: softmax bwd   dx = y * (dy - sum(dy*y))          (y = probabilities)
// MODE 1: log_softmax bwd dx = dy - exp(y) * sum(dy)         (y = log-probabilities)
// A workgroup walks rows_per_block consecutive rows.  slab (optional, [gridDim.x][C] f32): the workgroup's column sums of the
// dx it wrote (as stored, i.e. after rounding to TO) - the bias gradient of the Linear that produced the logits, without a
// second pass over dx.
template <typename TY, typename TG, typename TO, int MODE>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const TY* __restrict__ y, const TG* __restrict__ dy,
                                                          TO* __restrict__ dx, float* __restrict__ slab, long M, int C, int rows_per_block) {
    __shared__ float sh[16];
    float cs[SM_IT][4];
#pragma unroll
    for (int it = 0; it < SM_IT; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[it][e] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    // raw loads of the NEXT row are issued before the current row's block-wide reduction (a serial chain per row otherwise)
    struct Raw { decltype(Raw4<TY>().v) y[SM_IT]; decltype(Raw4<TG>().v) g[SM_IT]; };
    auto fetch = [&](long row) {
        Raw r;
#pragma unroll
        for (int it = 0; it < SM_IT; ++it) {
            const int c = it * 1024 + threadIdx.x * 4;
            if (c < C) {
                r.y[it] = *reinterpret_cast<const decltype(Raw4<TY>().v)*>(y + row * C + c);
                r.g[it] = *reinterpret_cast<const decltype(Raw4<TG>().v)*>(dy + row * C + c);
            }
        }
        return r;
    };
    Raw cur;
    if (r0 < r1) cur = fetch(r0);
    for (long row = r0; row < r1; ++row) {
        Raw nxt;
        if (row + 1 < r1) nxt = fetch(row + 1);
        float yv[SM_IT][4], gv[SM_IT][4];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < SM_IT; ++it) {
            const int c = it * 1024 + threadIdx.x * 4;
            if (c < C) {
                Raw4<TY> ry; ry.v = cur.y[it]; ry.get(yv[it]);
                Raw4<TG> rg; rg.v = cur.g[it]; rg.get(gv[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) s += (MODE == 0) ? gv[it][e] * yv[it][e] : gv[it][e];
            }
        }
        s = block_sum(s, sh);
#pragma unroll
        for (int it = 0; it < SM_IT; ++it) {
            const int c = it * 1024 + threadIdx.x * 4;
            if (c < C) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (MODE == 0) ? yv[it][e] * (gv[it][e] - s) : gv[it][e] - __expf(yv[it][e]) * s;
                    cs[it][e] += (float)(TO)o[e];
                }
                store4(dx + row * C + c, o);
            }
        }
        cur = nxt;
    }
    if (slab) {
#pragma unroll
        for (int it = 0; it < SM_IT; ++it) {
            const int c = it * 1024 + threadIdx.x * 4;
            if (c < C) store4(slab + (long)blockIdx.x * C + c, cs[it]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// out[n] += alpha * sum_m x[m][n]   (bias gradients).  Block = 256 threads = 64 column-quads x 4 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long M, int N, long ld,
                                                     int rows_per_block, float alpha) {
    __shared__ float sh[4][256];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + cq * 4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        for (long r = r0 + rl; r < r1; r += 4) {
            float v[4]; load4(x + r * ld + c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[rl][cq * 4 + e] = a[e];
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < N) atomicAdd(out + cc, alpha * (sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}

// Zero rows n >= len[b] of x[B][N][d] in place (attention.py:511,546-547; convolution.py:109-110).
template <typename T>
__global__ void mask_rows_kernel(T* __restrict__ x, const int* __restrict__ len, int B, int N, int d) {
    const long total = (long)B * N * (d / 4);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / (d / 4);
        const int b = (int)(row / N), n = (int)(row % N);
        if (n >= len[b]) { float z[4] = {0.f, 0.f, 0.f, 0.f}; store4(x + idx * 4, z); }
    }
}

}  // namespace

SCONF_API int sconf_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, hipStream_t stream) {
    if (n == 0) return 0;
    SCONF_REQUIRE(src_dtype != dst_dtype, "sconf_cast: same dtype");
    const int blocks = (int)std::min<long>(cdiv(n, 256 * 8), 4096);
    if (src_dtype == SCONF_F32) hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)src, (bf16*)dst, (long)n);
    else hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16*)src, (float*)dst, (long)n);
    SCONF_LAUNCH_OK("sconf_cast");
    return 0;
}

// dst (C,R) bf16 = transpose(src (R,C) f32)
SCONF_API int sconf_cast_transpose(const float* src, void* dst, int64_t R, int64_t C, hipStream_t stream) {
    if (R * C == 0) return 0;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, stream, src, (bf16*)dst, (int)R, (int)C);
    SCONF_LAUNCH_OK("sconf_cast_transpose");
    return 0;
}

// table: DEVICE array of n_entries + 1 records of 6 x int64 {src, dst, dstT, R, C, tile0} (see cast_shadows_kernel); the last
// record is the sentinel {0, 0, 0, 0, 0, total_tiles}.  Replaces the per-weight autocast casts of the reference
// (torch.autocast, training_tools.py) and their transposed variants with one launch per step.
SCONF_API int sconf_cast_shadows(const void* table, int64_t n_entries, int64_t total_tiles, hipStream_t stream) {
    static_assert(sizeof(ShadowEntry) == 48, "table record is 6 x 8 bytes");
    SCONF_REQUIRE(n_entries >= 0 && total_tiles >= 0 && total_tiles < (1L << 31), "sconf_cast_shadows: bad sizes");
    if (n_entries == 0 || total_tiles == 0) return 0;
    hipLaunchKernelGGL(cast_shadows_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, (const ShadowEntry*)table, (int)n_entries);
    SCONF_LAUNCH_OK("sconf_cast_shadows");
    return 0;
}

// apply_rotary_pos_emb (rotary_emb.py:61-73, attention.py:498-507) in place on the q and k blocks of qkv (B*N, 3, H, D) bf16 - the
// layout the qkv GEMM writes when its weight shadow is regrouped (sconf_cast_shadows, R < 0).
SCONF_API int sconf_rotary_inplace(void* qkv, const float* cos_tab, const float* sin_tab, int64_t B, int64_t N, int64_t H, int64_t D,
                                   hipStream_t stream) {
    SCONF_REQUIRE(D % 16 == 0, "sconf_rotary_inplace: head_dim %ld must be a multiple of 16", (long)D);
    const long M = B * N;
    if (M == 0) return 0;
    const long total = M * 2 * H * (D / 16);
    hipLaunchKernelGGL(rotary_inplace_kernel, dim3((unsigned)std::min<long>(cdiv(total, 256), 16384)), dim3(256), 0, stream, (bf16*)qkv, cos_tab, sin_tab, M, (int)N, (int)H, (int)D);
    SCONF_LAUNCH_OK("sconf_rotary_inplace");
    return 0;
}

// Replaces the rearrange "b n (h d qkv) -> qkv b n h d" + apply_rotary_pos_emb (attention.py:485,498-507,
// rotary_emb.py:61-73).  bwd != 0 runs the transpose: (dq,dk,dv) -> dqkv (written into `qkv`).
SCONF_API int sconf_rotary_qkv(int bwd, void* qkv, const float* cos_tab, const float* sin_tab, void* q, void* k, void* v,
                               int64_t B, int64_t N, int64_t H, int64_t D, int use_rotary, hipStream_t stream) {
    SCONF_REQUIRE(D % 16 == 0, "sconf_rotary_qkv: head_dim %ld must be a multiple of 16", (long)D);
    const long M = B * N;
    if (M == 0) return 0;
    const long total = M * H * (D / 16);
    const int blocks = (int)std::min<long>(cdiv(total, 256), 8192);
    if (!bwd) hipLaunchKernelGGL((rotary_qkv_kernel<false>), dim3(blocks), dim3(256), 0, stream, (bf16*)qkv, cos_tab, sin_tab, (bf16*)q, (bf16*)k, (bf16*)v, M, (int)N, (int)H, (int)D, use_rotary);
    else      hipLaunchKernelGGL((rotary_qkv_kernel<true>), dim3(blocks), dim3(256), 0, stream, (bf16*)qkv, cos_tab, sin_tab, (bf16*)q, (bf16*)k, (bf16*)v, M, (int)N, (int)H, (int)D, use_rotary);
    SCONF_LAUNCH_OK("sconf_rotary_qkv");
    return 0;
}

// mode 0 softmax (sconformer_xl.py:242), mode 1 log_softmax (decoder.py:25).
SCONF_API int sconf_softmax_fwd(int mode, const void* x, int x_dtype, void* y, int y_dtype, int64_t M, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 4 == 0 && C <= SM_IT * 1024 && C > 0, "sconf_softmax_fwd: C=%ld must be a multiple of 4 and <= 8192", (long)C);
    if (M == 0) return 0;
    dim3 g((unsigned)M), b(256);
#define L(TI, TO, MD) hipLaunchKernelGGL((softmax_fwd_kernel<TI, TO, MD>), g, b, 0, stream, (const TI*)x, (TO*)y, (int)C)
    if (mode == 0) {
        if (x_dtype == SCONF_BF16 && y_dtype == SCONF_BF16) L(bf16, bf16, 0);
        else if (x_dtype == SCONF_F32 && y_dtype == SCONF_BF16) L(float, bf16, 0);
        else if (x_dtype == SCONF_F32 && y_dtype == SCONF_F32) L(float, float, 0);
        else L(bf16, float, 0);
    } else {
        if (x_dtype == SCONF_BF16 && y_dtype == SCONF_BF16) L(bf16, bf16, 1);
        else if (x_dtype == SCONF_F32 && y_dtype == SCONF_BF16) L(float, bf16, 1);
        else if (x_dtype == SCONF_F32 && y_dtype == SCONF_F32) L(float, float, 1);
        else L(bf16, float, 1);
    }
#undef L
    SCONF_LAUNCH_OK("sconf_softmax_fwd");
    return 0;
}

constexpr int SOFTMAX_BWD_SLABS = 2048;             // workgroups (= column-sum slabs) of the fused softmax backward
// out[m] = sum_c a[m][c] * (b[m][c] - bias[c]): one wave per row, 8 bf16 per lane and load (d % 8 == 0).
namespace {
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, const float* __restrict__ bias,
                                                     float* __restrict__ out, long M, int d, long lda, long ldb) {
    const int lane = threadIdx.x & 63;
    for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane * 8; c < d; c += 512) {
            float x[8], y[8], z[8];
            load8(a + m * lda + c, x); load8(b + m * ldb + c, y);
            if (bias) load8(bias + c, z);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[e] * (y[e] - (bias ? z[e] : 0.f));
        }
        s = wave_sum(s);
        if (lane == 0) out[m] = s;
    }
}
}  // namespace

SCONF_API int sconf_rowdot(const void* a, const void* b, const float* bias, float* out, int64_t M, int64_t d, int64_t lda, int64_t ldb,
                           hipStream_t stream) {
    SCONF_REQUIRE(d % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "sconf_rowdot: d, lda, ldb must be multiples of 8");
    if (M == 0) return 0;
    const unsigned g = (unsigned)std::min<long>(cdiv(M, 4), 65536);
    hipLaunchKernelGGL(rowdot_kernel, dim3(g), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, bias, out, (long)M, (int)d, (long)lda, (long)ldb);
    SCONF_LAUNCH_OK("sconf_rowdot");
    return 0;
}

SCONF_API int sconf_colsum(const void* x, int x_dtype, float* out, int64_t M, int64_t N, int64_t ld, float alpha, hipStream_t stream);
// floats of scratch sconf_softmax_bwd needs when it also produces the column sums of dx
SCONF_API int64_t sconf_softmax_bwd_workspace(int64_t M, int64_t C) { return (int64_t)std::min<long>(M, SOFTMAX_BWD_SLABS) * C; }

// colsum_out (optional, f32 [C], ACCUMULATED): column sums of dx, i.e. the bias gradient of the Linear that produced the logits.
SCONF_API int sconf_softmax_bwd(int mode, const void* y, int y_dtype, const void* dy, int dy_dtype, void* dx, int dx_dtype,
                                float* colsum_out, float* workspace, int64_t M, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 4 == 0 && C <= SM_IT * 1024 && C > 0, "sconf_softmax_bwd: C=%ld must be a multiple of 4 and <= 8192", (long)C);
    SCONF_REQUIRE(dx_dtype == SCONF_BF16 || dx_dtype == SCONF_F32, "sconf_softmax_bwd: bad dx dtype");
    if (M == 0) return 0;
    SCONF_REQUIRE(!colsum_out || workspace, "sconf_softmax_bwd: colsum_out needs the workspace (sconf_softmax_bwd_workspace floats)");
    const int rpb = colsum_out ? (int)cdiv(M, SOFTMAX_BWD_SLABS) : 1;
    dim3 g((unsigned)cdiv(M, rpb)), b(256);
    float* slab = colsum_out ? workspace : nullptr;
#define L(TY, TG, TO, MD) hipLaunchKernelGGL((softmax_bwd_kernel<TY, TG, TO, MD>), g, b, 0, stream, (const TY*)y, (const TG*)dy, (TO*)dx, slab, (long)M, (int)C, rpb)
#define D3(MD) \
    if (y_dtype == SCONF_BF16 && dy_dtype == SCONF_BF16) { if (dx_dtype == SCONF_BF16) L(bf16, bf16, bf16, MD); else L(bf16, bf16, float, MD); } \
    else if (y_dtype == SCONF_BF16) { if (dx_dtype == SCONF_BF16) L(bf16, float, bf16, MD); else L(bf16, float, float, MD); } \
    else if (dy_dtype == SCONF_BF16) { if (dx_dtype == SCONF_BF16) L(float, bf16, bf16, MD); else L(float, bf16, float, MD); } \
    else { if (dx_dtype == SCONF_BF16) L(float, float, bf16, MD); else L(float, float, float, MD); }
    if (mode == 0) { D3(0) } else { D3(1) }
#undef D3
#undef L
    SCONF_LAUNCH_OK("sconf_softmax_bwd");
    if (colsum_out) return sconf_colsum(slab, SCONF_F32, colsum_out, (int64_t)g.x, C, C, 1.f, stream);     // += over the slabs
    return 0;
}

// out[n] += sum over rows (bias gradients of Linear / Conv layers).
SCONF_API int sconf_colsum(const void* x, int x_dtype, float* out, int64_t M, int64_t N, int64_t ld, float alpha, hipStream_t stream) {
    SCONF_REQUIRE(N % 4 == 0 && ld % 4 == 0, "sconf_colsum: N and ld must be multiples of 4");
    if (M == 0 || N == 0) return 0;
    const int cb = cdiv(N, 256);
    int rb = std::max(1, std::min(cdiv(M, 64), 2048 / cb));
    const int rpb = cdiv(M, rb);
    rb = cdiv(M, rpb);
    dim3 g(cb, rb), b(256);
    if (x_dtype == SCONF_BF16) hipLaunchKernelGGL((colsum_kernel<bf16>), g, b, 0, stream, (const bf16*)x, out, (long)M, (int)N, (long)ld, rpb, alpha);
    else hipLaunchKernelGGL((colsum_kernel<float>), g, b, 0, stream, (const float*)x, out, (long)M, (int)N, (long)ld, rpb, alpha);
    SCONF_LAUNCH_OK("sconf_colsum");
    return 0;
}

SCONF_API int sconf_mask_rows(void* x, int dtype, const int32_t* lengths, int64_t B, int64_t N, int64_t d, hipStream_t stream) {
    SCONF_REQUIRE(d % 4 == 0, "sconf_mask_rows: d must be a multiple of 4");
    const long total = B * N * (d / 4);
    if (total == 0) return 0;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 8192);
    if (dtype == SCONF_BF16) hipLaunchKernelGGL((mask_rows_kernel<bf16>), dim3(blocks), dim3(256), 0, stream, (bf16*)x, lengths, (int)B, (int)N, (int)d);
    else hipLaunchKernelGGL((mask_rows_kernel<float>), dim3(blocks), dim3(256), 0, stream, (float*)x, lengths, (int)B, (int)N, (int)d);
    SCONF_LAUNCH_OK("sconf_mask_rows");
    return 0;
}
