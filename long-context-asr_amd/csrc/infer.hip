// Forward-only inference helpers for the sliding-window transcription path:
//   * overlap-average of window posteriors   (lcasr/eval/utils.py:45-111, fetch_logits)
//   * row argmax for greedy CTC decoding     (lcasr/decoding/greedy.py:9-23)
// All HBM-bound; one pass over the window log-probs.
#include "common.h"
#include <algorithm>

namespace {

// Gather form of the reference's running "+= exp(logits)" over overlapping windows: output row r sums the rows r - pos_w
// of every window w that covers it, so overlapping windows need neither atomics nor a serial loop over windows and the
// summation order per element is fixed (window index ascending = the reference's order).
//   logp : (W, n, C) f32 log-probs of W equally long windows; window w starts at output row pos0 + w * stride.
//   acc  : (N, C) f32, count : (N) f32 - both ACCUMULATED (+=), so a ragged last window is a second call with W = 1.
__global__ __launch_bounds__(256) void overlap_add_exp_kernel(const float* __restrict__ logp, float* __restrict__ acc,
                                                             float* __restrict__ count, long W, long n, int C, long stride,
                                                             long pos0, long N) {
    const long span = (W - 1) * stride + n;                    // rows touched: [pos0, pos0 + span)
    const int cq = C / 4;
    const long total = span * cq;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long rr = idx / cq;                              // row relative to pos0
        const int c = (int)(idx - rr * cq) * 4;
        const long r = pos0 + rr;
        if (r >= N) continue;
        // windows covering rr: w*stride <= rr < w*stride + n
        long w_hi = rr / stride; if (w_hi > W - 1) w_hi = W - 1;
        long w_lo = rr - n + 1 <= 0 ? 0 : (rr - n + stride) / stride;      // ceil((rr - n + 1) / stride)
        float a[4]; load4(acc + r * C + c, a);
        int cnt = 0;
        for (long w = w_lo; w <= w_hi; ++w) {
            float v[4]; load4(logp + (w * n + (rr - w * stride)) * C + c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += __expf(v[e]);
            ++cnt;
        }
        store4(acc + r * C + c, a);
        if (c == 0) count[r] += (float)cnt;
    }
}

// out[r][c] = log(acc[r][c] / count[r])   for the rows with count > 0 (a prefix of the buffer: windows are contiguous)
__global__ __launch_bounds__(256) void overlap_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ count,
                                                              float* __restrict__ out, long N, int C) {
    const int cq = C / 4;
    const long total = N * cq;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cq;
        const int c = (int)(idx - r * cq) * 4;
        const float k = count[r];
        float a[4]; load4(acc + r * C + c, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = __logf(a[e] / k);
        store4(out + r * C + c, a);
    }
}

// idx[m] = argmax_c x[m][c], first index on ties (torch.argmax on CPU/GPU returns the first maximal index for our inputs:
// NaNs aside).  One wave per row.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int* __restrict__ idx, long M, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane * 4; c < C; c += 256) {
        float v[4]; load4(xr + c, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (v[e] > best || (v[e] == best && c + e < bi)) { best = v[e]; bi = c + e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) idx[row] = bi == 0x7fffffff ? 0 : bi;
}

}  // namespace

// acc (N,C) += sum over the windows covering each row of exp(logp); count (N) += number of covering windows.
// Replaces the per-window loop body of fetch_logits (lcasr/eval/utils.py:96-104) for a whole batch of windows.
SCONF_API int sconf_overlap_add_exp(const float* logp, int64_t W, int64_t n, int64_t C, int64_t stride, int64_t pos0,
                                    float* acc, float* count, int64_t N, hipStream_t stream) {
    SCONF_REQUIRE(C % 4 == 0 && C > 0, "sconf_overlap_add_exp: C=%ld must be a positive multiple of 4", (long)C);
    SCONF_REQUIRE(W >= 0 && n >= 0 && pos0 >= 0 && N >= 0, "sconf_overlap_add_exp: negative size");
    if (W == 0 || n == 0) return 0;
    SCONF_REQUIRE(W == 1 || stride > 0, "sconf_overlap_add_exp: stride must be positive for more than one window");
    SCONF_REQUIRE(pos0 + (W - 1) * stride + n <= N, "sconf_overlap_add_exp: windows run past the %ld-row buffer", (long)N);
    if (W == 1) stride = n;                                     // unused, but keeps the divisions defined
    const long total = ((W - 1) * stride + n) * (C / 4);
    const int blocks = (int)std::min<long>(cdiv(total, 256), 16384);
    hipLaunchKernelGGL(overlap_add_exp_kernel, dim3(blocks), dim3(256), 0, stream, logp, acc, count, (long)W, (long)n, (int)C,
                       (long)stride, (long)pos0, (long)N);
    SCONF_LAUNCH_OK("sconf_overlap_add_exp");
    return 0;
}

// out = log(acc / count) over the first N rows (all of which must have count > 0): fetch_logits, utils.py:107-111.
SCONF_API int sconf_overlap_finalize(const float* acc, const float* count, float* out, int64_t N, int64_t C, hipStream_t stream) {
    SCONF_REQUIRE(C % 4 == 0 && C > 0, "sconf_overlap_finalize: C=%ld must be a positive multiple of 4", (long)C);
    if (N == 0) return 0;
    const int blocks = (int)std::min<long>(cdiv(N * (C / 4), 256), 16384);
    hipLaunchKernelGGL(overlap_finalize_kernel, dim3(blocks), dim3(256), 0, stream, acc, count, out, (long)N, (int)C);
    SCONF_LAUNCH_OK("sconf_overlap_finalize");
    return 0;
}

// idx[m] = argmax over the C columns of row m (first index on ties): GreedyCTCDecoder.forward, greedy.py:19.
SCONF_API int sconf_argmax_rows(const float* x, int64_t M, int64_t C, int32_t* idx, hipStream_t stream) {
    SCONF_REQUIRE(C % 4 == 0 && C > 0, "sconf_argmax_rows: C=%ld must be a positive multiple of 4", (long)C);
    if (M == 0) return 0;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, stream, x, idx, (long)M, (int)C);
    SCONF_LAUNCH_OK("sconf_argmax_rows");
    return 0;
}
