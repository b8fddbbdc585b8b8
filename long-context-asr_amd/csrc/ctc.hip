// CTC loss (sum reduction, arbitrary blank) forward + backward in log space, replacing
// torch.nn.CTCLoss(blank=num_classes-1, reduction='sum') on the (N,B,C) view of final_posteriors
// (exp/train.py:104,249).  Batch-major (B,N,C) f32 log-probs are consumed directly.
//
// Structure (the recursion is serial in time with only B-way batch parallelism):
//   1. gather   lpg[b][t][s] = lp[b][t][l'_s]        fully parallel; makes the serial pass read contiguous rows
//   2. alpha/beta  one workgroup per (sample, direction): 2B workgroups run concurrently; the lattice row lives
//                  in LDS (double-buffered, one barrier per step); lpg rows are prefetched 4 steps ahead into
//                  registers so the serial chain never waits on HBM.  beta is alpha on the mirrored lattice.
//                  Round 3 - precision at the 131072-frame context.  Plain log-space f32 rows reach |alpha| ~ 1e5 at 16384
//                  frames: one ulp is 0.008 there and the roundings of the serial chain add up to a 0.4-0.5 drift of the
//                  per-frame gradient sums and a 0.24 relative L2 error of the gradient against an f64 lattice - torch's own f32
//                  op has exactly that error (measured).  Subtracting the row maximum is not enough: with weak emissions the
//                  prefix probabilities peak hundreds of states ahead of the states the posterior lives on, which then sit
//                  hundreds to thousands below the maximum (measured: 6e-3 gradient error left at 16384 frames).  So the STATE of
//                  the recursion (the LDS rows) is f64 and only the transcendental part runs in f32:
//                      new = m + (double) logf(expf(a - m) + expf(b - m) + expf(c - m)) + emission,   m = max(a, b, c) in f64
//                  - an absolute error of ~1e-7 per step whatever |alpha| is.  What is STORED for the gradient pass stays f32:
//                  alpha_t - A_t, with A_t (f64, one scalar per frame, `offs`) tracking the maximum of the previous stored row, so
//                  the stored values are small where it matters and their rounding is a one-time error, not an accumulated one;
//                  the large parts cancel in f64 once per frame (A_t + B_t + nll).  The row maximum costs no barrier: wave maxima
//                  are published in LDS in front of the step's own barrier and used one step late.  (Lattices of more than
//                  10230 states do not fit the LDS in f64 and run the same code with f32 state.)
//   3. grad     one workgroup per (b,t) row: occupancy scattered into an LDS histogram over classes, then
//               grad = g * (exp(lp) - occupancy)   [ATen convention; zero for t >= input_length].
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

#ifdef CTC_STAMP
__device__ unsigned long long g_ctc_stamps[16 * 8];
#define CSTAMP_DECL unsigned long long st_acc_[8] = {}, st_last_ = 0; { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_last_ = t_; }
#define CSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); st_acc_[i] += t_ - st_last_; st_last_ = t_; } while (0)
#define CSTAMP_OUT do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) g_ctc_stamps[(threadIdx.x >> 6) * 8 + i_] = st_acc_[i_]; } while (0)
#else
#define CSTAMP_DECL
#define CSTAMP(i)
#define CSTAMP_OUT
#endif

// log(e^a + e^b + e^c): the largest term contributes exactly 1, so only the median and the minimum need an exponential
// (v_max3 / v_med3 / v_min3 pick them without branches) - 2 exp + 1 log on the serial chain instead of 3 + 1.
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    const float md = __builtin_amdgcn_fmed3f(a, b, c), mn = fminf(a, fminf(b, c));
    return m + __logf(1.f + __expf(md - m) + __expf(mn - m));
}
// The f64-state form works in LOG2 units (v_exp_f32 / v_log_f32 are base-2: no multiplies, and none of the denormal-range fix-ups
// of __expf / __logf - the arguments are <= 0 and the sum is in [1, 3]).  Maximum and differences in f64, the transcendental part in
// f32: ~1e-7 absolute whatever the magnitude of a, b, c.  "Unreachable" is NEG_BIG (finite), not -inf: three unreachable inputs give
// NEG_BIG + log2(3) = NEG_BIG (absorbed) without a special case, and (float)NEG_BIG = -inf is what the stored rows show.
constexpr double NEG_BIG = -1e300;
constexpr double LOG2E_D = 1.4426950408889634, LN2_D = 0.6931471805599453;
__device__ __forceinline__ double max_f64(double a, double b) {          // fmax() canonicalises its inputs first (3 instructions)
    double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ double max_lt(double a, double b) { return max_f64(a, b); }
__device__ __forceinline__ float max_lt(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double lse3_log2(double a, double b, double c) {
    const double m = max_f64(a, max_f64(b, c));
    const float s = __builtin_amdgcn_exp2f((float)(a - m)) + __builtin_amdgcn_exp2f((float)(b - m)) + __builtin_amdgcn_exp2f((float)(c - m));
    return m + (double)__builtin_amdgcn_logf(s);
}
__device__ __forceinline__ float lse3_log2(float a, float b, float c) {  // f32 state (lattices too long for the LDS in f64)
    const float m = fmaxf(a, fmaxf(b, c));
    return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m));
}
// Maximum over the wave by DPP (no LDS round trips): quad swaps, half-row and row mirrors, then the four row maxima by readlane.
__device__ __forceinline__ float wave_max_dpp(float v) {
    int x;
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);  v = fmaxf(v, __int_as_float(x));   // quad_perm [1,0,3,2]
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);  v = fmaxf(v, __int_as_float(x));   // quad_perm [2,3,0,1]
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true); v = fmaxf(v, __int_as_float(x));   // row_half_mirror
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true); v = fmaxf(v, __int_as_float(x));   // row_mirror
    const int iv = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}

// One workgroup per (sample, frame): the frame's whole log-prob row is streamed into LDS with 16-byte loads and the 2S+1 lattice
// emissions are gathered from there (a thread-per-emission gather from global memory touched every 64-byte sector of the row
// for 4 useful bytes each: 1.68 ms at B = 128, N = 2048, C = 4096 against 0.9 ms for row + lattice bytes at the HBM rate).
// LOGITS: `lp` holds raw logits; the row's log-sum-exp is computed from the LDS copy, written to lse[b][t], and the emissions are
// logit - lse - the log_softmax pass (and its (B,N,C) output) is then not needed for the loss (sconf_ctc_fwd_logits).
template <bool LOGITS>
__global__ __launch_bounds__(256) void ctc_gather_kernel(const float* __restrict__ lp, const int* __restrict__ targets, const int* __restrict__ in_len,
                                                         const int* __restrict__ tg_len, float* __restrict__ lpg, float* __restrict__ lse_out,
                                                         int B, int N, int C, int Smax, int Lmax, int blank) {
    extern __shared__ float row[];                      // [C]
    __shared__ float red[16];
    for (long bt = blockIdx.x; bt < (long)B * N; bt += gridDim.x) {
        const int t = (int)(bt % N), b = (int)(bt / N);
        float* out = lpg + bt * Lmax;
        const int L = 2 * tg_len[b] + 1;
        if (t >= in_len[b]) {                           // frames past the sample's length: zeros (never read by the lattice)
            for (int s = threadIdx.x; s < Lmax; s += 256) out[s] = 0.f;
            if (LOGITS && threadIdx.x == 0) lse_out[bt] = 0.f;
            continue;
        }
        __syncthreads();                                // the previous row's gathers are done
        const float* src = lp + bt * C;
        float mx = -INFINITY;
        for (int c = threadIdx.x * 4; c < C; c += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(src + c);
            *reinterpret_cast<float4*>(row + c) = v;
            if (LOGITS) mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
        float lse = 0.f;
        if (LOGITS) {
            mx = block_max(mx, red);                    // (its barriers also publish the row)
            float sm = 0.f;
            for (int c = threadIdx.x * 4; c < C; c += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(row + c);
                sm += __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx);
            }
            lse = mx + __logf(block_sum(sm, red));
            if (threadIdx.x == 0) lse_out[bt] = lse;
        } else __syncthreads();
        for (int s = threadIdx.x; s < Lmax; s += 256) {
            float v = 0.f;
            if (s < L) {
                int lab = (s & 1) ? targets[(long)b * Smax + (s >> 1)] : blank;
                lab = min(max(lab, 0), C - 1);          // an out-of-range label poisons the sample (alpha/beta kernel); never index with it
                v = row[lab] - lse;
            }
            out[s] = v;
        }
    }
}

// LDS-DMA of 4 bytes per lane through a buffer descriptor, as inline asm: the builtin makes hipcc wait for EVERY pending DMA
// (vmcnt(0)) in front of the next LDS read of any kind (attention.hip has the long story).  lds_base (uniform) + 4 * lane <- srd[voff].
typedef __amdgpu_buffer_rsrc_t ctc_srd_t;
__device__ __forceinline__ ctc_srd_t ctc_make_srd(const void* base, long nbytes) {
    const unsigned long a = (unsigned long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)(nbytes > 0xffffffffL ? 0xffffffffL : nbytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long)hi << 32) | lo), 0, nb, 0x00020000);
}
__device__ __forceinline__ void ctc_dma_dword(ctc_srd_t srd, unsigned voff, unsigned soff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %3, %4, %1 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "+s"(soff), "+s"(lds_base) : "v"(voff), "s"(srd) : "memory");
}

// One workgroup per (sample, direction).  MAXS = max lattice states per thread; LT = type of the recursion's state (double; float
// only for lattices that do not fit the LDS in f64).  The recursion runs in log2 units relative to nothing (absolute, in LT); what
// is STORED for the gradient pass is natural-log, f32, relative to the frame's offset A_t (f64, written to offs), which follows the
// maximum of the stored row every 4th frame.  The serial step is: LDS reads of three neighbours -> lse3 -> LDS write -> barrier.
//  * the barrier is a raw s_barrier behind `s_waitcnt lgkmcnt(0)`: __syncthreads() also waits for the frame's global stores to be
//    acknowledged (vmcnt(0)), a memory round trip per frame that nothing depends on;
//  * a wave whose states are all unreachable at this frame (s > 2i + 1) or can no longer reach the end (s < L - 2 (T - i)) skips the
//    arithmetic and stores -inf (a quarter of the lattice at T = 4 S).
template <int MAXS, typename LT>
__global__ __launch_bounds__(1024) void ctc_alphabeta_kernel(const float* __restrict__ lpg, const int* __restrict__ targets,
                                                             const int* __restrict__ in_len, const int* __restrict__ tg_len,
                                                             float* __restrict__ alpha, float* __restrict__ beta,
                                                             float* __restrict__ nll, double* __restrict__ offs,
                                                             int B, int N, int C, int Smax, int Lmax, int blank) {
    extern __shared__ double lat_raw[];                 // LT [2][W]; then float [16] wave maxima
    LT* lat = reinterpret_cast<LT*>(lat_raw);
    const int b = blockIdx.x % B;
    const bool is_beta = blockIdx.x >= B;
    const int T = in_len[b], S = tg_len[b], L = 2 * S + 1;
    float* out = (is_beta ? beta : alpha) + (long)b * N * Lmax;
    double* off_out = offs + ((long)(is_beta ? B : 0) + b) * N;     // [alpha offsets (B,N) | beta offsets (B,N) | nll in f64 (B)]
    double* nll64 = offs + 2L * B * N;
    const float* lg = lpg + (long)b * N * Lmax;
    const int nt = blockDim.x, tid = threadIdx.x;
    const int W = Lmax + MAXS + 2;                      // two guard cells in front, MAXS cells of slack behind
    float* wm = reinterpret_cast<float*>(lat_raw + (((size_t)2 * W * sizeof(LT) / 8 + 2) & ~(size_t)1));   // 16-byte aligned, behind the rows
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LT NEG = sizeof(LT) == 8 ? (LT)NEG_BIG : (LT)-1e30f;
    if (T <= 0) { if (!is_beta && tid == 0) { nll[b] = INFINITY; nll64[b] = INFINITY; } return; }
    // Inputs torch.nn.CTCLoss rejects on the host (input_length > N, target_length > Smax, a label outside [0, C)): the lengths
    // live on the device here, so the sample is poisoned instead - nll = NaN, its gradient rows NaN (ctc_grad_kernel), which the
    // optimiser's non-finite check turns into a skipped step - and nothing is indexed with the bad value.
    {
        int bad = (T > N) | (S < 0) | (S > Smax);
        if (!bad) for (int i = tid; i < S; i += nt) { const int lab = targets[(long)b * Smax + i]; bad |= (lab < 0) | (lab >= C); }
        if (__syncthreads_or(bad)) { if (!is_beta && tid == 0) { nll[b] = NAN; nll64[b] = NAN; } return; }
    }

    // A thread owns MAXS ADJACENT states, in MIRRORED coordinates sp (beta walks the reversed lattice): its MAXS + 2 inputs are one
    // contiguous LDS read, a wave covers one contiguous range (so the band test is per wave), and the MAXS recursions are
    // independent straight-line chains the compiler interleaves (the step is bound by the LATENCY of one chain - LDS read, f64
    // max / differences, three exponentials, a logarithm, f64 adds, LDS write - not by issue).
    const int sp0 = tid * MAXS, cnt = min(max(L - sp0, 0), MAXS);
    bool skip_ok[MAXS];
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
        const int sp = sp0 + k;
        skip_ok[k] = false;
        if (sp < L && sp >= 2) {
            const int s = is_beta ? L - 1 - sp : sp;
            if (s & 1) {
                const int s2 = is_beta ? s + 2 : s - 2;
                skip_ok[k] = targets[(long)b * Smax + (s >> 1)] != targets[(long)b * Smax + (s2 >> 1)];
            }
        }
    }
    for (int i = tid; i < 2 * W; i += nt) lat[i] = NEG;
    if (tid < 16) wm[tid] = -INFINITY;                   // slots of waves that do not exist stay -inf
    __syncthreads();
    if (tid == 0) lat[W + 2] = (LT)0;                    // virtual frame -1 (row 1): all mass in state 0, so frame 0 is the general step
    __syncthreads();

    // Frames are prefetched in groups of G (2 * G * MAXS registers): 4 at a few states per thread, 1 for the long lattices, whose
    // frames are slow enough to cover a load by themselves (with MAXS = 16 and groups of 4 the prefetch alone was 128 VGPRs of
    // the 128 a 1024-thread workgroup has: the kernel lived in scratch - 282 ms at N = 16384, S = 4096).
    constexpr int G = MAXS <= 4 ? 4 : 1;
    constexpr int CH = MAXS <= 6 ? MAXS : (MAXS <= 12 ? MAXS / 2 : 4);   // recursions interleaved at a time (stage by stage)
    constexpr bool COOP = MAXS > 4;                      // stored rows written cooperatively from the LDS row (below) instead of by the owning threads
    constexpr bool LONG = MAXS > 4;                      // emissions by LDS-DMA into a staging row (below); implies G == 1 ... 2 is folded to 1
    float pf[G][MAXS], nx[G][MAXS];
    // Unconditional loads from clamped (always valid) addresses: a load inside a branch is waited for inside that branch (vmcnt(0),
    // in order behind every store in flight) - eight serial memory round trips per group instead of a prefetch.  What the clamped
    // loads bring for frames >= T or states >= L is never used.
    auto load_group = [&](float (&dst)[G][MAXS], int i0) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int i = min(i0 + j, T - 1);
            const int t = is_beta ? T - 1 - i : i;
#pragma unroll
            for (int k = 0; k < MAXS; ++k) {
                const int sp = min(sp0 + k, L - 1);
                const int s = is_beta ? L - 1 - sp : sp;
                dst[j][k] = lg[(long)t * Lmax + s];
            }
        }
    };
    // Long lattices (LONG): the emissions arrive by LDS-DMA, one frame ahead, instead of as MAXS strided dword loads per thread and
    // frame (which, with the row stores, filled the memory pipeline: 3.4 k of 12 k cycles per frame went into ISSUING them).  Every
    // blank state has the same emission, so what a wave needs is [blank, the labels of its own state range]: each wave stages exactly
    // that into a segment of its own - its vmcnt(0) covers everything it reads, and no barrier orders the DMA against the reads.
    constexpr int SEGN = 64 * (MAXS / 2 + 1);            // floats per wave segment: 1 + 32 MAXS (+ 1) entries, whole pieces of 64
    float* stage = reinterpret_cast<float*>(reinterpret_cast<char*>(wm) + 64) + wave * SEGN;
    const unsigned stage_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)stage;
    // the wave's states in natural coordinates, and its first label
    const int wsp0 = wave * 64 * MAXS, wsp1 = min(wsp0 + 64 * MAXS - 1, L - 1);
    const int s_lo = is_beta ? L - 1 - wsp1 : wsp0, jl0 = max(s_lo, 0) >> 1;
    auto dma_frame = [&](int i) {                          // frame i's emissions of this wave -> its segment
        if (wsp0 >= L) return;
        const int Tu = __builtin_amdgcn_readfirstlane(T), iu = __builtin_amdgcn_readfirstlane(i);
        const int t = is_beta ? Tu - 1 - min(iu, Tu - 1) : min(iu, Tu - 1);
        const ctc_srd_t esrd = ctc_make_srd(lg, (long)N * Lmax * 4);
#pragma unroll
        for (int pc = 0; pc < MAXS / 2 + 1; ++pc) {
            const int e = pc * 64 + (tid & 63);            // entry 0: the blank; entry e: label jl0 + e - 1
            const int src = e == 0 ? 0 : 2 * min(jl0 + e - 1, max(S - 1, 0)) + 1;
            ctc_dma_dword(esrd, (unsigned)(((long)t * Lmax + min(src, Lmax - 1)) * 4), 0u,        // (row offset in the per-lane part: < 4 GB per sample)
                          (unsigned)__builtin_amdgcn_readfirstlane((int)(stage_lds + (unsigned)(pc * 256))));
        }
    };
    auto read_stage = [&](float (&dst)[MAXS]) {
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int sp = min(sp0 + k, L - 1), s_ = is_beta ? L - 1 - sp : sp;
            dst[k] = fmaxf(stage[(s_ & 1) ? 1 + (s_ >> 1) - jl0 : 0], -1e30f);
        }
    };
    if constexpr (LONG) {
        dma_frame(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        read_stage(pf[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_frame(1);
    } else {
        load_group(pf, 0);
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < MAXS; ++k) pf[j][k] = fmaxf(pf[j][k], -1e30f);
    }
    double A = 0.0;                                      // offset of the stored rows, log2 units
    CSTAMP_DECL
    const int w0 = wave * 64 * MAXS, w1 = w0 + 64 * MAXS - 1;   // the wave's range of states
    for (int i0 = 0; i0 < T; i0 += G) {
        if constexpr (!LONG) load_group(nx, i0 + G);     // prefetch the next G time steps
        CSTAMP(4);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int i = i0 + j;
            if (i < T) {                                  // uniform across the workgroup
                if (i > 0 && (i & 3) == 0) {              // the offset follows the maximum of the row stored at frame i - 1: one LDS round trip
                    const float4* pm = reinterpret_cast<const float4*>(wm);
                    const float4 m0 = pm[0], m1 = pm[1], m2 = pm[2], m3 = pm[3];
                    const float m = fmaxf(fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w))),
                                          fmaxf(fmaxf(fmaxf(m2.x, m2.y), fmaxf(m2.z, m2.w)), fmaxf(fmaxf(m3.x, m3.y), fmaxf(m3.z, m3.w))));
                    if (m > -1e30f) A += (double)m;
                }
                const int t = is_beta ? T - 1 - i : i;
                LT* cur = lat + (i & 1) * W + 2;
                const LT* prev = lat + ((i & 1) ^ 1) * W + 2;
                if (tid == 0) off_out[t] = A * LN2_D;
                const int hi = 2 * i + 1, lo = L - 2 * (T - i);       // live band of (mirrored) states at this frame
                float mine = -INFINITY;
                float* orow = out + (long)t * Lmax;
                if (w0 <= hi && w1 >= lo) {                           // wave-uniform
                    if (cnt > 0) {
                        // CH recursions at a time, stage by stage (sched_barrier pins the order): every stage is CH independent
                        // instructions, so a chain's latency (f64 max / differences, exponentials, logarithm) is covered by its siblings.
#pragma unroll
                        for (int k0 = 0; k0 < MAXS; k0 += CH) {
                            LT pv[CH + 2];
#pragma unroll
                            for (int q = 0; q < CH + 2; ++q) pv[q] = prev[sp0 + k0 - 2 + q];
                            if (k0 == 0) CSTAMP(0);
                            LT c[CH], m[CH], base[CH];
                            float e0[CH], e1[CH], e2[CH];
#pragma unroll
                            for (int k = 0; k < CH; ++k) if (k0 + k < MAXS) { c[k] = skip_ok[k0 + k] ? pv[k] : NEG; m[k] = max_lt(pv[k + 2], pv[k + 1]); }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < CH; ++k) if (k0 + k < MAXS) m[k] = max_lt(m[k], c[k]);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < CH; ++k) if (k0 + k < MAXS) {
                                e0[k] = (float)(pv[k + 2] - m[k]); e1[k] = (float)(pv[k + 1] - m[k]); e2[k] = (float)(c[k] - m[k]);
                                base[k] = sizeof(LT) == 8 ? (LT)__builtin_fma((double)pf[j][k0 + k], LOG2E_D, (double)m[k]) : (LT)(m[k] + pf[j][k0 + k] * (float)LOG2E_D);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < CH; ++k) if (k0 + k < MAXS) { e0[k] = __builtin_amdgcn_exp2f(e0[k]); e1[k] = __builtin_amdgcn_exp2f(e1[k]); e2[k] = __builtin_amdgcn_exp2f(e2[k]); }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < CH; ++k) if (k0 + k < MAXS) e0[k] = __builtin_amdgcn_logf(e0[k] + e1[k] + e2[k]);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < CH; ++k) if (k0 + k < MAXS) {
                                const LT v = base[k] + (LT)e0[k];
                                const float rel = (float)((double)v - A); // log2 units, relative to the frame's offset
                                if (k0 + k < cnt) {
                                    cur[sp0 + k0 + k] = v;
                                    if (!COOP) orow[is_beta ? L - 1 - sp0 - k0 - k : sp0 + k0 + k] = rel * (float)LN2_D;
                                    mine = fmaxf(mine, rel);
                                }
                            }
                        }
                        CSTAMP(1);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < MAXS; ++k)
                        if (k < cnt) { cur[sp0 + k] = NEG; if (!COOP) orow[is_beta ? L - 1 - sp0 - k : sp0 + k] = -INFINITY; }
                }
                if ((i & 3) == 3) {
                    mine = wave_max_dpp(mine);
                    if ((tid & 63) == 0) wm[wave] = mine;
                }
                CSTAMP(2);
                if constexpr (LONG) {                      // next frame's emissions (DMA issued a frame ago by this wave), then the frame after
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    read_stage(nx[0]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the segment has been read: it may be overwritten
                    dma_frame(i + 2);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                CSTAMP(3);
                // Long lattices (COOP): the row leaves for the gradient pass from its LDS copy, lanes on consecutive states (a thread's own MAXS states sit
                // 4 MAXS bytes apart from its neighbour's: ten strided dword stores per thread and frame filled the memory pipeline of the
                // 8 k-state lattices).  Row i & 1 is not written again before the barrier of frame i + 1.
                if (COOP) {                                 // (all LDS reads first: as a rolled loop each of the <= MAXS rounds waited for its own read)
                    LT rv[MAXS];
#pragma unroll
                    for (int k = 0; k < MAXS; ++k) rv[k] = cur[min(tid + k * nt, L - 1)];
#pragma unroll
                    for (int k = 0; k < MAXS; ++k) {
                        const int s = tid + k * nt;
                        if (s < L) orow[is_beta ? L - 1 - s : s] = (float)((double)rv[k] - A) * (float)LN2_D;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < MAXS; ++k) pf[j][k] = fmaxf(nx[j][k], -1e30f);            // an emission of -inf must stay finite in the recursion (LONG: clamped already)
        CSTAMP(5);
    }
    CSTAMP_OUT;
    if constexpr (LONG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no staging DMA in flight when the waves end
    if (!is_beta && tid == 0) {
        const LT* last = lat + ((T - 1) & 1) * W + 2;
        const double v = -(double)lse3_log2(last[L - 1], L > 1 ? last[L - 2] : NEG, NEG) * LN2_D;
        const bool none = v > (sizeof(LT) == 8 ? 1e290 : 1e29);          // no alignment reaches the end: +inf, as torch's zero_infinity=False
        nll64[b] = none ? (double)INFINITY : v;
        nll[b] = none ? INFINITY : (float)v;
    }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ lp, const float* __restrict__ lpg,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const float* __restrict__ nll, const double* __restrict__ offs,
                                                       const int* __restrict__ targets,
                                                       const int* __restrict__ in_len, const int* __restrict__ tg_len,
                                                       const float* __restrict__ grad_out, float* __restrict__ grad,
                                                       int B, int N, int C, int Smax, int Lmax, int blank) {
    extern __shared__ float occ[];                      // [C]
    const long bt = blockIdx.x;
    const int b = (int)(bt / N), t = (int)(bt % N);
    float* gr = grad + bt * C;
    if (t >= in_len[b]) {
        for (int c = threadIdx.x * 4; c < C; c += 1024) { float z[4] = {0.f, 0.f, 0.f, 0.f}; store4(gr + c, z); }
        return;
    }
    const float nl = nll[b];
    if (!(nl < INFINITY)) {
        // No alignment fits (nll = +inf, zero_infinity=False): torch's backward evaluates exp(-inf + inf - lp) for every class
        // of the sample's frames, i.e. the whole row is NaN - keep that signal instead of a half-valid gradient.  Also the
        // poisoned samples (nll = NaN: invalid lengths / labels), whose lattice must not be walked.
        for (int c = threadIdx.x * 4; c < C; c += 1024) { float z[4] = {NAN, NAN, NAN, NAN}; store4(gr + c, z); }
        return;
    }
    for (int c = threadIdx.x; c < C; c += 256) occ[c] = 0.f;
    __syncthreads();
    const int L = 2 * tg_len[b] + 1;
    const long base = bt * Lmax;
    // alpha, beta are stored relative to their per-frame offsets: the large parts cancel here, once per frame, in f64
    const float adj = (float)(offs[bt] + offs[(long)B * N + bt] + offs[2L * B * N + b]);
    for (int s = threadIdx.x; s < L; s += 256) {
        const int lab = (s & 1) ? targets[(long)b * Smax + (s >> 1)] : blank;
        atomicAdd(&occ[lab], __expf(alpha[base + s] + beta[base + s] - lpg[base + s] + adj));
    }
    __syncthreads();
    const float g = grad_out ? grad_out[b] : 1.f;          // per-sample upstream gradient
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float v[4]; load4(lp + bt * C + c, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = g * (__expf(v[e]) - occ[c + e]);
        store4(gr + c, v);
    }
}

// Gradient with respect to the LOGITS in one pass (the CTC gradient followed by the log_softmax backward):
//   dlogit[c] = g (softmax[c] S - occupancy[c]),   S = sum_c occupancy[c]   (= 1 up to rounding for a feasible alignment)
// [ATen: dlogp = g (exp(lp) - occ); log_softmax backward: dlogit = dlogp - exp(lp) sum_c dlogp, and sum_c dlogp = g (1 - S)].
// bf16 output (the operand of the decoder's dgrad / wgrad GEMMs); `slab` (optional, [gridDim.x][C] f32): the workgroup's column sums
// of what it stored - the decoder bias gradient.  A workgroup walks rows_per_block consecutive (b, t) rows.
__global__ __launch_bounds__(256) void ctc_grad_logits_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                              const float* __restrict__ lpg, const float* __restrict__ alpha,
                                                              const float* __restrict__ beta, const float* __restrict__ nll,
                                                              const double* __restrict__ offs,
                                                              const int* __restrict__ targets, const int* __restrict__ in_len,
                                                              const int* __restrict__ tg_len, const float* __restrict__ grad_out,
                                                              bf16* __restrict__ dlogits, float* __restrict__ slab,
                                                              long rows, int B, int N, int C, int Smax, int Lmax, int blank, int rows_per_block) {
    extern __shared__ float occ[];                      // [C]
    __shared__ float ssum;
    constexpr int MAXIT = 8;                            // C <= 8192
    float cs[MAXIT][4];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[k][e] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (long bt = r0; bt < r1; ++bt) {
        const int b = (int)(bt / N), t = (int)(bt % N);
        bf16* out = dlogits + bt * C;
        const float nl = nll[b];
        const bool pad = t >= in_len[b], bad = !(nl < INFINITY);          // uniform over the workgroup
        if (pad || bad) {
            // padded frames: zero gradient; a sample with no alignment (nll = +inf) or a poisoned one (NaN): NaN rows, as ATen's
            const float f = pad ? 0.f : NAN;
#pragma unroll
            for (int k = 0; k < MAXIT; ++k) {
                const int c = k * 1024 + threadIdx.x * 4;
                if (c < C) {
                    float v[4] = {f, f, f, f};
                    store4(out + c, v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) cs[k][e] += f;
                }
            }
            continue;
        }
        // this row's logits are requested first: their latency runs under the occupancy phase (which waits on its own loads)
        const float* lr = logits + bt * C;
        float4 lv[MAXIT];
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            const int c = k * 1024 + threadIdx.x * 4;
            if (c < C) lv[k] = *reinterpret_cast<const float4*>(lr + c);
        }
        __syncthreads();                                // the previous row's reads of occ / ssum are done
        for (int c = threadIdx.x; c < C; c += 256) occ[c] = 0.f;
        if (threadIdx.x == 0) ssum = 0.f;
        __syncthreads();
        const int L = 2 * tg_len[b] + 1;
        const long base = bt * Lmax;
        // Half of the states are the blank: as LDS atomics they would all hit ONE address (a 32-way conflict per wave
        // instruction).  The stride (256) is even, so a thread sees either blanks only or labels only: blanks are summed in
        // registers, then per wave (no barrier), and added with one atomic per wave - as is S, the sum of all occupancies.
        float tot = 0.f, blk = 0.f;
        const float adj = (float)(offs[bt] + offs[(long)B * N + bt] + offs[2L * B * N + b]);      // see ctc_grad_kernel
        for (int s = threadIdx.x; s < L; s += 256) {
            const float o = __expf(alpha[base + s] + beta[base + s] - lpg[base + s] + adj);
            if (s & 1) atomicAdd(&occ[targets[(long)b * Smax + (s >> 1)]], o); else blk += o;
            tot += o;
        }
        tot = wave_sum(tot); blk = wave_sum(blk);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&ssum, tot); atomicAdd(&occ[blank], blk); }
        __syncthreads();                                // occ and ssum are complete
        const float S = ssum;
        const float g = grad_out ? grad_out[b] : 1.f, ls = lse[bt];
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            const int c = k * 1024 + threadIdx.x * 4;
            if (c < C) {
                float v[4] = {lv[k].x, lv[k].y, lv[k].z, lv[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = g * (__expf(v[e] - ls) * S - occ[c + e]); cs[k][e] += (float)(bf16)v[e]; }
                store4(out + c, v);
            }
        }
    }
    if (slab) {
#pragma unroll
        for (int k = 0; k < MAXIT; ++k) {
            const int c = k * 1024 + threadIdx.x * 4;
            if (c < C) store4(slab + (long)blockIdx.x * C + c, cs[k]);
        }
    }
}

}  // namespace

SCONF_API int sconf_colsum(const void* x, int x_dtype, float* out, int64_t M, int64_t N, int64_t ld, float alpha, hipStream_t stream);

// Workspace contract: lpg, alpha, beta are f32 [B][N][Lmax] with Lmax = 2*Smax+1 (caller-allocated); offs is f64 [2*B*N + B]:
// the per-frame offsets of the renormalised alpha rows, of the beta rows, and the nll in f64 (read back by the backward).
// targets int32 [B][Smax]; input_lengths / target_lengths int32 [B].  nll f32 [B] (loss = sum).
static int ctc_fwd_impl(const char* who, bool from_logits, const float* in, float* lse, const int32_t* targets, const int32_t* input_lengths,
                        const int32_t* target_lengths, float* lpg, float* alpha, float* beta, double* offs, float* nll,
                        int64_t B, int64_t N, int64_t C, int64_t Smax, int blank, hipStream_t stream) {
    if (B == 0) return 0;
    SCONF_REQUIRE(offs != nullptr, "%s: the f64 offset workspace (2*B*N + B doubles) is required", who);
    const int Lmax = (int)(2 * Smax + 1);
    SCONF_REQUIRE(blank >= 0 && blank < C, "%s: blank %d out of range", who, blank);
    SCONF_REQUIRE((long)(Lmax + 18) * 8 + 144 <= 160 * 1024, "%s: lattice of %d states does not fit LDS", who, Lmax);   // (f32 state)
    SCONF_REQUIRE(C % 4 == 0 && C * 4 <= 64 * 1024, "%s: C must be a multiple of 4 and one row must fit LDS (%ld classes)", who, (long)C);
    const dim3 gg((unsigned)std::min<long>(B * N, 65536));
    if (from_logits) hipLaunchKernelGGL(ctc_gather_kernel<true>, gg, dim3(256), (size_t)C * 4, stream, in, targets, input_lengths, target_lengths,
                                        lpg, lse, (int)B, (int)N, (int)C, (int)Smax, Lmax, blank);
    else hipLaunchKernelGGL(ctc_gather_kernel<false>, gg, dim3(256), (size_t)C * 4, stream, in, targets, input_lengths, target_lengths,
                            lpg, (float*)nullptr, (int)B, (int)N, (int)C, (int)Smax, Lmax, blank);
    int nt = Lmax <= 256 ? 256 : (Lmax <= 512 ? 512 : 1024);     // the serial step costs a barrier + the slowest thread: few states each
    if (const char* e = getenv("SCONF_CTC_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) nt = v; }   // tuning
    const int spt = cdiv(Lmax, nt);
    const int spt_pad = spt <= 4 ? spt : spt <= 6 ? 6 : spt <= 8 ? 8 : spt <= 10 ? 10 : spt <= 12 ? 12 : 16;   // MAXS of the instantiation taken below
    const size_t stage = spt_pad > 4 ? (size_t)(nt / 64) * 64 * (spt_pad / 2 + 1) * 4 + 64 : 0;      // per-wave emission staging segments of the long-lattice form
    const bool f64_state = ((size_t)2 * (Lmax + spt_pad + 2)) * 8 + 144 + stage <= 160 * 1024;        // else f32 state (lattices of more than ~9000 states)
    const size_t sh = ((size_t)2 * (Lmax + spt_pad + 2)) * (f64_state ? 8 : 4) + 144 + stage;
    SCONF_REQUIRE(spt <= 16, "%s: target too long (%ld labels)", who, (long)Smax);
#define L2(MS, LT) do { \
        if (sh > 48 * 1024) (void)hipFuncSetAttribute((const void*)ctc_alphabeta_kernel<MS, LT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL((ctc_alphabeta_kernel<MS, LT>), dim3((unsigned)(2 * B)), dim3(nt), sh, stream, lpg, targets, input_lengths, \
                           target_lengths, alpha, beta, nll, offs, (int)B, (int)N, (int)C, (int)Smax, Lmax, blank); } while (0)
#define L(MS) do { if (f64_state) L2(MS, double); else L2(MS, float); } while (0)
    switch (spt_pad) { case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; case 6: L(6); break; case 8: L(8); break;
                       case 10: L(10); break; case 12: L(12); break; default: L(16); }
#undef L2
#undef L
#ifdef CTC_STAMP
    if (getenv("SCONF_CTC_STAMP_PRINT")) {
        (void)hipStreamSynchronize(stream);
        unsigned long long h[16 * 8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ctc_stamps), sizeof(h));
        for (int w = 0; w < nt / 64; ++w) {
            fprintf(stderr, "[ctc stamps] wave %2d:", w);
            for (int i = 0; i < 6; ++i) fprintf(stderr, " %8.1f", (double)h[w * 8 + i] / (double)N);
            fprintf(stderr, "   (memtime ticks per frame: lds-read, math, write+store, barrier, prefetch-issue, prefetch-wait)\n");
        }
    }
#endif
    return 0;
}

SCONF_API int sconf_ctc_fwd(const float* log_probs, const int32_t* targets, const int32_t* input_lengths,
                            const int32_t* target_lengths, float* lpg, float* alpha, float* beta, double* offs, float* nll,
                            int64_t B, int64_t N, int64_t C, int64_t Smax, int blank, hipStream_t stream) {
    if (ctc_fwd_impl("sconf_ctc_fwd", false, log_probs, nullptr, targets, input_lengths, target_lengths, lpg, alpha, beta, offs, nll, B, N, C, Smax, blank, stream)) return 1;
    SCONF_LAUNCH_OK("sconf_ctc_fwd");
    return 0;
}

// The same loss from the decoder's LOGITS (f32 (B,N,C)): log_softmax is folded into the emission gather (row log-sum-exp -> lse
// (B,N) f32, kept for the backward), so neither the log-probabilities nor their gradient exist as (B,N,C) tensors:
// decoder.py:25 F.log_softmax + exp/train.py:104,249 CTCLoss as one operator.
SCONF_API int sconf_ctc_fwd_logits(const float* logits, const int32_t* targets, const int32_t* input_lengths,
                                   const int32_t* target_lengths, float* lse, float* lpg, float* alpha, float* beta, double* offs, float* nll,
                                   int64_t B, int64_t N, int64_t C, int64_t Smax, int blank, hipStream_t stream) {
    if (ctc_fwd_impl("sconf_ctc_fwd_logits", true, logits, lse, targets, input_lengths, target_lengths, lpg, alpha, beta, offs, nll, B, N, C, Smax, blank, stream)) return 1;
    SCONF_LAUNCH_OK("sconf_ctc_fwd_logits");
    return 0;
}

SCONF_API int sconf_num_cus(void);
static int ctc_bwd_slabs() {                            // workgroups of the fused gradient kernel (each keeps a slab of column sums)
    static int v = 0;
    if (!v) { const char* e = getenv("SCONF_CTC_BWD_SLABS"); v = e ? atoi(e) : 0; if (v < 64 || v > 65536) { const int n = sconf_num_cus(); v = 12 * (n > 0 ? n : 256); } }   // 6 resident per CU: whole rounds (2048: 2.78 ms, 3072: 2.63 at B = 128)
    return v;
}
#define CTC_BWD_SLABS ctc_bwd_slabs()
SCONF_API int64_t sconf_ctc_bwd_logits_workspace(int64_t rows, int64_t C) { return (int64_t)std::min<long>(rows, CTC_BWD_SLABS) * C; }
// dlogits (B,N,C) bf16 = d nll / d logits (CTC gradient through log_softmax), scaled by grad_out[b] (null = 1).
// colsum_out (optional, f32 [C], ACCUMULATED): column sums of dlogits - the decoder bias gradient; needs the workspace
// (sconf_ctc_bwd_logits_workspace(B * N, C) floats).
SCONF_API int sconf_ctc_bwd_logits(const float* logits, const float* lse, const float* lpg, const float* alpha, const float* beta,
                                   const double* offs, const float* nll, const int32_t* targets, const int32_t* input_lengths, const int32_t* target_lengths,
                                   const float* grad_out, void* dlogits_bf16, float* colsum_out, float* workspace,
                                   int64_t B, int64_t N, int64_t C, int64_t Smax, int blank, hipStream_t stream) {
    if (B * N == 0) return 0;
    SCONF_REQUIRE(C % 4 == 0 && C <= 8192, "sconf_ctc_bwd_logits: C must be a multiple of 4 and <= 8192");
    SCONF_REQUIRE(!colsum_out || workspace, "sconf_ctc_bwd_logits: colsum_out needs the workspace");
    const int Lmax = (int)(2 * Smax + 1);
    const long rows = B * N;
    const int rpb = colsum_out ? (int)cdiv(rows, CTC_BWD_SLABS) : 1;
    const unsigned grid = (unsigned)cdiv(rows, rpb);
    hipLaunchKernelGGL(ctc_grad_logits_kernel, dim3(grid), dim3(256), (size_t)C * 4, stream, logits, lse, lpg, alpha, beta, nll, offs, targets,
                       input_lengths, target_lengths, grad_out, (bf16*)dlogits_bf16, colsum_out ? workspace : nullptr, rows, (int)B, (int)N, (int)C,
                       (int)Smax, Lmax, blank, rpb);
    SCONF_LAUNCH_OK("sconf_ctc_bwd_logits");
    if (colsum_out) return sconf_colsum(workspace, SCONF_F32, colsum_out, (int64_t)grid, C, C, 1.f, stream);     // += over the slabs
    return 0;
}

// grad (B,N,C) f32 = grad_out[b] * (exp(lp) - occupancy), zero for t >= input_length (ATen ctc_loss backward).
// grad_out: f32 [B] (per-sample upstream gradient of the nll vector) or null (= 1).
SCONF_API int sconf_ctc_bwd(const float* log_probs, const float* lpg, const float* alpha, const float* beta, const double* offs, const float* nll,
                            const int32_t* targets, const int32_t* input_lengths, const int32_t* target_lengths,
                            const float* grad_out, float* grad, int64_t B, int64_t N, int64_t C, int64_t Smax, int blank,
                            hipStream_t stream) {
    if (B * N == 0) return 0;
    SCONF_REQUIRE(C % 4 == 0, "sconf_ctc_bwd: C must be a multiple of 4");
    SCONF_REQUIRE(C * 4 <= 64 * 1024, "sconf_ctc_bwd: %ld classes do not fit LDS", (long)C);
    const int Lmax = (int)(2 * Smax + 1);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)(B * N)), dim3(256), (size_t)C * 4, stream, log_probs, lpg, alpha, beta, nll, offs,
                       targets, input_lengths, target_lengths, grad_out, grad, (int)B, (int)N, (int)C, (int)Smax, Lmax, blank);
    SCONF_LAUNCH_OK("sconf_ctc_bwd");
    return 0;
}
