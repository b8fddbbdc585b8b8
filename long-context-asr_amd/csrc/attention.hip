// Flash-style bidirectional self-attention for gfx950, forward + backward, head_dim 32 or 128.
// Replaces flash_attn_qkvpacked_func / flash_attn_varlen_qkvpacked_func + bert_padding un/pad
// (attention.py:235-257, 527-535) and the CPU SDPA branch (attention.py:536-544):
// non-causal softmax(QK^T/sqrt(D))V, optional key-padding via per-sample lengths (ragged batches
// stay padded: no unpad/pad copies), optional sliding window (left,right) with flash-attn semantics.
//
// MI355X design
//  * v_mfma_f32_32x32x16_bf16 everywhere; scores are computed TRANSPOSED (S^T = K Q^T) so the query
//    index sits on the lane: running max / sum / LSE / delta are lane-local scalars and the f32
//    score accumulators are, after a bf16 pack, directly the B operand of the next MFMA
//    (O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS) — P never touches LDS.
//  * The other operand of those products is read from the row-major LDS tile with the hardware
//    transpose read ds_read_b64_tr_b16; ONE swizzled LDS image per tile serves both the row reads
//    (ds_read_b128) and the transposed reads conflict-free.
//  * K/V (or Q/dO) tiles are staged global -> registers -> LDS, issue-early / write-late, double
//    buffered, one barrier per tile.
//  * Backward = two kernels (dK/dV with keys resident per wave; dQ with queries resident per wave):
//    no atomics, bitwise reproducible, at the price of recomputing S and dP once more.
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include <stdio.h>

namespace {

struct AttnParams {
    const bf16 *q, *k, *v;            // (B,N,H,D) views: element (b,n,h,d) at b*sb + n*sn + h*sh + d
    bf16* o;
    const bf16 *dout;                 // backward
    bf16 *dq, *dk, *dv;
    float* lse;                       // (B,H,N) natural-log LSE of the scaled scores
    float* delta;                     // (B,H,N) rowsum(dO * O)
    long q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
    long do_sb, do_sn, do_sh, dq_sb, dq_sn, dq_sh, dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh;
    const int* lengths;               // per-sample valid length (queries and keys), or null
    int B, N, H;
    int win_left, win_right;          // -1 = unbounded
    float scale;                      // 1/sqrt(D)
    int xcd_remap;                    // 1: XCD-contiguous workgroup order (decode_block)
    const float *rot_cos, *rot_sin;   // backward: (N, D/2) rotary tables or null - dq, dk are returned as gradients of the UNROTATED q, k
    unsigned long long* stamps;       // diagnostic builds only (-DSCONF_ATTN_STAMP): per (workgroup, wave) segment cycle sums
};

// In-kernel time stamps (cdna_hip_programming.md section 7): a DIAGNOSTIC build only (make EXTRA=-DSCONF_ATTN_STAMP); in the product
// build the macros are empty and no stamp executes.  Segment sums are kept per wave and written once after the loop, to a buffer
// nothing else reads.
#ifdef SCONF_ATTN_STAMP
#define STAMP_DECL(n) unsigned long long st_acc_[n] = {}, st_last_ = 0; { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_last_ = t_; }
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      __builtin_amdgcn_sched_barrier(0); st_acc_[i] += t_ - st_last_; st_last_ = t_; } while (0)
#define STAMP_OUT(n) do { if (p.stamps && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < (n); ++i_) p.stamps[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + i_] = st_acc_[i_]; } while (0)
#else
#define STAMP_DECL(n)
#define STAMP(i)
#define STAMP_OUT(n)
#endif

// Workgroup -> (batch, head, row block) for a 1-D launch of nx * H * B workgroups.  The hardware deals consecutive workgroup ids
// to the 8 XCDs round-robin, so with the natural order the 8 row blocks of one (b, h) land on 8 different XCDs and every XCD's
// 4 MiB L2 sees the K/V (or Q/dO) tiles of ALL ~32 concurrently running (b, h) pairs: measured 3.6-4.2 GB of fabric reads per
// launch against 0.4 GB of operands (9x, once per XCD).  The remap gives each XCD a CONTIGUOUS range of logical ids (bijective
// for any total) and orders logical ids row-block-fastest, so the ~32 workgroups resident on an XCD cover a few whole (b, h).
struct BlkId { int b, h, x; };
__device__ __forceinline__ BlkId decode_block(const AttnParams& p, int nx) {
    const int total = nx * p.H * p.B, v = blockIdx.x;
    int lid = v;
    if (p.xcd_remap) {
        const int q = total >> 3, r = total & 7, xcd = v & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    BlkId o;
    o.x = lid % nx;
    const int bh = lid / nx;
    o.h = bh % p.H; o.b = bh / p.H;
    return o;
}

constexpr float RESCALE_LOG2 = 6.f;   // forward: rescale O / l only when a row's maximum has grown by more than 2^6 (see attn_fwd*_kernel)

template <int D> __device__ __forceinline__ int tile_off(int row, int ch) {
    const int swz = (D == 128) ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
    return row * (2 * D) + ((ch ^ swz) << 4);
}

// ---- staging of a [ROWS][D] bf16 tile ---------------------------------------------------------
template <int D, int ROWS> struct Stage {
    static constexpr int CPR = D / 8, CHUNKS = ROWS * CPR, PER = (CHUNKS + 255) / 256;
    uint4 r[PER];
    __device__ __forceinline__ void gload(const bf16* base, long sn, int row0, int nrows_valid, int tid) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + 256 * i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < CHUNKS) {
                const int row = c / CPR, ch = c % CPR;
                if (row0 + row < nrows_valid) v = *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * sn + ch * 8);
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void lstore(char* s, int tid) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + 256 * i;
            if (c < CHUNKS) *reinterpret_cast<uint4*>(s + tile_off<D>(c / CPR, c % CPR)) = r[i];
        }
    }
};

// ---- LDS-DMA staging of a [ROWS][D] bf16 tile (global_load_lds_dwordx4): no staging VGPRs, no ds_write, no guards --
// LDS destination is linear (wave base + lane*16), so the tile_off swizzle is applied to the per-lane SOURCE chunk and
// undone by the same XOR on the fragment reads.  Rows beyond the tensor are clamped to its last row: their scores are
// masked (keys) or their LSE is +inf (queries), so the duplicated finite data never reaches an output.
template <int D, int ROWS>
__device__ __forceinline__ void glds_tile(const bf16* base, long sn, int row0, int nrows_valid, char* lds, int tid) {
    constexpr int CPR = D / 8, CHUNKS = ROWS * CPR, PER = (CHUNKS + 255) / 256;
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + 256 * i;
        if (CHUNKS % 256 == 0 || c < CHUNKS) {
            const int row = c / CPR, pos = c % CPR;
            const int swz = (D == 128) ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
            const int gr = min(row0 + row, nrows_valid - 1);
            __builtin_amdgcn_global_load_lds((gptr)(base + (long)gr * sn + (pos ^ swz) * 8), (lptr)(lds + ((tid & ~63) + 256 * i) * 16), 16, 0, 0);
        }
    }
}

// the same for workgroups of NTHR threads (tile = whole passes of the workgroup)
template <int D, int ROWS, int NTHR>
__device__ __forceinline__ void glds_tile_n(const bf16* base, long sn, int row0, int nrows_valid, char* lds, int tid) {
    constexpr int CPR = D / 8, CHUNKS = ROWS * CPR, PER = CHUNKS / NTHR;
    static_assert(CHUNKS % NTHR == 0, "tile must be whole passes of the workgroup");
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + NTHR * i;
        const int row = c / CPR, pos = c % CPR;
        const int swz = (D == 128) ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
        const int gr = min(row0 + row, nrows_valid - 1);
        __builtin_amdgcn_global_load_lds((gptr)(base + (long)gr * sn + (pos ^ swz) * 8), (lptr)(lds + ((tid & ~63) + NTHR * i) * 16), 16, 0, 0);
    }
}
template <int D, int ROWS>
__device__ __forceinline__ void glds_tile512(const bf16* base, long sn, int row0, int nrows_valid, char* lds, int tid) {
    glds_tile_n<D, ROWS, 512>(base, sn, row0, nrows_valid, lds, tid);
}

// ---- LDS-DMA issued from inline asm ----------------------------------------------------------------------------------
// hipcc cannot tell an LDS read from the destination of an in-flight global_load_lds issued through the builtin, so it puts
// `s_waitcnt vmcnt(0)` in front of the first LDS read that follows one in program order: issued at the top of a stage, the next
// stage's tile had to LAND before the current stage could be computed (60 % of the wave-cycles of the 8-wave dK/dV kernel were
// parked there).  An asm statement is opaque to that bookkeeping: the 8-wave kernels below issue their DMA here, wait for it
// themselves (`dma_wait_all` right before the barrier that publishes the stage) and leave every LDS READ to the compiler.
// For that to work the loop must hold NO vector-memory operation the compiler knows of (the hardware counter is in order and
// shared: any `s_waitcnt vmcnt(N)` it emits for a load of its own - a fragment loaded before the loop whose wait it sinks to the
// first use, a spill reload, a per-stage statistics load - also waits for the DMA issued before it).
// M0 carries the wave's LDS destination base and is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma16_asm(const void* gsrc, unsigned lds_dst_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_wave_base) : "memory");
}
__device__ __forceinline__ void dma4_asm(const void* gsrc, unsigned lds_dst_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_wave_base) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// [ROWS][D] bf16 tile by asm DMA, NTHR threads; same swizzled image as glds_tile_n
// `lds` = byte address of the tile in LDS as a wave-uniform (scalar) value
template <int D, int ROWS, int NTHR>
__device__ __forceinline__ void dma_tile(const bf16* base, long sn, int row0, int nrows_valid, unsigned lds, int tid) {
    constexpr int CPR = D / 8, CHUNKS = ROWS * CPR, PER = CHUNKS / NTHR;
    static_assert(CHUNKS % NTHR == 0, "tile must be whole passes of the workgroup");
    const unsigned wbase = lds + (unsigned)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + NTHR * i;
        const int row = c / CPR, pos = c % CPR;
        const int swz = (D == 128) ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
        const int gr = min(row0 + row, nrows_valid - 1);
        dma16_asm(base + (long)gr * sn + (pos ^ swz) * 8, wbase + (unsigned)(NTHR * i) * 16u);
    }
}

// ---- LDS-DMA through a buffer descriptor, a whole tile per asm statement (round 3) -----------------------------------------------
// In-kernel stamps of the 8-wave forward (DESIGN, round 3) showed the DMA ISSUE of dma_tile above on every wave's critical
// path: 950 cycles per stage for waves 0-3 and 1950 for waves 4-7 (12-25 % of the kernel) - per 16-byte piece a 64-bit address
// (row clamp, multiply by the row stride, swizzle) rebuilt on the VALU plus an M0 save / set / restore around it.  Here the
// per-lane part of the address is ONE loop-invariant 32-bit offset (a piece is NTHR / 16 whole rows further down: the swizzle
// depends on row & 15 only), the rest is scalar: `buffer_load_dwordx4 voff, srd, soff offen lds` with soff and M0 stepped by
// s_add.  Rows past the tensor need no clamp: the descriptor's range check returns zeros for them (their keys are masked and
// their query rows are never stored).  3 scalar instructions + the load per piece, M0 saved and restored once per tile.
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base, long nbytes) {
    const unsigned long a = (unsigned long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)(nbytes > 0xffffffffL ? 0xffffffffL : nbytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long)hi << 32) | lo), 0, nb, 0x00020000);
}
// bytes of a (rows, D) bf16 view with row stride sn (elements) that start at its first row
template <int D> __device__ __forceinline__ long view_bytes(int rows, long sn) { return rows > 0 ? ((long)(rows - 1) * sn + D) * 2 : 0; }
// this lane's source byte offset inside one pass (NTHR / (D / 8) rows) of a [rows][D] tile: the tile_off swizzle on the source chunk
template <int D, int NTHR> __device__ __forceinline__ unsigned tile_voff(long sn, int tid) {
    constexpr int CPR = D / 8;
    static_assert((NTHR / CPR) % 16 == 0, "a pass must be whole groups of 16 rows (the swizzle period)");
    const int row = tid / CPR, pos = tid % CPR;
    const int swz = (D == 128) ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
    return (unsigned)(row * sn * 2 + ((pos ^ swz) << 4));
}
// NP pieces of a tile: piece k reads srd base + soff + k * sstep + voff and lands at lds_wave_base + k * LSTEP + lane * 16
#define SCONF_DMA_FIRST "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %1 offen lds\n\t"
#define SCONF_DMA_NEXT  "s_add_u32 m0, m0, %6\n\ts_add_u32 %1, %1, %4\n\tbuffer_load_dwordx4 %2, %3, %1 offen lds\n\t"
#define SCONF_DMA_LAST  "s_mov_b32 m0, %0"
template <int NP, int LSTEP> __device__ __forceinline__ void dma_pieces(srd_t srd, unsigned voff, unsigned soff, unsigned sstep, unsigned lds_wave_base) {
    static_assert(NP == 2 || NP == 4 || NP == 8, "tiles of 2, 4 or 8 pieces per wave");
    unsigned keep;
    if constexpr (NP == 8)
        asm volatile(SCONF_DMA_FIRST SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_LAST
                     : "=&s"(keep), "+s"(soff) : "v"(voff), "s"(srd), "s"(sstep), "s"(lds_wave_base), "n"(LSTEP) : "memory");
    else if constexpr (NP == 4)
        asm volatile(SCONF_DMA_FIRST SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_NEXT SCONF_DMA_LAST
                     : "=&s"(keep), "+s"(soff) : "v"(voff), "s"(srd), "s"(sstep), "s"(lds_wave_base), "n"(LSTEP) : "memory");
    else
        asm volatile(SCONF_DMA_FIRST SCONF_DMA_NEXT SCONF_DMA_LAST
                     : "=&s"(keep), "+s"(soff) : "v"(voff), "s"(srd), "s"(sstep), "s"(lds_wave_base), "n"(LSTEP) : "memory");
}
// A [ROWS][128] tile staged by a 512-thread workgroup = ROWS / 32 passes of 8 wave pieces.  Who issues them: every wave its own
// piece of each pass, or - LOADERS - waves 0-3 only, each also the piece of wave w + 4 (16 rows further down: same swizzle, the
// LDS block 4 KiB further on).  Stamps of the 8-wave forward: waves 4-7 lose every arbitration on their SIMD (priority, then age:
// MI355X_MICROARCH.md 'Two waves per SIMD') and are the critical path, while waves 0-3 sit in the stage barrier for a third of
// the kernel; an LDS-DMA piece costs its issuing wave 75-180 cycles, so the pieces go to the waves that have the slack.  Which
// waves are the older ones is an observation, not a contract: a different placement changes speed only.
template <int ROWS, bool LOADERS> __device__ __forceinline__ void dma_tile128(srd_t srd, unsigned voff, long sn, int row0, unsigned lds_tile, int wave_u) {
    constexpr int PASSES = ROWS / 32;
    const unsigned rb = (unsigned)(sn * 2);                                       // bytes per row
    if constexpr (!LOADERS) dma_pieces<PASSES, 8192>(srd, voff, (unsigned)row0 * rb, 32u * rb, lds_tile + (unsigned)wave_u * 1024u);
    else if (wave_u < 4) dma_pieces<2 * PASSES, 4096>(srd, voff, (unsigned)row0 * rb, 16u * rb, lds_tile + (unsigned)wave_u * 1024u);
}

// make a fragment array's loads complete, as far as the compiler can tell, HERE (an empty asm that "rewrites" each register)
template <int NF> __device__ __forceinline__ void pin_frags(bf16x8 (&f)[NF]) {
#pragma unroll
    for (int i = 0; i < NF; ++i) asm volatile("" : "+v"(f[i]));
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// Loop-invariant per-lane LDS byte offsets.  tile_off's swizzle depends only on (row & 3) and ((row >> 2) & 3), so a
// row base that is a multiple of 16 adds linearly (rbase * 2D) and every fragment read is "lane offset + constant":
// the address math leaves the inner loops (it was ~8 VALU ops per ds_read, several hundred per MFMA block).
template <int D> struct LaneOffs {
    int row[D / 16];                 // row form, k-step st
    int trlo[D / 32];                // transposed form, column block db, rows rb+4hh+q4; rows +8: (trlo ^ 32) + 16 D (frag_tr)
    __device__ __forceinline__ LaneOffs(int lane) {
#pragma unroll
        for (int st = 0; st < D / 16; ++st) row[st] = tile_off<D>(lane & 31, 2 * st + (lane >> 5));
        const int i = lane & 15, q4 = i >> 2, p4 = i & 3, G = (lane >> 4) & 1, hh = lane >> 5;
        const int row0 = 4 * hh + q4, sub = (p4 & 1) * 8;
#pragma unroll
        for (int db = 0; db < D / 32; ++db) trlo[db] = tile_off<D>(row0, db * 4 + 2 * G + (p4 >> 1)) + sub;
    }
};
// A operand, row form: A[row = rbase + lane&31][k = 16*st + 8*hh + j]          (rbase % 16 == 0)
template <int D> __device__ __forceinline__ bf16x8 frag_row(const char* s, const LaneOffs<D>& L, int rbase, int st) {
    return *reinterpret_cast<const bf16x8*>(s + rbase * 2 * D + L.row[st]);
}
// A operand, transposed form: A[row = tile column db*32 + lane&31][k-slot j] where slot j of lane half hh is
// tile row  rb + 8*(j>>2) + 4*hh + (j&3)  — the k order of a packed 32x32 accumulator (B operand).   (rb % 16 == 0)
// The second half (tile rows +8) needs no offset table of its own: 8 rows further the swizzle's low chunk bits are flipped by 2
// (tile_off: ((row >> 2) & 3) ^ 2), i.e. the byte offset is (trlo ^ 32) + 8 rows.  `x32` is that 32: callers short of registers
// pass an opaque copy made inside their loop, so that hipcc cannot hoist the D/32 derived offsets back into live registers.
template <int D> __device__ __forceinline__ bf16x8 frag_tr(const char* s, const LaneOffs<D>& L, int rb, int db, int x32 = 32) {
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(s + rb * 2 * D + L.trlo[db]));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(s + rb * 2 * D + 16 * D + (L.trlo[db] ^ x32)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// B operand straight from global: B[k = 16*st + 8*hh + j][col = lane&31] = X[row0 + lane&31][d]
template <int D> __device__ __forceinline__ void load_bfrags(bf16x8 (&f)[D / 16], const bf16* base, long sn, int row0, int nvalid, int lane) {
    const int r = row0 + (lane & 31);
#pragma unroll
    for (int st = 0; st < D / 16; ++st) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < nvalid) v = *reinterpret_cast<const uint4*>(base + (long)r * sn + 16 * st + 8 * (lane >> 5));
        f[st] = __builtin_bit_cast(bf16x8, v);
    }
}
// delta[q] = sum_d dO[q][d] O[q][d] for this lane's query (lane & 31), from the dO fragments the dQ kernels hold anyway and O
// fragments loaded the same way (each half-wave holds half of the d range: one cross-half exchange).  The dQ kernels run FIRST in
// the backward, keep delta in a register for themselves and write it out for the dK/dV kernel: the separate delta pass
// (round 1: attn_delta_kernel, 0.14 ms per layer at B = 128) is gone.
template <int D> __device__ __forceinline__ float delta_from_frags(const bf16x8 (&gf)[D / 16], const bf16* op, long o_sn, int row0, int nvalid, int lane) {
    bf16x8 of[D / 16];
    load_bfrags<D>(of, op, o_sn, row0, nvalid, lane);
    float acc = 0.f;
#pragma unroll
    for (int st = 0; st < D / 16; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (float)gf[st][j] * (float)of[st][j];
    return acc + __shfl_xor(acc, 32, 64);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s2) {
    bf16x8 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = (bf16)a[8 * s2 + j];
    return p;
}
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// store a transposed accumulator set X^T[d][row] (d over D/32 blocks) as X[row][d] bf16, scaled
template <int D> __device__ __forceinline__ void store_t(const f32x16 (&acc)[D / 32], bf16* dst_row, float sc, int hh) {
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[db][4 * r4 + e] * sc;
            store4(dst_row + db * 32 + 8 * r4 + 4 * hh, v);
        }
}

// The same store with the transpose of the NeoX rotation applied first: acc is the gradient with respect to the ROTATED q (or k) of
// row n; what is stored is the gradient with respect to the unrotated one (rotary_emb.py:61-73):
//     g[d] = g'[d] cos + g'[d + D/2] sin,   g[d + D/2] = g'[d + D/2] cos - g'[d] sin        (d < D/2, cos/sin at (n, d)).
// Both partners sit in the same lane: d = 32 db + acc_row(r, hh), so d + 64 is accumulator block db + 2 (D = 128) and d + 16 is
// register r + 8 (D = 32).  This replaces the separate (dq, dk, dv) -> dqkv rotary pass over the activations.
template <int D> __device__ __forceinline__ void store_t_rot(f32x16 (&acc)[D / 32], bf16* dst_row, float sc, int hh, const float* cs, const float* sn) {
    if constexpr (D == 128) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float c[4], s[4];
                load4(cs + db * 32 + 8 * r4 + 4 * hh, c); load4(sn + db * 32 + 8 * r4 + 4 * hh, s);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = acc[db][4 * r4 + e], b = acc[db + 2][4 * r4 + e];
                    acc[db][4 * r4 + e] = a * c[e] + b * s[e];
                    acc[db + 2][4 * r4 + e] = b * c[e] - a * s[e];
                }
            }
    } else {
#pragma unroll
        for (int r4 = 0; r4 < 2; ++r4) {
            float c[4], s[4];
            load4(cs + 8 * r4 + 4 * hh, c); load4(sn + 8 * r4 + 4 * hh, s);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = acc[0][4 * r4 + e], b = acc[0][4 * r4 + 8 + e];
                acc[0][4 * r4 + e] = a * c[e] + b * s[e];
                acc[0][4 * r4 + 8 + e] = b * c[e] - a * s[e];
            }
        }
    }
    store_t<D>(acc, dst_row, sc, hh);
}

// =============================================================================================
// forward: grid (ceil(N/128), H, B), 4 waves x 32 query rows, 64-key tiles
// =============================================================================================
template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * 2 * D;                     // bytes of one 64-row tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const BlkId bid = decode_block(p, (p.N + 127) / 128);
    const int b = bid.b, h = bid.h, qb0 = bid.x * 128, q0 = qb0 + wave * 32;
    const int len = p.lengths ? p.lengths[b] : p.N;
    const bf16* qp = p.q + b * p.q_sb + h * p.q_sh;
    const bf16* kp = p.k + b * p.k_sb + h * p.k_sh;
    const bf16* vp = p.v + b * p.v_sb + h * p.v_sh;
    const int qi = q0 + (lane & 31);
    const float c = p.scale * 1.4426950408889634f;

    bf16x8 qf[D / 16];
    load_bfrags<D>(qf, qp, p.q_sn, q0, p.N, lane);
    const LaneOffs<D> L(lane);

    const int kv_lo = p.win_left < 0 ? 0 : max(0, qb0 - p.win_left);
    const int kv_hi = min(len, p.win_right < 0 ? len : qb0 + 128 + p.win_right);
    const int t_lo = kv_lo / 64, t_hi = (kv_hi + 63) / 64;
    const bool windowed = p.win_left >= 0 || p.win_right >= 0;

    f32x16 o[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    if (t_lo < t_hi) { glds_tile<D, 64>(kp, p.k_sn, t_lo * 64, p.N, smem, tid); glds_tile<D, 64>(vp, p.v_sn, t_lo * 64, p.N, smem + TB, tid); }
    __syncthreads();
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1;
        const char* sK = smem + cur * 2 * TB;
        const char* sV = sK + TB;
        if (t + 1 < t_hi) {                                  // next K/V tile streams into the other buffer during this tile
            char* dK = smem + (cur ^ 1) * 2 * TB;
            glds_tile<D, 64>(kp, p.k_sn, (t + 1) * 64, p.N, dK, tid); glds_tile<D, 64>(vp, p.v_sn, (t + 1) * 64, p.N, dK + TB, tid);
        }
        const int kv0 = t * 64;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int st = 0; st < D / 16; ++st)
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sK, L, kt * 32, st), qf[st], s[kt], 0, 0, 0);
        }
        if (kv0 + 64 > kv_hi || windowed) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + acc_row(r, hh);
                    bool ok = key < len;
                    if (p.win_left >= 0) ok = ok && key >= qi - p.win_left;
                    if (p.win_right >= 0) ok = ok && key <= qi + p.win_right;
                    if (!ok) s[kt][r] = -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Deferred rescale: m is the reference point of the exponentials, not necessarily the running maximum.  It moves (and
        // O, l are rescaled: 64 + 1 multiplies per lane) only when some row of the wave has outgrown it by more than 2^RESCALE_LOG2;
        // until then P <= 2^RESCALE_LOG2 instead of <= 1, which costs nothing (bf16 P keeps its relative precision, O and l stay
        // far inside f32 range) and O / l at the end is the same quotient.  The softmax VALU work is as long as the tile's
        // MFMAs here, so the skipped rescale is time, not just instructions.
        if (__any((mx - m) * c > RESCALE_LOG2)) {            // wave-uniform; also the first tile of every row (m = -inf)
            const float mn = fmaxf(m, mx);
            const float alpha = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m - mn) * c);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        const float mc = (m == -INFINITY) ? 0.f : m * c;
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] * c - mc); s[kt][r] = e; rs += e; }
        l += rs;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { pf[kt][0] = pack8(s[kt], 0); pf[kt][1] = pack8(s[kt], 1); }
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sV, L, kt * 32 + 16 * s2, db), pf[kt][s2], o[db], 0, 0, 0);
        __syncthreads();
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    if (qi < p.N) {
        const bool live = qi < len && lt > 0.f;
        const float inv = live ? 1.f / lt : 0.f;
        store_t<D>(o, p.o + b * p.o_sb + (long)qi * p.o_sn + h * p.o_sh, inv, hh);
        if (hh == 0 && p.lse) p.lse[((long)b * p.H + h) * p.N + qi] = live ? (m * c + __log2f(lt)) * 0.6931471805599453f : INFINITY;
    }
}

// 8-wave form of the forward (D = 128): 256 queries per workgroup, 128-key stages shared by 8 waves.
template <int D>
__global__ __launch_bounds__(512) void attn_fwd8_kernel(const AttnParams p) {
    constexpr int KT = 128;                            // keys per stage, consumed as two 64-key halves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = KT * 2 * D;                     // bytes of one stage tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const BlkId bid = decode_block(p, (p.N + 255) / 256);
    const int b = bid.b, h = bid.h, qb0 = bid.x * 256, q0 = qb0 + wave * 32;
    const int len = p.lengths ? p.lengths[b] : p.N;
    const bf16* qp = p.q + b * p.q_sb + h * p.q_sh;
    const bf16* kp = p.k + b * p.k_sb + h * p.k_sh;
    const bf16* vp = p.v + b * p.v_sb + h * p.v_sh;
    const int qi = q0 + (lane & 31);
    const float c = p.scale * 1.4426950408889634f;

    bf16x8 qf[D / 16];
    load_bfrags<D>(qf, qp, p.q_sn, q0, p.N, lane);
    const LaneOffs<D> L(lane);

    const int kv_lo = p.win_left < 0 ? 0 : max(0, qb0 - p.win_left);
    const int kv_hi = min(len, p.win_right < 0 ? len : qb0 + 256 + p.win_right);
    const int t_lo = kv_lo / KT, t_hi = (kv_hi + KT - 1) / KT;
    const bool windowed = p.win_left >= 0 || p.win_right >= 0;

    f32x16 o[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto issue = [&](int t, int buf) {                     // K | V stage by asm-issued LDS-DMA (invisible to hipcc's vmcnt bookkeeping)
        const int tid_ = opaque(tid);                      // offsets recomputed per stage: kept live across the loop they would be spilled
        dma_tile<D, KT, 512>(kp, p.k_sn, t * KT, p.N, lds0 + (unsigned)(buf * 2 * TB), tid_);
        dma_tile<D, KT, 512>(vp, p.v_sn, t * KT, p.N, lds0 + (unsigned)(buf * 2 * TB + TB), tid_);
    };
    if (t_lo < t_hi) issue(t_lo, 0);
    pin_frags(qf);
    dma_wait_all();
    __syncthreads();
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1;
        const char* sK = smem + cur * 2 * TB;
        const char* sV = sK + TB;
        if (t + 1 < t_hi) issue(t + 1, cur ^ 1);             // next K/V tile streams into the other buffer during this tile
        for (int half = 0; half < KT / 64; ++half) {
        const int kv0 = t * KT + half * 64;
        if (kv0 >= kv_hi) break;                           // uniform: the second half of the last stage may be past the keys
        const char* sKh = sK + half * 64 * 2 * D;
        const char* sVh = sV + half * 64 * 2 * D;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
        {   // S^T = K Q^T: 16 (K row fragment, MFMA) pairs, fragments requested PF pairs ahead.  Left alone hipcc reuses ONE
            // destination register set (read -> lgkmcnt(0) -> MFMA, the LDS latency exposed 16 times); the order is pinned with
            // sched_group_barrier (mask 0x100 = LDS read, 0x008 = MFMA).
            constexpr int NP = 2 * (D / 16), PF = 3;
            bf16x8 ka[NP];
#pragma unroll
            for (int i = 0; i < PF; ++i) ka[i] = frag_row<D>(sKh, L, (i / (D / 16)) * 32, i % (D / 16));
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (i + PF < NP) ka[i + PF] = frag_row<D>(sKh, L, ((i + PF) / (D / 16)) * 32, (i + PF) % (D / 16));
                s[i / (D / 16)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[i], qf[i % (D / 16)], s[i / (D / 16)], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
            for (int i = 0; i < NP - PF; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
        }
        if (kv0 + 64 > kv_hi || windowed) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + acc_row(r, hh);
                    bool ok = key < len;
                    if (p.win_left >= 0) ok = ok && key >= qi - p.win_left;
                    if (p.win_right >= 0) ok = ok && key <= qi + p.win_right;
                    if (!ok) s[kt][r] = -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // Deferred rescale: m is the reference point of the exponentials, not necessarily the running maximum.  It moves (and
        // O, l are rescaled: 64 + 1 multiplies per lane) only when some row of the wave has outgrown it by more than 2^RESCALE_LOG2;
        // until then P <= 2^RESCALE_LOG2 instead of <= 1, which costs nothing (bf16 P keeps its relative precision, O and l stay
        // far inside f32 range) and O / l at the end is the same quotient.  The softmax VALU work is as long as the tile's
        // MFMAs here, so the skipped rescale is time, not just instructions.
        if (__any((mx - m) * c > RESCALE_LOG2)) {            // wave-uniform; also the first tile of every row (m = -inf)
            const float mn = fmaxf(m, mx);
            const float alpha = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m - mn) * c);
            l *= alpha;
            m = mn;
#pragma unroll
            for (int i = 0; i < D / 32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        const float mc = (m == -INFINITY) ? 0.f : m * c;
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] * c - mc); s[kt][r] = e; rs += e; }
        l += rs;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { pf[kt][0] = pack8(s[kt], 0); pf[kt][1] = pack8(s[kt], 1); }
        {   // O^T += V^T P^T: 16 (transposed V fragment = 2 tr reads, MFMA) pairs, same pinned pipeline
            constexpr int NP = 4 * (D / 32), PF = 2;
            bf16x8 va[NP];
#pragma unroll
            for (int i = 0; i < PF; ++i) va[i] = frag_tr<D>(sVh, L, ((i >> 1) & 1) * 32 + 16 * (i & 1), i >> 2);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (i + PF < NP) { const int j = i + PF; va[j] = frag_tr<D>(sVh, L, ((j >> 1) & 1) * 32 + 16 * (j & 1), j >> 2); }
                o[i >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[i], pf[(i >> 1) & 1][i & 1], o[i >> 2], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PF, 0);
#pragma unroll
            for (int i = 0; i < NP - PF; ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, PF, 0);
        }
        }
        dma_wait_all();                                      // the next stage has landed (issued a stage of MFMAs ago)
        __syncthreads();
    }
    const float lt = l + __shfl_xor(l, 32, 64);
    if (qi < p.N) {
        const bool live = qi < len && lt > 0.f;
        const float inv = live ? 1.f / lt : 0.f;
        store_t<D>(o, p.o + b * p.o_sb + (long)qi * p.o_sn + h * p.o_sh, inv, hh);
        if (hh == 0 && p.lse) p.lse[((long)b * p.H + h) * p.N + qi] = live ? (m * c + __log2f(lt)) * 0.6931471805599453f : INFINITY;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the 8-wave forward, software-pipelined inside the wave, without a running maximum.
//
// In-kernel stamps of the round-2 kernel above (B = 128, N = 2048; cycles per 64-key half-tile, waves 0-3 / waves 4-7):
//   DMA issue 474 / 990, S chain 742 / 1175, softmax 705 / 921, PV chain 635 / 709, stage barrier 1304 / 66  (sum 3930 for
//   2048 cycles of MFMA work per SIMD).  Three things, in that order:
//  (1) an LDS-DMA piece costs the issuing wave 75-180 cycles, and the younger half of the workgroup (waves 4-7 lose every
//      arbitration on their SIMD) was the critical path while waves 0-3 sat in the barrier: waves 0-3 issue ALL pieces now, through
//      a buffer descriptor (dma_tile128);
//  (2) S chain -> softmax -> PV chain were three serial segments per wave, and two waves in near lock-step do not overlap each
//      other's segments: the softmax of the first 32 keys now runs UNDER the S chain of the second 32 and that of the second 32
//      under the first half of the PV chain - possible because
//  (3) there is no per-tile maximum any more: the reference point m of the exponentials is the row maximum of the FIRST half-tile
//      a workgroup sees and never moves (any reference gives the same O / l; bf16 keeps P's relative precision at any magnitude).
//      Scores that later exceed it simply give P > 1, and f32 l, O have room for 2^100.  Only an overflow (or a row whose
//      reference was 2^126 above everything it sees later) breaks this: it shows up as a non-finite or zero l at the very end,
//      and the workgroup then redoes its rows with the careful per-tile maximum (pass 1 below).  No branch in the steady state.
// ---------------------------------------------------------------------------------------------------------------------------
// The softmax arithmetic of four scores as ONE asm statement of single VALU instructions.  Why asm: hipcc's SLP vectoriser packs
// neighbouring f32 operations into v_pk_mul / v_pk_add and pays two v_mov per pair (and an s_nop behind each dependent packed
// operation): MI355X_MICROARCH.md - packed f32 beside MFMAs is an anti-lever.  Why four at a time: inside an asm statement
// nobody pads hazards (cdna_hip_programming.md 5.7 item 2) - a VALU instruction that reads the result of the v_exp_f32 right in
// front of it gets a stale register (this cost a wrong dK once) - so the four fma, the four exponentials and the four
// consumers are issued in that order and every result is read four instructions after it was produced.
//   e[j] = exp2(s[j] * c + m);  l0 += e[0] + e[2];  l1 += e[1] + e[3]                                   (forward)
__device__ __forceinline__ void exp4_sum(float (&e)[4], float s0, float s1, float s2, float s3, float c, float m, float& l0, float& l1) {
    asm("v_fma_f32 %0, %6, %10, %11\n\tv_fma_f32 %1, %7, %10, %11\n\tv_fma_f32 %2, %8, %10, %11\n\tv_fma_f32 %3, %9, %10, %11\n\t"
        "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
        "v_add_f32 %4, %4, %0\n\tv_add_f32 %5, %5, %1\n\tv_add_f32 %4, %4, %2\n\tv_add_f32 %5, %5, %3"
        : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "+v"(l0), "+v"(l1)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(c), "v"(m));
}
//   p[j] = exp2(s[j] * c + nl[j]);  d[j] *= p[j]                                                        (backward: dS = P dP')
__device__ __forceinline__ void exp4_mul(float (&pr)[4], float s0, float s1, float s2, float s3, float c, float n0, float n1, float n2, float n3,
                                         float& d0, float& d1, float& d2, float& d3) {
    asm("v_fma_f32 %0, %8, %12, %13\n\tv_fma_f32 %1, %9, %12, %14\n\tv_fma_f32 %2, %10, %12, %15\n\tv_fma_f32 %3, %11, %12, %16\n\t"
        "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
        "v_mul_f32 %4, %4, %0\n\tv_mul_f32 %5, %5, %1\n\tv_mul_f32 %6, %6, %2\n\tv_mul_f32 %7, %7, %3"
        : "=&v"(pr[0]), "=&v"(pr[1]), "=&v"(pr[2]), "=&v"(pr[3]), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(c), "v"(n0), "v"(n1), "v"(n2), "v"(n3));
}
//   d[j] = exp2(s[j] * c + nl) * (d[j] + nd)                                                           (backward, query on the lane)
__device__ __forceinline__ void exp4_submul(float s0, float s1, float s2, float s3, float c, float nl, float nd, float& d0, float& d1, float& d2, float& d3) {
    float e0, e1, e2, e3;
    asm("v_fma_f32 %0, %8, %12, %13\n\tv_fma_f32 %1, %9, %12, %13\n\tv_fma_f32 %2, %10, %12, %13\n\tv_fma_f32 %3, %11, %12, %13\n\t"
        "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
        "v_add_f32 %4, %4, %14\n\tv_add_f32 %5, %5, %14\n\tv_add_f32 %6, %6, %14\n\tv_add_f32 %7, %7, %14\n\t"
        "v_mul_f32 %4, %4, %0\n\tv_mul_f32 %5, %5, %1\n\tv_mul_f32 %6, %6, %2\n\tv_mul_f32 %7, %7, %3"
        : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(c), "v"(nl), "v"(nd));
}
// hipcc pads the wait states between an MFMA and a VALU instruction IT emits that reads the result (18 after a 16-pass MFMA); an asm
// statement is not such an instruction.  The first asm statement that reads an MFMA result stands behind this guard: it makes
// the accumulator opaque at this point and spends the wait states.
__device__ __forceinline__ void mfma_guard(f32x16& a) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a)); }
__device__ __forceinline__ void mfma_guard(f32x16& a, f32x16& b) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b)); }

template <int D>
__global__ __launch_bounds__(512) void attn_fwd8p_kernel(const AttnParams p) {
    constexpr int KT = 128;                            // keys per stage, consumed as two 64-key halves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = KT * 2 * D;                     // bytes of one stage tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const BlkId bid = decode_block(p, (p.N + 255) / 256);
    const int b = bid.b, h = bid.h, qb0 = bid.x * 256, q0 = qb0 + wave * 32;
    const int len = p.lengths ? p.lengths[b] : p.N;
    const bf16* qp = p.q + b * p.q_sb + h * p.q_sh;
    const bf16* kp = p.k + b * p.k_sb + h * p.k_sh;
    const bf16* vp = p.v + b * p.v_sb + h * p.v_sh;
    const int qi = q0 + (lane & 31);
    const float c = p.scale * 1.4426950408889634f;

    bf16x8 qf[D / 16];
    load_bfrags<D>(qf, qp, p.q_sn, q0, p.N, lane);
    const LaneOffs<D> L(lane);

    const int kv_lo = p.win_left < 0 ? 0 : max(0, qb0 - p.win_left);
    const int kv_hi = min(len, p.win_right < 0 ? len : qb0 + 256 + p.win_right);
    const int t_lo = kv_lo / KT, t_hi = (kv_hi + KT - 1) / KT;
    const bool windowed = p.win_left >= 0 || p.win_right >= 0;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const srd_t ksrd = make_srd(kp, view_bytes<D>(p.N, p.k_sn)), vsrd = make_srd(vp, view_bytes<D>(p.N, p.v_sn));
    const unsigned kvoff = tile_voff<D, 512>(p.k_sn, tid), vvoff = tile_voff<D, 512>(p.v_sn, tid);
    auto issue = [&](int t, int buf) {                     // K | V stage by LDS-DMA through buffer descriptors, issued by waves 0-3 (dma_tile128)
        dma_tile128<KT, true>(ksrd, kvoff, p.k_sn, t * KT, lds0 + (unsigned)(buf * 2 * TB), wave_u);
        dma_tile128<KT, true>(vsrd, vvoff, p.v_sn, t * KT, lds0 + (unsigned)(buf * 2 * TB + TB), wave_u);
    };
    // keys this lane's query may see: [key_lo, key_lo + key_rng] (length and window folded into one unsigned range test)
    const int key_lo = p.win_left >= 0 ? max(0, qi - p.win_left) : 0;
    const unsigned key_rng = (unsigned)max(-1, (p.win_right >= 0 ? min(len - 1, qi + p.win_right) : len - 1) - key_lo);   // -1 -> none... as unsigned: all
    const bool none = (p.win_right >= 0 ? min(len - 1, qi + p.win_right) : len - 1) < key_lo;
    auto mask_block = [&](f32x16& sb, int key0) {          // keys key0 + acc_row(r, hh): -inf outside length / window, branch-free
        const unsigned t0 = (unsigned)(key0 + 4 * hh - key_lo);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool ok = (t0 + (unsigned)acc_row(r, 0)) <= key_rng && !none;
            sb[r] = ok ? sb[r] : -INFINITY;
        }
    };
    auto s_chain = [&](const char* sKh, int kt, f32x16& acc) {   // acc = K[32 kt .. +31] Q^T: 8 (fragment read, MFMA) pairs, reads 3 ahead
        constexpr int NS = D / 16, PF = 3;
        bf16x8 ka[NS];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < PF; ++i) ka[i] = frag_row<D>(sKh, L, kt * 32, i);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i + PF < NS) ka[i + PF] = frag_row<D>(sKh, L, kt * 32, i + PF);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[i], qf[i], acc, 0, 0, 0);
        }
    };

    f32x16 o[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float mc = 0.f, l0 = 0.f, l1 = 0.f;                // reference point (scaled, log2 units) and the row sum relative to it (two chains)
    if (t_lo < t_hi) issue(t_lo, 0);
    pin_frags(qf);
    dma_wait_all();
    __syncthreads();
    if (t_lo < t_hi) {
        // the reference: this lane's row maximum over the first half-tile (0 for a row that sees no key there)
        const int kv0 = t_lo * KT;
        f32x16 s0, s1;
        s_chain(smem, 0, s0); s_chain(smem, 1, s1);
        if (kv0 + 64 > kv_hi || windowed) { mask_block(s0, kv0); mask_block(s1, kv0 + 32); }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        mc = (mx > -INFINITY) ? mx * c : 0.f;
    }
    const float nmc = -mc;
    STAMP_DECL(8)
    // One 64-key half-tile: [S0] [S1 | exp S0] [PV0 | exp S1] [PV1] - the exponentials of one 32-key block run under the MFMAs of
    // the next group (independent instruction streams in one basic block: hipcc interleaves them).
    auto exp_block = [&](f32x16& sb, bf16x8& plo, bf16x8& phi) {
        mfma_guard(sb);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float e[4];
            exp4_sum(e, sb[4 * g], sb[4 * g + 1], sb[4 * g + 2], sb[4 * g + 3], c, nmc, l0, l1);
#pragma unroll
            for (int j = 0; j < 4; ++j) sb[4 * g + j] = e[j];
        }
        plo = pack8(sb, 0); phi = pack8(sb, 1);
    };
    auto half_tile = [&](const char* sKh, const char* sVh, int kv0, const bool MASKED) {      // MASKED: wave-uniform
        f32x16 s0, s1;
        bf16x8 pf[2][2];
        s_chain(sKh, 0, s0);
        if (MASKED) mask_block(s0, kv0);
        STAMP(1);
        s_chain(sKh, 1, s1);
        exp_block(s0, pf[0][0], pf[0][1]);
        if (MASKED) mask_block(s1, kv0 + 32);
        STAMP(2);
#pragma unroll
        for (int i = 0; i < 8; ++i)              // PV of keys 0..31: d-block i >> 1, k-step i & 1
            o[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sVh, L, 16 * (i & 1), i >> 1), pf[0][i & 1], o[i >> 1], 0, 0, 0);
        exp_block(s1, pf[1][0], pf[1][1]);
        STAMP(3);
#pragma unroll
        for (int i = 0; i < 8; ++i)              // PV of keys 32..63
            o[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sVh, L, 32 + 16 * (i & 1), i >> 1), pf[1][i & 1], o[i >> 1], 0, 0, 0);
        STAMP(4);
    };
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1;
        const char* sK = smem + cur * 2 * TB;
        const char* sV = sK + TB;
        if (t + 1 < t_hi) issue(t + 1, cur ^ 1);             // next K/V tile streams into the other buffer during this tile
        STAMP(0);
        for (int half = 0; half < KT / 64; ++half) {
            const int kv0 = t * KT + half * 64;
            if (kv0 >= kv_hi) break;                       // uniform: the second half of the last stage may be past the keys
            const char* sKh = sK + half * 64 * 2 * D;
            const char* sVh = sV + half * 64 * 2 * D;
            half_tile(sKh, sVh, kv0, kv0 + 64 > kv_hi || windowed);
        }
        dma_wait_all();                                      // the next stage has landed (issued a stage of MFMAs ago)
        __syncthreads();
        STAMP(5);
    }
    STAMP_OUT(6);
    float l = l0 + l1;
    float lt = l + __shfl_xor(l, 32, 64);
    float mfin = mc;
    if (__syncthreads_or(qi < len && qi < p.N && !(lt > 0.f && lt < 1e37f))) {
        // ---- rare: an exponential overflowed (or a row's reference sat 2^126 above everything it saw later): redo this workgroup's
        // rows with the running reference of the round-2 kernel (per-tile maximum, rescale when a row outgrows it by 2^RESCALE_LOG2)
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
        float m = -INFINITY;
        l = 0.f;
        if (t_lo < t_hi) issue(t_lo, 0);
        dma_wait_all();
        __syncthreads();
        for (int t = t_lo; t < t_hi; ++t) {
            const int cur = (t - t_lo) & 1;
            const char* sK = smem + cur * 2 * TB;
            const char* sV = sK + TB;
            if (t + 1 < t_hi) issue(t + 1, cur ^ 1);
            for (int half = 0; half < KT / 64; ++half) {
                const int kv0 = t * KT + half * 64;
                if (kv0 >= kv_hi) break;
                const char* sKh = sK + half * 64 * 2 * D;
                const char* sVh = sV + half * 64 * 2 * D;
                f32x16 s0, s1;
                s_chain(sKh, 0, s0); s_chain(sKh, 1, s1);
                if (kv0 + 64 > kv_hi || windowed) { mask_block(s0, kv0); mask_block(s1, kv0 + 32); }
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (__any((mx - m) * c > RESCALE_LOG2)) {    // wave-uniform; also the first tile of every row (m = -inf)
                    const float mn = fmaxf(m, mx);
                    const float alpha = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m - mn) * c);
                    l *= alpha;
                    m = mn;
#pragma unroll
                    for (int i = 0; i < D / 32; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                }
                const float mcc = (m == -INFINITY) ? 0.f : m * c;
                float rs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s0[r] = __builtin_amdgcn_exp2f(s0[r] * c - mcc); s1[r] = __builtin_amdgcn_exp2f(s1[r] * c - mcc);
                    rs += s0[r] + s1[r];
                }
                l += rs;
                const bf16x8 p00 = pack8(s0, 0), p01 = pack8(s0, 1), p10 = pack8(s1, 0), p11 = pack8(s1, 1);
#pragma unroll
                for (int db = 0; db < D / 32; ++db) {
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sVh, L, 0, db), p00, o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sVh, L, 16, db), p01, o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sVh, L, 32, db), p10, o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sVh, L, 48, db), p11, o[db], 0, 0, 0);
                }
            }
            dma_wait_all();
            __syncthreads();
        }
        lt = l + __shfl_xor(l, 32, 64);
        mfin = (m == -INFINITY) ? 0.f : m * c;
    }
    if (qi < p.N) {
        const bool live = qi < len && lt > 0.f;
        const float inv = live ? 1.f / lt : 0.f;
        store_t<D>(o, p.o + b * p.o_sb + (long)qi * p.o_sn + h * p.o_sh, inv, hh);
        if (hh == 0 && p.lse) p.lse[((long)b * p.H + h) * p.N + qi] = live ? (mfin + __log2f(lt)) * 0.6931471805599453f : INFINITY;
    }
}

// delta[b][h][n] = sum_d dO * O
template <int D>
__global__ void attn_delta_kernel(const AttnParams p) {
    constexpr int LPR = D / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long rows = (long)p.B * p.N * p.H;
    const long row = idx / LPR;
    const int ch = (int)(idx % LPR);
    float acc = 0.f;
    long bb = 0, n = 0, hd = 0;
    if (row < rows) {
        hd = row % p.H; n = (row / p.H) % p.N; bb = row / ((long)p.H * p.N);
        float a[8], g[8];
        load8(p.o + bb * p.o_sb + n * p.o_sn + hd * p.o_sh + ch * 8, a);
        load8(p.dout + bb * p.do_sb + n * p.do_sn + hd * p.do_sh + ch * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += a[e] * g[e];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row < rows && ch == 0) p.delta[(bb * p.H + hd) * p.N + n] = acc;
}

// =============================================================================================
// backward dK/dV: grid (ceil(N/128), H, B); each wave owns 32 keys, sweeps 32-row query tiles
// =============================================================================================
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 32 * 2 * D;                     // Q tile / dO tile bytes (32 rows)
    constexpr int SB = 2 * TB + 256;                   // one stage: Q | dO | lse2[32] | delta[32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const BlkId bid = decode_block(p, (p.N + 127) / 128);
    const int b = bid.b, h = bid.h, kb0 = bid.x * 128, k0 = kb0 + wave * 32;
    const int len = p.lengths ? p.lengths[b] : p.N;
    const bf16* qp = p.q + b * p.q_sb + h * p.q_sh;
    const bf16* kp = p.k + b * p.k_sb + h * p.k_sh;
    const bf16* vp = p.v + b * p.v_sb + h * p.v_sh;
    const bf16* gp = p.dout + b * p.do_sb + h * p.do_sh;
    const float* delp = p.delta + ((long)b * p.H + h) * p.N;                       // -delta        } written by the dQ kernel,
    const float* lsep = delp + (long)p.B * p.H * p.N;                                // -lse log2(e)  } which runs first
    const int key = k0 + (lane & 31);
    const float c = p.scale * 1.4426950408889634f;

    bf16x8 kf[D / 16];
    load_bfrags<D>(kf, kp, p.k_sn, k0, p.N, lane);
    const LaneOffs<D> L(lane);
    // this workgroup's 128 V rows live in LDS for the whole kernel (B operand of dP = dO V^T, read per iteration):
    // keeping them in registers next to K, dK^T and dV^T (64 + 128 VGPRs) spills into the loop.
    char* sVt = smem + 2 * SB;
    glds_tile<D, 128>(vp, p.v_sn, kb0, p.N, sVt, tid);
    const char* sVw = sVt + wave * 32 * 2 * D;
    const bool need_mask = p.win_left >= 0 || p.win_right >= 0 || kb0 + 128 > len;   // uniform per workgroup

    // queries that can see this block's keys: key in [q-left, q+right]  <=>  q in [key-right, key+left]
    const int q_lo = p.win_right < 0 ? 0 : max(0, kb0 - p.win_right);
    const int q_hi = min(len, p.win_left < 0 ? len : kb0 + 128 + p.win_left);
    const int t_lo = q_lo / 32, t_hi = (kb0 < len) ? (q_hi + 31) / 32 : t_lo;

    f32x16 dkt[D / 32], dvt[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[i][r] = 0.f; dvt[i][r] = 0.f; }

    float st_l = 0.f, st_d = 0.f;
    auto gload_stats = [&](int q0) {
        if (tid < 32) { const int q = q0 + tid; st_l = (q < len) ? -lsep[q] : INFINITY; st_d = (q < len) ? -delp[q] : 0.f; }
    };
    auto lstore_stats = [&](char* s) {
        if (tid < 32) { reinterpret_cast<float*>(s + 2 * TB)[tid] = st_l; reinterpret_cast<float*>(s + 2 * TB + 128)[tid] = st_d; }
    };
    if (t_lo < t_hi) {
        glds_tile<D, 32>(qp, p.q_sn, t_lo * 32, p.N, smem, tid); glds_tile<D, 32>(gp, p.do_sn, t_lo * 32, p.N, smem + TB, tid);
        gload_stats(t_lo * 32); lstore_stats(smem);
    }
    __syncthreads();
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1;
        const char* sQ = smem + cur * SB;
        const char* sG = sQ + TB;
        const float* sL = reinterpret_cast<const float*>(sQ + 2 * TB);
        const float* sD = sL + 32;
        if (t + 1 < t_hi) {
            char* dS = smem + (cur ^ 1) * SB;
            glds_tile<D, 32>(qp, p.q_sn, (t + 1) * 32, p.N, dS, tid); glds_tile<D, 32>(gp, p.do_sn, (t + 1) * 32, p.N, dS + TB, tid);
            gload_stats((t + 1) * 32);
        }
        const int q0 = t * 32;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int st = 0; st < D / 16; ++st) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sQ, L, 0, st), kf[st], s, 0, 0, 0);     // S[q][key]
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sG, L, 0, st), frag_row<D>(sVw, L, 0, st), dp, 0, 0, 0);   // dP[q][key]
        }
        // row statistics of this lane's 16 query rows: 4 consecutive rows per b128 read (rows 8g + 4hh + 0..3).
        // Two straight-line bodies (mask-free / masked) chosen by ONE uniform branch.
        if (!need_mask) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = *reinterpret_cast<const float4*>(sL + 8 * g + 4 * hh);
                const float4 b_ = *reinterpret_cast<const float4*>(sD + 8 * g + 4 * hh);
                const float la[4] = {a.x, a.y, a.z, a.w}, da[4] = {b_.x, b_.y, b_.z, b_.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pr = __builtin_amdgcn_exp2f(s[r] * c - la[e]);   // lse2 = +inf for q >= len -> 0
                    s[r] = pr;
                    dp[r] = pr * (dp[r] - da[e]);
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = *reinterpret_cast<const float4*>(sL + 8 * g + 4 * hh);
                const float4 b_ = *reinterpret_cast<const float4*>(sD + 8 * g + 4 * hh);
                const float la[4] = {a.x, a.y, a.z, a.w}, da[4] = {b_.x, b_.y, b_.z, b_.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const int q = q0 + acc_row(r, hh);
                    bool ok = key < len;
                    if (p.win_left >= 0) ok = ok && key >= q - p.win_left;
                    if (p.win_right >= 0) ok = ok && key <= q + p.win_right;
                    const float pr = ok ? __builtin_amdgcn_exp2f(s[r] * c - la[e]) : 0.f;
                    s[r] = pr;
                    dp[r] = pr * (dp[r] - da[e]);
                }
            }
        }
        const bf16x8 pb0 = pack8(s, 0), pb1 = pack8(s, 1), db0 = pack8(dp, 0), db1 = pack8(dp, 1);
#pragma unroll
        for (int db = 0; db < D / 32; ++db) {
            dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sG, L, 0, db), pb0, dvt[db], 0, 0, 0);
            dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sG, L, 16, db), pb1, dvt[db], 0, 0, 0);
            dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sQ, L, 0, db), db0, dkt[db], 0, 0, 0);
            dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sQ, L, 16, db), db1, dkt[db], 0, 0, 0);
        }
        if (t + 1 < t_hi) lstore_stats(smem + (cur ^ 1) * SB);
        __syncthreads();
    }
    if (key < p.N) {
        if (p.rot_cos) store_t_rot<D>(dkt, p.dk + b * p.dk_sb + (long)key * p.dk_sn + h * p.dk_sh, p.scale, lane >> 5, p.rot_cos + (long)key * (D / 2), p.rot_sin + (long)key * (D / 2));
        else store_t<D>(dkt, p.dk + b * p.dk_sb + (long)key * p.dk_sn + h * p.dk_sh, p.scale, lane >> 5);
        store_t<D>(dvt, p.dv + b * p.dv_sb + (long)key * p.dv_sn + h * p.dv_sh, 1.f, lane >> 5);
    }
}

// =============================================================================================
// backward dK/dV, 8-wave form (D = 128): grid (ceil(N/256), H, B), 512 threads; each wave owns 32 keys, the workgroup's 256 V
// rows live in LDS (64 KiB) and the query side streams in 64-row stages (2 x 32.5 KiB) shared by all 8 waves: half the
// LDS-DMA bytes per MFMA of the 4-wave kernel and one barrier per 64 MFMAs instead of 32.  One workgroup per CU.
// =============================================================================================
template <int D>
__global__ __launch_bounds__(512) void attn_bwd_dkdv8_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QR = 64;                             // query rows per stage
    constexpr int TB = QR * 2 * D;                     // Q tile / dO tile bytes
    constexpr int SB = 2 * TB + 512;                   // one stage: Q | dO | lse2[64] | delta[64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const BlkId bid = decode_block(p, (p.N + 255) / 256);
    const int b = bid.b, h = bid.h, kb0 = bid.x * 256, k0 = kb0 + wave * 32;
    const int len = p.lengths ? p.lengths[b] : p.N;
    const bf16* qp = p.q + b * p.q_sb + h * p.q_sh;
    const bf16* kp = p.k + b * p.k_sb + h * p.k_sh;
    const bf16* vp = p.v + b * p.v_sb + h * p.v_sh;
    const bf16* gp = p.dout + b * p.do_sb + h * p.do_sh;
    const float* delp = p.delta + ((long)b * p.H + h) * p.N;                       // -delta        } written by the dQ kernel,
    const float* lsep = delp + (long)p.B * p.H * p.N;                                // -lse log2(e)  } which runs first
    const int key = k0 + (lane & 31);
    const float c = p.scale * 1.4426950408889634f;

    bf16x8 kf[D / 16];
    load_bfrags<D>(kf, kp, p.k_sn, k0, p.N, lane);
    const LaneOffs<D> L(lane);
    char* sVt = smem + 2 * SB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    dma_tile<D, 256, 512>(vp, p.v_sn, kb0, p.N, lds0 + (unsigned)(2 * SB), tid);
    const char* sVw = sVt + wave * 32 * 2 * D;
    const bool mask_wg = p.win_left >= 0 || p.win_right >= 0 || kb0 + 256 > len;     // uniform per workgroup
    // queries that may see this lane's key: [q_first, q_first + q_range]  (key in [q - left, q + right], q < len, key < len)
    const int q_first = p.win_right >= 0 ? max(0, key - p.win_right) : 0;
    const int q_last = key < len ? (p.win_left >= 0 ? min(len - 1, key + p.win_left) : len - 1) : -1;
    const unsigned q_range = q_last >= q_first ? (unsigned)(q_last - q_first) : 0u;
    const bool q_none = q_last < q_first;

    const int q_lo = p.win_right < 0 ? 0 : max(0, kb0 - p.win_right);
    const int q_hi = min(len, p.win_left < 0 ? len : kb0 + 256 + p.win_left);
    const int t_lo = q_lo / QR, t_hi = (kb0 < len) ? (q_hi + QR - 1) / QR : t_lo;

    f32x16 dkt[D / 32], dvt[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkt[i][r] = 0.f; dvt[i][r] = 0.f; }

    // a stage = Q | dO | lse[64] | delta[64], all by asm-issued LDS-DMA (the statistics as two 4-byte-per-lane pieces): the loop
    // holds no vector-memory operation the compiler knows of, so none of its waits can drain the DMA in flight
    const srd_t qsrd = make_srd(qp, view_bytes<D>(p.N, p.q_sn)), gsrd = make_srd(gp, view_bytes<D>(p.N, p.do_sn));
    auto issue = [&](int t, int buf) {
        const int tid_ = opaque(tid);                      // the two per-lane offsets are rebuilt per stage (a dozen VALU operations): kept
        const unsigned qvoff = tile_voff<D, 512>(p.q_sn, tid_), gvoff = tile_voff<D, 512>(p.do_sn, tid_);   // live across the loop they spill
        const int wave_u = __builtin_amdgcn_readfirstlane(tid_ >> 6);    // scalar: the pointer select below stays in SGPRs
        dma_tile128<QR, true>(qsrd, qvoff, p.q_sn, t * QR, lds0 + (unsigned)(buf * SB), wave_u);          // issued by waves 0-3 (dma_tile128)
        dma_tile128<QR, true>(gsrd, gvoff, p.do_sn, t * QR, lds0 + (unsigned)(buf * SB + TB), wave_u);
        if (wave_u < 2) {                                  // wave 0: lse, wave 1: delta; rows past the tensor are clamped and masked below
            const int q = min(t * QR + (tid_ & 63), p.N - 1);
            dma4_asm((wave_u == 0 ? lsep : delp) + q, lds0 + (unsigned)(buf * SB + 2 * TB) + 256u * (unsigned)wave_u);
        }
    };
    if (t_lo < t_hi) issue(t_lo, 0);
    pin_frags(kf);
    dma_wait_all();
    __syncthreads();
    STAMP_DECL(8)
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1;
        const char* sQ = smem + cur * SB;
        const char* sG = sQ + TB;
        const float* sL = reinterpret_cast<const float*>(sQ + 2 * TB);
        const float* sD = sL + 64;
        if (t + 1 < t_hi) issue(t + 1, cur ^ 1);
        const bool need_mask = mask_wg || (t + 1) * QR > len;                         // uniform
        const int x32 = opaque(32);                    // see frag_tr: keeps 4 derived offsets out of the (full) register file
        const int hh = (opaque(tid) >> 5) & 1;         // re-derived per stage for the same reason (shadows the kernel-scope hh)
        STAMP(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {            // two 32-query sub-tiles per stage
            const int rb = 32 * sub, q0 = t * QR + rb;
            // S = Q K^T from zero; dP' = dO V^T - delta: the chain STARTS from the rows' -delta (16 accumulator rows = 16 queries,
            // read straight from the stage's statistics in LDS), so dS = P dP' needs no subtraction.  Then per score: one fma
            // (scale and -lse), one exponential, one multiply - as single VALU instructions (asm helpers): left to itself hipcc
            // packs pairs into v_pk_mul / v_pk_add and pays two v_mov per pair for it.
            f32x16 s, dp;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 d4 = *reinterpret_cast<const float4*>(sD + rb + 8 * g + 4 * hh);
                dp[4 * g] = d4.x; dp[4 * g + 1] = d4.y; dp[4 * g + 2] = d4.z; dp[4 * g + 3] = d4.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int st = 0; st < D / 16; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sQ, L, rb, st), kf[st], s, 0, 0, 0);     // S[q][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sG, L, rb, st), frag_row<D>(sVw, L, 0, st), dp, 0, 0, 0);
            }
            STAMP(1);
            // Masked scores become -inf BEFORE the exponential (one unsigned range test and a select per score, branch-free): written
            // as `ok ? exp2(..) : 0` hipcc branched around every single exponential - 16 s_and_saveexec / s_cbranch pairs per
            // sub-tile, 1700-1950 cycles for ~460 cycles of arithmetic (in-kernel stamps, round 3), in every tile, masked or not.
            if (need_mask) {                               // uniform
                const unsigned t0 = (unsigned)(q0 + 4 * hh - q_first);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (t0 + (unsigned)acc_row(r, 0) <= q_range && !q_none) ? s[r] : -INFINITY;
            }
            mfma_guard(s, dp);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 a = *reinterpret_cast<const float4*>(sL + rb + 8 * g + 4 * hh);     // -lse log2(e); -inf for q >= len -> p = 0
                float pr[4], d0 = dp[4 * g], d1 = dp[4 * g + 1], d2 = dp[4 * g + 2], d3 = dp[4 * g + 3];
                exp4_mul(pr, s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3], c, a.x, a.y, a.z, a.w, d0, d1, d2, d3);
                dp[4 * g] = d0; dp[4 * g + 1] = d1; dp[4 * g + 2] = d2; dp[4 * g + 3] = d3;
#pragma unroll
                for (int e = 0; e < 4; ++e) s[4 * g + e] = pr[e];
            }
            bf16x8 pb0 = pack8(s, 0), pb1 = pack8(s, 1), db0 = pack8(dp, 0), db1 = pack8(dp, 1);
#ifdef SCONF_ATTN_STAMP
            asm volatile("" : "+v"(pb0)); asm volatile("" : "+v"(pb1)); asm volatile("" : "+v"(db0)); asm volatile("" : "+v"(db1));
#endif
            STAMP(2);
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sG, L, rb, db, x32), pb0, dvt[db], 0, 0, 0);
                dvt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sG, L, rb + 16, db, x32), pb1, dvt[db], 0, 0, 0);
                dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sQ, L, rb, db, x32), db0, dkt[db], 0, 0, 0);
                dkt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sQ, L, rb + 16, db, x32), db1, dkt[db], 0, 0, 0);
            }
            STAMP(3);
        }
        dma_wait_all();                                // the next stage has landed (issued a stage of MFMAs ago)
        __syncthreads();
        STAMP(4);
    }
    STAMP_OUT(5);
    if (key < p.N) {
        if (p.rot_cos) store_t_rot<D>(dkt, p.dk + b * p.dk_sb + (long)key * p.dk_sn + h * p.dk_sh, p.scale, lane >> 5, p.rot_cos + (long)key * (D / 2), p.rot_sin + (long)key * (D / 2));
        else store_t<D>(dkt, p.dk + b * p.dk_sb + (long)key * p.dk_sn + h * p.dk_sh, p.scale, lane >> 5);
        store_t<D>(dvt, p.dv + b * p.dv_sb + (long)key * p.dv_sn + h * p.dv_sh, 1.f, lane >> 5);
    }
}

// =============================================================================================
// backward dQ: grid (ceil(N/128), H, B); each wave owns 32 queries, sweeps 64-key tiles
// =============================================================================================
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = 64 * 2 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const BlkId bid = decode_block(p, (p.N + 127) / 128);
    const int b = bid.b, h = bid.h, qb0 = bid.x * 128, q0 = qb0 + wave * 32;
    const int len = p.lengths ? p.lengths[b] : p.N;
    const bf16* qp = p.q + b * p.q_sb + h * p.q_sh;
    const bf16* kp = p.k + b * p.k_sb + h * p.k_sh;
    const bf16* vp = p.v + b * p.v_sb + h * p.v_sh;
    const bf16* gp = p.dout + b * p.do_sb + h * p.do_sh;
    const int qi = q0 + (lane & 31);
    const float c = p.scale * 1.4426950408889634f;
    const float lse2 = (qi < len) ? p.lse[((long)b * p.H + h) * p.N + qi] * 1.4426950408889634f : INFINITY;

    bf16x8 qf[D / 16], gf[D / 16];
    load_bfrags<D>(qf, qp, p.q_sn, q0, p.N, lane);
    load_bfrags<D>(gf, gp, p.do_sn, q0, p.N, lane);
    float dlt = delta_from_frags<D>(gf, p.o + b * p.o_sb + h * p.o_sh, p.o_sn, q0, p.N, lane);
    if (qi >= len) dlt = 0.f;
    // row statistics for the dK/dV kernel, in the form its MFMA chains and exponentials take them: -delta and -lse * log2(e)
    // (-inf for rows past the sample's length: their probabilities come out 0)
    if (hh == 0 && qi < p.N) {
        const long si = ((long)b * p.H + h) * p.N + qi;
        p.delta[si] = -dlt;
        p.delta[(long)p.B * p.H * p.N + si] = -lse2;
    }
    const LaneOffs<D> L(lane);
    const bool windowed = p.win_left >= 0 || p.win_right >= 0;

    const int kv_lo = p.win_left < 0 ? 0 : max(0, qb0 - p.win_left);
    const int kv_hi = min(len, p.win_right < 0 ? len : qb0 + 128 + p.win_right);
    const int t_lo = kv_lo / 64, t_hi = (qb0 < len) ? (kv_hi + 63) / 64 : t_lo;

    f32x16 dqt[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[i][r] = 0.f;

    if (t_lo < t_hi) { glds_tile<D, 64>(kp, p.k_sn, t_lo * 64, p.N, smem, tid); glds_tile<D, 64>(vp, p.v_sn, t_lo * 64, p.N, smem + TB, tid); }
    __syncthreads();
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1;
        const char* sK = smem + cur * 2 * TB;
        const char* sV = sK + TB;
        if (t + 1 < t_hi) {
            char* dK = smem + (cur ^ 1) * 2 * TB;
            glds_tile<D, 64>(kp, p.k_sn, (t + 1) * 64, p.N, dK, tid); glds_tile<D, 64>(vp, p.v_sn, (t + 1) * 64, p.N, dK + TB, tid);
        }
        const int kv0 = t * 64;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < D / 16; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sK, L, kt * 32, st), qf[st], s, 0, 0, 0);    // S^T[key][q]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sV, L, kt * 32, st), gf[st], dp, 0, 0, 0);  // dP^T[key][q]
            }
            if (!(windowed || kv0 + 64 > kv_hi)) {                               // ONE uniform branch, two straight-line bodies
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = __builtin_amdgcn_exp2f(s[r] * c - lse2) * (dp[r] - dlt);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kt * 32 + acc_row(r, hh);
                    bool ok = key < len;
                    if (p.win_left >= 0) ok = ok && key >= qi - p.win_left;
                    if (p.win_right >= 0) ok = ok && key <= qi + p.win_right;
                    dp[r] = ok ? __builtin_amdgcn_exp2f(s[r] * c - lse2) * (dp[r] - dlt) : 0.f;
                }
            }
            const bf16x8 d0 = pack8(dp, 0), d1 = pack8(dp, 1);
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sK, L, kt * 32, db), d0, dqt[db], 0, 0, 0);
                dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sK, L, kt * 32 + 16, db), d1, dqt[db], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (qi < p.N) {
        if (p.rot_cos) store_t_rot<D>(dqt, p.dq + b * p.dq_sb + (long)qi * p.dq_sn + h * p.dq_sh, p.scale, hh, p.rot_cos + (long)qi * (D / 2), p.rot_sin + (long)qi * (D / 2));
        else store_t<D>(dqt, p.dq + b * p.dq_sb + (long)qi * p.dq_sn + h * p.dq_sh, p.scale, hh);
    }
}

// 8-wave form of the dQ kernel (D = 128): 256 queries per workgroup, 128-key stages shared by 8 waves (see attn_bwd_dkdv8_kernel)
template <int D>
__global__ __launch_bounds__(512) void attn_bwd_dq8_kernel(const AttnParams p) {
    constexpr int KT = 128;                            // keys per stage (two 64 KiB stages: one workgroup per CU)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TB = KT * 2 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
    const BlkId bid = decode_block(p, (p.N + 255) / 256);
    const int b = bid.b, h = bid.h, qb0 = bid.x * 256, q0 = qb0 + wave * 32;
    const int len = p.lengths ? p.lengths[b] : p.N;
    const bf16* qp = p.q + b * p.q_sb + h * p.q_sh;
    const bf16* kp = p.k + b * p.k_sb + h * p.k_sh;
    const bf16* vp = p.v + b * p.v_sb + h * p.v_sh;
    const bf16* gp = p.dout + b * p.do_sb + h * p.do_sh;
    const int qi = q0 + (lane & 31);
    const float c = p.scale * 1.4426950408889634f;
    const float lse2 = (qi < len) ? p.lse[((long)b * p.H + h) * p.N + qi] * 1.4426950408889634f : INFINITY;

    bf16x8 qf[D / 16], gf[D / 16];
    load_bfrags<D>(qf, qp, p.q_sn, q0, p.N, lane);
    load_bfrags<D>(gf, gp, p.do_sn, q0, p.N, lane);
    float dlt = delta_from_frags<D>(gf, p.o + b * p.o_sb + h * p.o_sh, p.o_sn, q0, p.N, lane);
    if (qi >= len) dlt = 0.f;
    // row statistics for the dK/dV kernel, in the form its MFMA chains and exponentials take them: -delta and -lse * log2(e)
    // (-inf for rows past the sample's length: their probabilities come out 0)
    if (hh == 0 && qi < p.N) {
        const long si = ((long)b * p.H + h) * p.N + qi;
        p.delta[si] = -dlt;
        p.delta[(long)p.B * p.H * p.N + si] = -lse2;
    }
    const LaneOffs<D> L(lane);
    const bool windowed = p.win_left >= 0 || p.win_right >= 0;
    // keys this lane's query may see: [key_lo, key_lo + key_rng]
    const int key_lo = p.win_left >= 0 ? max(0, qi - p.win_left) : 0;
    const int key_hi = p.win_right >= 0 ? min(len - 1, qi + p.win_right) : len - 1;
    const unsigned key_rng = key_hi >= key_lo ? (unsigned)(key_hi - key_lo) : 0u;
    const bool key_none = key_hi < key_lo;

    const int kv_lo = p.win_left < 0 ? 0 : max(0, qb0 - p.win_left);
    const int kv_hi = min(len, p.win_right < 0 ? len : qb0 + 256 + p.win_right);
    const int t_lo = kv_lo / KT, t_hi = (qb0 < len) ? (kv_hi + KT - 1) / KT : t_lo;

    f32x16 dqt[D / 32];
#pragma unroll
    for (int i = 0; i < D / 32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[i][r] = 0.f;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const srd_t ksrd = make_srd(kp, view_bytes<D>(p.N, p.k_sn)), vsrd = make_srd(vp, view_bytes<D>(p.N, p.v_sn));
    const unsigned kvoff = tile_voff<D, 512>(p.k_sn, tid), vvoff = tile_voff<D, 512>(p.v_sn, tid);
    auto issue = [&](int t, int buf) {                     // K | V stage by LDS-DMA through buffer descriptors, issued by waves 0-3 (dma_tile128)
        dma_tile128<KT, true>(ksrd, kvoff, p.k_sn, t * KT, lds0 + (unsigned)(buf * 2 * TB), wave_u);
        dma_tile128<KT, true>(vsrd, vvoff, p.v_sn, t * KT, lds0 + (unsigned)(buf * 2 * TB + TB), wave_u);
    };
    if (t_lo < t_hi) issue(t_lo, 0);
    pin_frags(qf); pin_frags(gf);
    dma_wait_all();
    __syncthreads();
    STAMP_DECL(8)
    for (int t = t_lo; t < t_hi; ++t) {
        const int cur = (t - t_lo) & 1;
        const char* sK = smem + cur * 2 * TB;
        const char* sV = sK + TB;
        if (t + 1 < t_hi) issue(t + 1, cur ^ 1);
        STAMP(0);
        const int kv0 = t * KT;
        // Software-pipelined over the four 32-key blocks of the stage: the S | dP chains of block kt + 1 are issued BEFORE the
        // exponentials of block kt, which then run under those MFMAs (two sets of score accumulators; stamps of the serial form:
        // 16 MFMAs, then 750 cycles with the matrix pipe idle in both waves of the SIMD at once, then 8 MFMAs).
        const bool masked = windowed || kv0 + KT > kv_hi;                        // uniform
        const float nl = -lse2, nd = -dlt;
        f32x16 s2[2], dp2[2];
        auto chains = [&](int kt, f32x16& s, f32x16& dp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int st = 0; st < D / 16; ++st) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sK, L, kt * 32, st), qf[st], s, 0, 0, 0);    // S^T[key][q]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<D>(sV, L, kt * 32, st), gf[st], dp, 0, 0, 0);  // dP^T[key][q]
            }
        };
        chains(0, s2[0], dp2[0]);
        STAMP(1);
#pragma unroll
        for (int kt = 0; kt < KT / 32; ++kt) {
            f32x16& s = s2[kt & 1];
            f32x16& dp = dp2[kt & 1];
            if (kt + 1 < KT / 32) chains(kt + 1, s2[(kt + 1) & 1], dp2[(kt + 1) & 1]);
            if (masked) {                                                        // masked scores become -inf before the exponential (branch-free)
                const unsigned t0 = (unsigned)(kv0 + kt * 32 + 4 * hh - key_lo);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (t0 + (unsigned)acc_row(r, 0) <= key_rng && !key_none) ? s[r] : -INFINITY;
            }
            mfma_guard(s, dp);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float d0 = dp[4 * g], d1 = dp[4 * g + 1], d2 = dp[4 * g + 2], d3 = dp[4 * g + 3];
                exp4_submul(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3], c, nl, nd, d0, d1, d2, d3);
                dp[4 * g] = d0; dp[4 * g + 1] = d1; dp[4 * g + 2] = d2; dp[4 * g + 3] = d3;
            }
            bf16x8 d0 = pack8(dp, 0), d1 = pack8(dp, 1);
#ifdef SCONF_ATTN_STAMP
            asm volatile("" : "+v"(d0)); asm volatile("" : "+v"(d1));
#endif
            STAMP(2);
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sK, L, kt * 32, db), d0, dqt[db], 0, 0, 0);
                dqt[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr<D>(sK, L, kt * 32 + 16, db), d1, dqt[db], 0, 0, 0);
            }
            STAMP(3);
        }
        dma_wait_all();
        __syncthreads();
        STAMP(4);
    }
    STAMP_OUT(5);
    if (qi < p.N) {
        if (p.rot_cos) store_t_rot<D>(dqt, p.dq + b * p.dq_sb + (long)qi * p.dq_sn + h * p.dq_sh, p.scale, hh, p.rot_cos + (long)qi * (D / 2), p.rot_sin + (long)qi * (D / 2));
        else store_t<D>(dqt, p.dq + b * p.dq_sb + (long)qi * p.dq_sn + h * p.dq_sh, p.scale, hh);
    }
}

#ifdef SCONF_ATTN_STAMP
static unsigned long long* stamp_buf(long nwg, hipStream_t stream) {
    static unsigned long long* sbuf = nullptr; static long scap = 0;
    if (scap < nwg * 64) { if (sbuf) (void)hipFree(sbuf); (void)hipMalloc(&sbuf, nwg * 64 * 8); scap = nwg * 64; }
    (void)hipMemsetAsync(sbuf, 0, nwg * 64 * 8, stream);
    return sbuf;
}
static void stamp_report(const char* what, unsigned long long* sbuf, long nwg, double units_per_wave, int nseg, const char* const* nm, hipStream_t stream) {
    if (!getenv("SCONF_ATTN_STAMP_PRINT")) return;
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned long long> hb(nwg * 64);
    (void)hipMemcpy(hb.data(), sbuf, nwg * 64 * 8, hipMemcpyDeviceToHost);
    double sums[2][8] = {};
    for (long w = 0; w < nwg * 8; ++w) for (int i = 0; i < 8; ++i) sums[(w & 7) >= 4][i] += (double)hb[w * 8 + i];
    const double n = (double)nwg * 4 * units_per_wave;
    for (int g = 0; g < 2; ++g) {
        double tot = 0; for (int i = 0; i < nseg; ++i) tot += sums[g][i];
        fprintf(stderr, "[%s stamps] waves %d-%d: cycles per unit %.0f:", what, 4 * g, 4 * g + 3, tot / n);
        for (int i = 0; i < nseg; ++i) fprintf(stderr, "  %s %.0f (%.1f%%)", nm[i], sums[g][i] / n, 100.0 * sums[g][i] / tot);
        fprintf(stderr, "\n");
    }
}
#endif

void set_lds_attrs() {
    static bool done = false;
    if (done) return;
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 256);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 256);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * 32 * 256 + 256) + 128 * 256);
    done = true;
}

int check_common(const char* fn, int64_t B, int64_t N, int64_t H, int64_t D, const int64_t* strides, int nstr) {
    if (!(D == 32 || D == 128)) return sconf_set_error("%s: head_dim %ld not supported (32 or 128)", fn, (long)D);
    if (B <= 0 || N <= 0 || H <= 0) return sconf_set_error("%s: empty problem", fn);
    if (B > 65535 || H > 65535) return sconf_set_error("%s: B and H must be <= 65535", fn);
    for (int i = 0; i < nstr; ++i) if (strides[i] % 8 != 0) return sconf_set_error("%s: strides must be multiples of 8 elements", fn);
    return 0;
}

}  // namespace

// q,k,v,o: bf16 (B,N,H,D) strided views (stride_b, stride_n, stride_h in elements; d contiguous).
// lengths: int32 [B] or null — keys >= length are masked and query rows >= length are written as zeros
// (attention.py:546-547).  lse: f32 (B,H,N) or null.  window -1 = unbounded.
SCONF_API int sconf_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* lengths,
                             int64_t B, int64_t N, int64_t H, int64_t D, const int64_t* q_strides, const int64_t* k_strides,
                             const int64_t* v_strides, const int64_t* o_strides, int win_left, int win_right, float scale,
                             hipStream_t stream) {
    int64_t all[12];
    for (int i = 0; i < 3; ++i) { all[i] = q_strides[i]; all[3 + i] = k_strides[i]; all[6 + i] = v_strides[i]; all[9 + i] = o_strides[i]; }
    if (check_common("sconf_attn_fwd", B, N, H, D, all, 12)) return 1;
    AttnParams p = {};
    p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)o; p.lse = lse; p.lengths = lengths;
    p.q_sb = q_strides[0]; p.q_sn = q_strides[1]; p.q_sh = q_strides[2];
    p.k_sb = k_strides[0]; p.k_sn = k_strides[1]; p.k_sh = k_strides[2];
    p.v_sb = v_strides[0]; p.v_sn = v_strides[1]; p.v_sh = v_strides[2];
    p.o_sb = o_strides[0]; p.o_sn = o_strides[1]; p.o_sh = o_strides[2];
    p.B = (int)B; p.N = (int)N; p.H = (int)H; p.win_left = win_left; p.win_right = win_right; p.scale = scale;
    dim3 grid((unsigned)(cdiv(N, 128) * H * B)), block(256);
    { const char* ex = getenv("SCONF_ATTN_XCD"); p.xcd_remap = !(ex && ex[0] == '0'); }       // A/B switch, read per call
    set_lds_attrs();
    const char* e8 = getenv("SCONF_ATTN_WIDE");            // "0" keeps the 4-wave kernels (A/B, tests); read per call
    // the 8-wave kernels address a (batch, head) slice through a buffer descriptor: 32-bit byte offsets
    const bool fits32 = (N - 1) * std::max(std::max(k_strides[1], v_strides[1]), q_strides[1]) * 2 + 2 * D < (1L << 32);
    const bool wide = !(e8 && e8[0] == '0') && N >= 256 && fits32;
    if (D == 128 && wide) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_fwd8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 256);
            (void)hipFuncSetAttribute((const void*)attn_fwd8p_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 256);
            attr_set = true;
        }
        const char* erc = getenv("SCONF_ATTN_RC");         // "0": the round-2 kernel (running maximum per tile); A/B and tests, read per call
#ifdef SCONF_ATTN_STAMP
        const long nwg = cdiv(N, 256) * H * B;
        static unsigned long long* sbuf = nullptr; static long scap = 0;
        if (scap < nwg * 64) { if (sbuf) (void)hipFree(sbuf); (void)hipMalloc(&sbuf, nwg * 64 * 8); scap = nwg * 64; }
        (void)hipMemsetAsync(sbuf, 0, nwg * 64 * 8, stream);
        p.stamps = sbuf;
#endif
        if (erc && erc[0] == '0') hipLaunchKernelGGL((attn_fwd8_kernel<128>), dim3((unsigned)(cdiv(N, 256) * H * B)), dim3(512), 4 * 128 * 256, stream, p);
        else hipLaunchKernelGGL((attn_fwd8p_kernel<128>), dim3((unsigned)(cdiv(N, 256) * H * B)), dim3(512), 4 * 128 * 256, stream, p);
#ifdef SCONF_ATTN_STAMP
        if (getenv("SCONF_ATTN_STAMP_PRINT")) {
            (void)hipStreamSynchronize(stream);
            std::vector<unsigned long long> hb(nwg * 64);
            (void)hipMemcpy(hb.data(), sbuf, nwg * 64 * 8, hipMemcpyDeviceToHost);
            double sums[2][8] = {};
            for (long w = 0; w < nwg * 8; ++w) for (int i = 0; i < 8; ++i) sums[(w & 7) >= 4][i] += (double)hb[w * 8 + i];
            const double ntile = (double)nwg * 4 * ((N + 63) / 64);          // (wave, half-tile) pairs per wave group
            static const char* nm[6] = {"dma-issue", "S0", "S1|exp0", "PV0|exp1", "PV1", "wait+barrier"};
            for (int g = 0; g < 2; ++g) {
                double tot = 0; for (int i = 0; i < 6; ++i) tot += sums[g][i];
                fprintf(stderr, "[attn stamps] waves %d-%d: cycles per half-tile %.0f:", 4 * g, 4 * g + 3, tot / ntile);
                for (int i = 0; i < 6; ++i) fprintf(stderr, "  %s %.0f (%.1f%%)", nm[i], sums[g][i] / ntile, 100.0 * sums[g][i] / tot);
                fprintf(stderr, "\n");
            }
        }
#endif
    } else if (D == 128) hipLaunchKernelGGL((attn_fwd_kernel<128>), grid, block, 4 * 64 * 256, stream, p);
    else          hipLaunchKernelGGL((attn_fwd_kernel<32>), grid, block, 4 * 64 * 64, stream, p);
    SCONF_LAUNCH_OK("sconf_attn_fwd");
    return 0;
}

// Backward.  delta: f32 (2,B,H,N) scratch (the dQ kernel, which runs first, leaves -rowsum(dO * O) and -lse log2(e) there for the dK/dV kernel).  dq/dk/dv: bf16 strided like q/k/v.  rot_cos / rot_sin (nullable, f32 (N, D/2)): q and k
// are rotary-rotated activations; dq and dk are then returned with the transpose of the rotation applied, i.e. as gradients of
// the unrotated projections (apply_rotary_pos_emb backward, rotary_emb.py:61-73).
SCONF_API int sconf_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                             float* delta, void* dq, void* dk, void* dv, const int32_t* lengths,
                             int64_t B, int64_t N, int64_t H, int64_t D, const int64_t* q_strides, const int64_t* k_strides,
                             const int64_t* v_strides, const int64_t* o_strides, const int64_t* do_strides,
                             const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides,
                             int win_left, int win_right, float scale, const float* rot_cos, const float* rot_sin, hipStream_t stream) {
    int64_t all[24];
    const int64_t* ss[8] = {q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides, dk_strides, dv_strides};
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 3; ++i) all[3 * j + i] = ss[j][i];
    if (check_common("sconf_attn_bwd", B, N, H, D, all, 24)) return 1;
    AttnParams p = {};
    p.q = (const bf16*)q; p.k = (const bf16*)k; p.v = (const bf16*)v; p.o = (bf16*)const_cast<void*>(o); p.dout = (const bf16*)dout;
    p.lse = const_cast<float*>(lse); p.delta = delta; p.dq = (bf16*)dq; p.dk = (bf16*)dk; p.dv = (bf16*)dv; p.lengths = lengths;
    p.q_sb = q_strides[0]; p.q_sn = q_strides[1]; p.q_sh = q_strides[2];
    p.k_sb = k_strides[0]; p.k_sn = k_strides[1]; p.k_sh = k_strides[2];
    p.v_sb = v_strides[0]; p.v_sn = v_strides[1]; p.v_sh = v_strides[2];
    p.o_sb = o_strides[0]; p.o_sn = o_strides[1]; p.o_sh = o_strides[2];
    p.do_sb = do_strides[0]; p.do_sn = do_strides[1]; p.do_sh = do_strides[2];
    p.dq_sb = dq_strides[0]; p.dq_sn = dq_strides[1]; p.dq_sh = dq_strides[2];
    p.dk_sb = dk_strides[0]; p.dk_sn = dk_strides[1]; p.dk_sh = dk_strides[2];
    p.dv_sb = dv_strides[0]; p.dv_sn = dv_strides[1]; p.dv_sh = dv_strides[2];
    p.B = (int)B; p.N = (int)N; p.H = (int)H; p.win_left = win_left; p.win_right = win_right; p.scale = scale;
    SCONF_REQUIRE((rot_cos == nullptr) == (rot_sin == nullptr), "sconf_attn_bwd: rot_cos and rot_sin go together");
    p.rot_cos = rot_cos; p.rot_sin = rot_sin;
    set_lds_attrs();
    dim3 grid((unsigned)(cdiv(N, 128) * H * B)), block(256);
    { const char* ex = getenv("SCONF_ATTN_XCD"); p.xcd_remap = !(ex && ex[0] == '0'); }       // A/B switch, read per call
    // dQ first: it computes delta = rowsum(dO * O) from fragments it holds anyway and writes it for the dK/dV kernel
    if (D == 128) {
        const char* eq = getenv("SCONF_ATTN_WIDE");
        int64_t smax = 0;
        for (int j = 0; j < 8; ++j) smax = std::max(smax, ss[j][1]);
        const bool fits32 = (N - 1) * smax * 2 + 2 * D < (1L << 32);      // 8-wave kernels: 32-bit byte offsets inside a (batch, head) slice
        if (!(eq && eq[0] == '0') && N >= 256 && fits32) {
            static bool attr_set = false;
            if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_bwd_dq8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 256); attr_set = true; }
#ifdef SCONF_ATTN_STAMP
            p.stamps = stamp_buf(cdiv(N, 256) * H * B, stream);
#endif
            hipLaunchKernelGGL((attn_bwd_dq8_kernel<128>), dim3((unsigned)(cdiv(N, 256) * H * B)), dim3(512), 4 * 128 * 256, stream, p);
#ifdef SCONF_ATTN_STAMP
            { static const char* nm[5] = {"dma-issue", "S|dP chains", "exp*", "dQ chain", "wait+barrier"}; stamp_report("dq8 (unit = 32 keys)", p.stamps, cdiv(N, 256) * H * B, (double)((N + 31) / 32), 5, nm, stream); }
#endif
        } else
            hipLaunchKernelGGL((attn_bwd_dq_kernel<128>), grid, block, 4 * 64 * 256, stream, p);
        const char* e8 = getenv("SCONF_ATTN_DKDV8");           // "0" keeps the 4-wave dK/dV kernel (A/B, tests); read per call
        const bool wide = !(e8 && e8[0] == '0');
        if (wide && N >= 256 && fits32) {
            static bool attr_set = false;
            const int sh8 = 2 * (2 * 64 * 256 + 512) + 256 * 256;
            if (!attr_set) { (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv8_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, sh8); attr_set = true; }
#ifdef SCONF_ATTN_STAMP
            p.stamps = stamp_buf(cdiv(N, 256) * H * B, stream);
#endif
            hipLaunchKernelGGL((attn_bwd_dkdv8_kernel<128>), dim3((unsigned)(cdiv(N, 256) * H * B)), dim3(512), sh8, stream, p);
#ifdef SCONF_ATTN_STAMP
            { static const char* nm[5] = {"dma-issue", "S|dP chains", "exp*", "dV|dK chains", "wait+barrier"}; stamp_report("dkdv8 (unit = 32 queries)", p.stamps, cdiv(N, 256) * H * B, (double)((N + 31) / 32), 5, nm, stream); }
#endif
        } else
            hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128>), grid, block, 2 * (2 * 32 * 256 + 256) + 128 * 256, stream, p);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<32>), grid, block, 4 * 64 * 64, stream, p);
        hipLaunchKernelGGL((attn_bwd_dkdv_kernel<32>), grid, block, 2 * (2 * 32 * 64 + 256) + 128 * 64, stream, p);
    }
    SCONF_LAUNCH_OK("sconf_attn_bwd");
    return 0;
}
