// Micro-benchmark: how fast can ONE CU issue global stores, by access pattern?  (diagnostic for the GEMM epilogue)
// Each workgroup (512 threads = 8 waves, one per CU when few are launched) writes `iters` x 128 KB with dwordx4 stores.
//   pattern 0: wave instruction = 16 rows x 64 B   (the 256-row GEMM epilogue today; row stride = ld bytes)
//   pattern 1: 8 rows x 128 B      pattern 2: 4 rows x 256 B      pattern 3: 2 rows x 512 B     pattern 4: 1 KB contiguous
// build: hipcc --offload-arch=gfx950 -O3 -o store_issue store_issue.hip ; run: ./store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(char* out, long ld, int pattern, int iters, long wg_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = out + blockIdx.x * wg_stride;
    const int rows_per = 16 >> pattern, segs = 4 << pattern;            // rows per instruction, 16-B segments per row
    const int r = lane / segs, sgm = lane % segs;
    const uint4 v = make_uint4(lane, wave, 3, 4);
    for (int it = 0; it < iters; ++it) {
        // a wave owns a 16-row x 1-KB slab per iteration (16 KB): 16 instructions
        char* slab = base + ((long)it * 8 + wave) * 16 * ld;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lin = i * rows_per + r;                            // (row, chunk) unit: row = lin % 16, chunk = lin / 16
            *reinterpret_cast<uint4*>(slab + (long)(lin & 15) * ld + ((lin >> 4) * segs + sgm) * 16) = v;
        }
    }
}
int main() {
    const long ld = 6144;                                               // N = 3072 bf16 row
    const int iters = 64;
    for (int nwg : {8, 32, 256}) {
        const long wg_stride = (long)iters * 8 * 16 * ld;               // disjoint row ranges per workgroup
        char* buf; hipMalloc(&buf, wg_stride * nwg);
        for (int pattern = 0; pattern <= 4; ++pattern) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<nwg, 512>>>(buf, ld, pattern, iters, wg_stride);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int rep = 0; rep < 5; ++rep) k<<<nwg, 512>>>(buf, ld, pattern, iters, wg_stride);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double bytes = (double)nwg * iters * 8 * 16 * 1024;
            printf("wgs %3d pattern %d (%2d rows x %4d B per instr): %8.1f us  %7.1f GB/s per WG  %6.2f TB/s total  (~%.1f B/clk/CU at 2.1 GHz)\n", nwg, pattern,
                   16 >> pattern, 64 << pattern, ms * 1e3, bytes / nwg / ms / 1e6, bytes / ms / 1e9, bytes / nwg / ms / 1e6 / 2.1);
        }
        hipFree(buf);
    }
    return 0;
}
