// L2 -> CU load-path probe (kernel experiments): bytes per clock per CU that buffer_load_b128 streams deliver from an
// L2-resident (L1-thrashing) working set, for the access shapes of the conv kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_bw_probe.hip -o /tmp/l2probe && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Every step a 256-thread block re-reads its private span with NL loads per thread:
//   off(i) = lane_off + i * step_bytes,  lane_off = (tid / LPR) * row_stride + (tid % LPR) * 16
template <int NL>
__global__ __launch_bounds__(256) void probe(const char* base, size_t block_stride, int lpr, int row_stride, int step_bytes,
                                             int steps, float* sink, int phases = 1, int phase_bytes = 0) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)blockIdx.x * block_stride), 0, 0x7fffffff, 0x00020000);
    const unsigned lane_off = (threadIdx.x / lpr) * row_stride + (threadIdx.x % lpr) * 16;
    f32x4 acc = {0, 0, 0, 0};
    int ph = 0;
    for (int s = 0; s < steps; ++s) {
        f32x4 v[NL];
        const unsigned po = lane_off + ph * phase_bytes;
        if (++ph == phases) ph = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i)
            v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, po + i * step_bytes, 0, 0));
#pragma unroll
        for (int i = 0; i < NL; ++i) acc += v[i];
        asm volatile("" ::: "memory");
    }
    if (acc[0] == 123.456f) sink[0] = acc[1] + acc[2] + acc[3];
}

int main() {
    char* buf;
    float* sink;
    (void)hipMalloc(&buf, 1ull << 30);
    (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 0, 1ull << 30);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int steps = 2000;
    auto run = [&](const char* name, int blocks, int nl, int lpr, int row_stride, int step_bytes, size_t block_stride,
                   int phases = 1, int phase_bytes = 0) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (nl == 8) hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, buf, block_stride, lpr, row_stride, step_bytes, steps, sink, phases, phase_bytes);
            else if (nl == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, buf, block_stride, lpr, row_stride, step_bytes, steps, sink, phases, phase_bytes);
            else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, buf, block_stride, lpr, row_stride, step_bytes, steps, sink, phases, phase_bytes);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double bytes = (double)blocks * 256 * 16 * nl * steps;
        const double tbs = bytes / ms / 1e9;
        printf("%-62s blocks %4d span/block %4d KB : %6.2f TB/s = %5.1f B/clk/CU\n", name, blocks, phases * nl * 256 * 16 / 1024, tbs,
               tbs * 1e12 / 256 / 2.4e9);
    };
    // span per block = NL * 4 KB.  512 blocks x 32 KB = 2 MB per XCD (L2 resident), 2 blocks x 32 KB per CU (> 32 KB L1)
    run("linear (1 KB per wave-instr, 4 KB apart)", 512, 8, 64, 0, 4096, 32 << 10);
    run("gather: 128-B rows @1 KB (8 lanes/row), 8 chunks/row", 512, 8, 8, 1024, 128, 32 << 10);
    run("weights: 128-B rows @128 B, slabs 4 KB apart", 512, 8, 8, 128, 4096, 32 << 10);
    run("gather, 16 loads in flight (64 KB span)", 512, 16, 8, 2048, 128, 64 << 10);
    run("gather, 1024 blocks (4/CU)", 1024, 8, 8, 1024, 128, 32 << 10);
    run("gather, 256 blocks (1/CU)", 256, 8, 8, 1024, 128, 32 << 10);
    run("gather, all blocks the SAME 32 KB", 512, 8, 8, 1024, 128, 0);
    run("gather, span 128 KB apart (64 MB total: L2 miss, MALL hit)", 512, 8, 8, 1024, 128, 128 << 10);
    run("gather, 1 MB apart (512 MB: HBM)", 512, 8, 8, 1024, 128, 1 << 20);
    // working sets that MISS the 32 KB L1 (96-192 KB per CU) and stay in the 4 MB L2 of the XCD
    run("L1-miss/L2-hit: gather, 3 phases x 16 KB, 2 blocks/CU", 512, 4, 8, 512, 128, 48 << 10, 3, 16 << 10);
    run("L1-miss/L2-hit: linear, 3 phases x 16 KB, 2 blocks/CU", 512, 4, 64, 0, 4096, 48 << 10, 3, 16 << 10);
    run("L1-miss/L2-hit: gather, 3 phases x 32 KB, 1 block/CU", 256, 8, 8, 1024, 128, 96 << 10, 3, 32 << 10);
    run("L1-miss/L2-hit: gather 16 in flight, 2 x 64 KB, 1 block/CU", 256, 16, 8, 2048, 128, 128 << 10, 2, 64 << 10);
    run("L2-miss (24 MB/XCD), MALL: gather, 12 phases x 32 KB", 512, 8, 8, 1024, 128, 384 << 10, 12, 32 << 10);
    return 0;
}
