// xcd_probe.hip -- where do the blocks of a persistent one-block-per-CU launch land when TWO such launches are in flight at once?
// The persistent kernels of this library (conv3x3_wino24pc, conv3x3_f16pc, conv1x1_pc, ...) take `blockIdx & 7` for the XCD a block
// runs on (workgroups are handed to the eight XCDs round-robin in launch order) and give the blocks of one XCD a contiguous range of the
// launch's tiles, so that neighbouring tiles meet in that XCD's L2.  The fp16 path's "slow mode" (the same kernels on two
// streams, 205-220 ms instead of 152) asked whether that still holds when another launch's blocks reach the dispatcher in between.
// Each block records XCC_ID (s_getreg_b32 hwreg(HW_REG_XCC_ID)) and spins ~100 us holding 140 KiB of LDS (one block per CU).
//   case 1: one launch of 256 blocks alone;   case 2: two launches of 256 blocks on two streams at the same time;
//   case 3: the same with the second stream's launch 20 us late.
// build: hipcc -O2 --offload-arch=gfx950 xcd_probe.hip -o xcd_probe
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(512) void probe(unsigned* out, long long spin_cycles) {
    extern __shared__ char smem[];
    const long long t0 = __builtin_amdgcn_s_memtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        smem[0] = 1;
        out[blockIdx.x * 2] = xcc & 0xf;
        out[blockIdx.x * 2 + 1] = (unsigned)(t0 >> 6);      // arrival stamp (64-cycle units of the 100 MHz counter)
    }
    while (__builtin_amdgcn_s_memtime() - t0 < spin_cycles) __builtin_amdgcn_s_sleep(8);
}

static void report(const char* name, const std::vector<unsigned>& h, int nblk) {
    int ok = 0, hist[8][8] = {};
    for (int b = 0; b < nblk; ++b) {
        const int x = h[2 * b] & 7;
        ok += x == (b & 7);
        hist[b & 7][x]++;
    }
    printf("%-34s blocks with XCC_ID == blockIdx & 7: %3d of %d", name, ok, nblk);
    int spread = 0;
    for (int r = 0; r < 8; ++r) { int k = 0; for (int x = 0; x < 8; ++x) k += hist[r][x] > 0; if (k > spread) spread = k; }
    printf("   (the blocks of one blockIdx & 7 class sit on up to %d XCDs; class -> XCC_ID:", spread);
    for (int r = 0; r < 8; ++r) { int best = 0; for (int x = 0; x < 8; ++x) if (hist[r][x] > hist[r][best]) best = x; printf(" %d", best); }
    printf(")\n");
}

int main() {
    const int nblk = 256, lds = 140 * 1024;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned *a, *b;
    hipMalloc(&a, nblk * 8); hipMalloc(&b, nblk * 8);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<unsigned> ha(nblk * 2), hb(nblk * 2);
    const long long spin = 10000;       // s_memtime counts at 100 MHz: 100 us
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), lds, s1, a, spin);
        hipDeviceSynchronize();
        hipMemcpy(ha.data(), a, nblk * 8, hipMemcpyDeviceToHost);
        report("alone", ha, nblk);
    }
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), lds, s1, a, spin);
        hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), lds, s2, b, spin);
        hipDeviceSynchronize();
        hipMemcpy(ha.data(), a, nblk * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), b, nblk * 8, hipMemcpyDeviceToHost);
        report("two streams at once, stream 1", ha, nblk);
        report("two streams at once, stream 2", hb, nblk);
    }
    for (int rep = 0; rep < 3; ++rep) {
        // a train of launches per stream, as a step issues them: the streams drift against each other
        for (int k = 0; k < 6; ++k) {
            hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), lds, s1, a, spin + 700 * k);
            hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), lds, s2, b, spin - 900 * k);
        }
        hipDeviceSynchronize();
        hipMemcpy(ha.data(), a, nblk * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), b, nblk * 8, hipMemcpyDeviceToHost);
        report("trains of 6, last of stream 1", ha, nblk);
        report("trains of 6, last of stream 2", hb, nblk);
    }
    // smaller launches that share the chip: 128 + 128 blocks
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(128), dim3(512), lds, s1, a, spin);
        hipLaunchKernelGGL(probe, dim3(128), dim3(512), lds, s2, b, spin);
        hipDeviceSynchronize();
        hipMemcpy(ha.data(), a, 128 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hb.data(), b, 128 * 8, hipMemcpyDeviceToHost);
        report("128 + 128 blocks, stream 1", ha, 128);
        report("128 + 128 blocks, stream 2", hb, 128);
    }
    return 0;
}
