// Which block ids share a CU?  (dispatch-order probe used to design the conv tail split; GPU box only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    __shared__ char pad[73728];
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
        pad[0] = (char)hw;
    }
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 1 && pad[0] == 123) out[0] = 0;
}
int main() {
    const int nb = 528;
    unsigned* d; hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 400000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 2); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("distinct CUs used: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) { if (shown++ >= 12) break; printf("cu %llx:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    return 0;
}
