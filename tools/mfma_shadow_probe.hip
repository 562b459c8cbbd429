// mfma_shadow_probe.hip -- what can issue in the shadow of v_mfma_f32_32x32x2_f32 on gfx950?
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mfma_shadow_probe.hip -o /tmp/probe && /tmp/probe
//
// Part 1 (one stream): every wave runs  [MFMA ; NF fillers of type FT] x N  with 4 rotating accumulators, one or two waves per
//   SIMD.  Prints shader cycles per MFMA (s_memtime) and the effective clock (s_memtime / wall_clock64 at 100 MHz).
//   64 cycles per MFMA = the matrix pipe never idles.
// Part 2 (two roles): waves 0-3 of a 512-thread block stream bare MFMAs, waves 4-7 (their SIMD partners) stream fillers only --
//   the K-loop-beside-an-epilogue situation of conv_igemm.  Prints the MFMA waves' cycles per MFMA and the filler waves' cycles
//   per filler, at equal priority and with the filler waves at s_setprio 3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE = 0, F_VADD = 1, F_VIADD = 2, F_PKFMA = 3, F_DSREAD = 4, F_DSWRITE32 = 5, F_DSWRITE128 = 6, F_LOAD = 7, F_STORE = 8,
       F_SNOP = 9, F_SALU = 10, F_ACCREAD = 11 };
static const char* kNames[] = {"none", "v_add_f32", "v_add_u32", "v_pk_fma_f32", "ds_read_b128", "ds_write_b32", "ds_write_b128",
                               "buffer_load_b128", "buffer_store_b128", "s_nop 0", "s_add_u32", "v_accvgpr_read"};

struct Fill {
    float v[8];
    f32x4 q[4];
    unsigned ia[4];
    unsigned lds_addr;
    unsigned goff;
    __amdgpu_buffer_rsrc_t rsrc, wrsrc;
    unsigned sa;
};

template <int FT>
__device__ __forceinline__ void filler(Fill& f, int i) {
    if constexpr (FT == F_VADD) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(f.v[i & 7]) : "v"(f.v[(i + 4) & 7]));
    } else if constexpr (FT == F_VIADD) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(f.ia[i & 3]) : "v"(f.ia[(i + 2) & 3]));
    } else if constexpr (FT == F_PKFMA) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2& a = reinterpret_cast<f32x2*>(&f.q[i & 3])[0];
        const f32x2 b = reinterpret_cast<f32x2*>(&f.q[(i + 1) & 3])[1];
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    } else if constexpr (FT == F_DSREAD) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.q[i & 3]) : "v"(f.lds_addr));
    } else if constexpr (FT == F_DSWRITE32) {
        asm volatile("ds_write_b32 %0, %1" ::"v"(f.lds_addr), "v"(f.v[i & 7]));
    } else if constexpr (FT == F_DSWRITE128) {
        asm volatile("ds_write_b128 %0, %1" ::"v"(f.lds_addr), "v"(f.q[i & 3]));
    } else if constexpr (FT == F_LOAD) {
        f.q[i & 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f.rsrc, f.goff + (unsigned)((i & 15) * 1024), 0, 0));
    } else if constexpr (FT == F_STORE) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f.q[i & 3]), f.wrsrc, f.goff + (unsigned)((i & 15) * 1024), 0, 0);
    } else if constexpr (FT == F_SNOP) {
        asm volatile("s_nop 0");
    } else if constexpr (FT == F_SALU) {
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(f.sa));
    }
}

__device__ __forceinline__ unsigned long long memtime() { return __builtin_readcyclecounter(); }

// role: 0 = MFMA + NF fillers interleaved; 1 = MFMA only on waves 0-3, fillers only on waves 4-7 (prio = s_setprio for those)
template <int FT, int NF, int ROLE>
__global__ __launch_bounds__(512, 1) void probe(const float* __restrict__ in, float* __restrict__ scratch, float* __restrict__ out,
                                               unsigned long long* __restrict__ stamps, int iters, int prio) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int gw = blockIdx.x * (blockDim.x >> 6) + wid;
    f32x16 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        a[u] = in[(gw * 64 + lane) * 8 + u];
        b[u] = in[(gw * 64 + lane) * 8 + 4 + u];
    }
    Fill f;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.v[i] = in[tid * 8 + i] * 1e-3f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f.q[i] = *reinterpret_cast<const f32x4*>(in + tid * 8 + (i & 1) * 4);
        f.ia[i] = tid + i;
    }
    f.lds_addr = (unsigned)(wid * 4096 + lane * 16);
    f.goff = (unsigned)(lane * 16);
    f.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(scratch + (size_t)gw * 8192), 0, 32768, 0x00020000);
    f.wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(scratch + (size_t)gw * 8192 + 4096), 0, 16384, 0x00020000);
    f.sa = 0;
    *reinterpret_cast<f32x4*>(lds + f.lds_addr) = f.q[0];
    __syncthreads();

    const bool mfma_wave = ROLE == 0 || wid < 4;
    const bool fill_wave = ROLE == 0 || wid >= 4;
    if (ROLE == 1 && fill_wave && prio) __builtin_amdgcn_s_setprio(3);
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = memtime();
    if (ROLE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NF; ++i) filler<FT>(f, u * NF + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mfma_wave) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // NF = fillers per MFMA-equivalent: same count of loop trips, 4*NF fillers per trip
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4 * NF; ++i) filler<FT>(f, i);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = memtime();
    const unsigned long long w1 = wall_clock64();
    if (ROLE == 1 && prio) __builtin_amdgcn_s_setprio(0);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[u][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f.v[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += f.q[i][0] + f.q[i][1] + f.q[i][2] + f.q[i][3] + (float)f.ia[i];
    s += (float)f.sa;
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) {
        stamps[gw * 2 + 0] = t1 - t0;
        stamps[gw * 2 + 1] = w1 - w0;
    }
}

struct Ctx {
    float *in, *scratch, *out;
    unsigned long long* stamps;
    int iters;
};

template <int FT, int NF, int ROLE>
void run(const Ctx& c, int threads, int prio, const char* tag) {
    const int blocks = 256;
    const int waves = blocks * threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<FT, NF, ROLE>), dim3(blocks), dim3(threads), 100 * 1024, 0, c.in, c.scratch, c.out, c.stamps, c.iters, prio);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(waves * 2);
    hipMemcpy(st.data(), c.stamps, st.size() * 8, hipMemcpyDeviceToHost);
    const int wpb = threads / 64;
    double cyc_m = 0, cyc_f = 0, wall = 0;
    int nm = 0, nf = 0;
    for (int w = 0; w < waves; ++w) {
        const bool is_fill = ROLE == 1 && (w % wpb) >= 4;
        (is_fill ? cyc_f : cyc_m) += (double)st[w * 2];
        (is_fill ? nf : nm) += 1;
        wall += (double)st[w * 2 + 1];
    }
    const double nmf = (double)c.iters * 4;
    const double clk_ghz = (cyc_m + cyc_f) / wall * 0.1;      // wall_clock64 ticks at 100 MHz
    if (ROLE == 0) {
        const double tf = (double)waves * nmf * 4096.0 / (ms * 1e-3) / 1e12;
        printf("%-22s NF=%2d waves/SIMD=%d  cyc/MFMA %7.1f  clock %.2f GHz  kernel %8.3f ms  %6.1f TFLOP/s\n", tag, NF, threads / 256,
               cyc_m / nm / nmf, clk_ghz, ms, tf);
    } else {
        const double tf = (double)nm * nmf * 4096.0 / (ms * 1e-3) / 1e12;
        printf("%-22s NF=%2d prio=%d  MFMA waves: cyc/MFMA %7.1f | filler waves: cyc/filler %7.1f (%.0f cyc total vs %.0f)  clock %.2f GHz  %6.1f TFLOP/s\n",
               tag, NF, prio, cyc_m / nm / nmf, nf ? cyc_f / nf / (nmf * NF) : 0.0, nf ? cyc_f / nf : 0.0, cyc_m / nm, clk_ghz, tf);
    }
    fflush(stdout);
}

template <int FT>
void sweep(const Ctx& c) {
    const char* tag = kNames[FT];
    run<FT, 1, 0>(c, 256, 0, tag);
    run<FT, 2, 0>(c, 256, 0, tag);
    run<FT, 4, 0>(c, 256, 0, tag);
    run<FT, 8, 0>(c, 256, 0, tag);
    run<FT, 12, 0>(c, 256, 0, tag);
    run<FT, 16, 0>(c, 256, 0, tag);
    run<FT, 4, 0>(c, 512, 0, tag);
    run<FT, 8, 0>(c, 512, 0, tag);
    run<FT, 2, 1>(c, 512, 0, tag);
    run<FT, 2, 1>(c, 512, 1, tag);
    run<FT, 8, 1>(c, 512, 0, tag);
    run<FT, 8, 1>(c, 512, 1, tag);
}

int main(int argc, char** argv) {
    const int zero = argc > 1 && atoi(argv[1]) == 1;
    Ctx c;
    c.iters = 4000;
    const size_t nin = (size_t)256 * 512 * 8;
    std::vector<float> h(nin);
    srand(1);
    for (auto& v : h) v = zero ? 0.f : ((float)rand() / RAND_MAX * 2.f - 1.f);
    hipMalloc(&c.in, nin * 4);
    hipMemcpy(c.in, h.data(), nin * 4, hipMemcpyHostToDevice);
    hipMalloc(&c.scratch, (size_t)256 * 8 * 8192 * 4);
    hipMemset(c.scratch, 0, (size_t)256 * 8 * 8192 * 4);
    hipMalloc(&c.out, 256 * 512 * 4);
    hipMalloc(&c.stamps, 256 * 8 * 2 * 8);
    printf("data: %s\n", zero ? "zeros" : "random");
    run<F_NONE, 0, 0>(c, 256, 0, "none");
    run<F_NONE, 0, 0>(c, 512, 0, "none");
    sweep<F_VADD>(c);
    sweep<F_VIADD>(c);
    sweep<F_PKFMA>(c);
    sweep<F_DSREAD>(c);
    sweep<F_DSWRITE32>(c);
    sweep<F_DSWRITE128>(c);
    sweep<F_LOAD>(c);
    sweep<F_STORE>(c);
    sweep<F_SNOP>(c);
    sweep<F_SALU>(c);
    return 0;
}
