// pc_probe.hip -- why does a transform wave starve beside an MFMA-streaming wave of the same SIMD?  (round 5, conv3x3_wino24pc)
// Waves 0-3 of a 512-thread block stream v_mfma_f32_32x32x2_f32 over NACC accumulator tiles (arch VGPRs, or AccVGPRs with ACCA = 1),
// optionally with the consumer's own operand traffic (one ds_read_b128 + two buffer loads per eight MFMAs); waves 4-7 loop over
// [NR ds_read_b128 -> wait -> NV v_pk_fma_f32 -> NW ds_write_b128 -> wait].  Prints cycles per MFMA and cycles per producer phase.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/pc_probe.hip -o /tmp/pc_probe && /tmp/pc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <string>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int ACCA, int OPER, int NR, int NV, int NW, int PRIO, int GAP = 0, int SWAP = 0, int PU = 1, int NL = 0>
__global__ __launch_bounds__(512, 2) void probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamps,
                                               int iters) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid0 = __builtin_amdgcn_readfirstlane(tid >> 6), wid = SWAP ? (wid0 ^ 4) : wid0;   // SWAP: the MFMA waves are the YOUNGER half (hardware waves 4-7)
    const int gw = blockIdx.x * 8 + wid;
    *reinterpret_cast<f32x4*>(lds + tid * 16) = *reinterpret_cast<const f32x4*>(in + tid * 4);
    *reinterpret_cast<f32x4*>(lds + 8192 + tid * 16) = *reinterpret_cast<const f32x4*>(in + tid * 4);
    __syncthreads();
    float s = 0.f;
    unsigned long long t0, t1;
    if (wid < 4) {
        f32x16 acc[NACC];
#pragma unroll
        for (int u = 0; u < NACC; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        f32x4 a = *reinterpret_cast<const f32x4*>(in + tid * 4), b[2];
        b[0] = a * 0.5f; b[1] = a * 0.25f;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 20, 0x00020000);
        if (PRIO & 1) __builtin_amdgcn_s_setprio(3);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < NACC / 2; ++g) {         // one "position": 8 MFMAs on two accumulator tiles
                f32x4 an = a, bn0 = b[0], bn1 = b[1];
                if (OPER) {
                    an = *reinterpret_cast<const f32x4*>(lds + lane * 16 + g * 1024);
                    bn0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * 2048, 0));
                    bn1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, g * 2048 + 1024, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if (ACCA == 2) {        // fp16 MFMA (32x32x16, fp32 accumulate): the matrix core proper, not the fp32 FMA lanes
                            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                            acc[2 * g + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b[nt]), acc[2 * g + nt], 0, 0, 0);
                        } else if (ACCA) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[2 * g + nt]) : "v"(a[kk]), "v"(b[nt][kk]));
                        else acc[2 * g + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[nt][kk], acc[2 * g + nt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        // GAP: what the MFMA wave does while its MFMA executes (the partner wave needs issue slots)
                        if (GAP >= 1 && GAP <= 4) {
#pragma unroll
                            for (int q = 0; q < GAP; ++q) asm volatile("s_nop 15");
                        }
                        if (GAP == 5) asm volatile("s_nop 7");
                        if (GAP == 6) asm volatile("s_nop 0");
                        if (GAP == 7) asm volatile("s_sleep 1");
                        if (GAP == 8) { asm volatile("s_setprio 0"); asm volatile("s_nop 15"); asm volatile("s_setprio 3"); }
                        if (GAP == 10) asm volatile("s_branch 0");
                        if (GAP == 11 && kk == 3 && nt == 1) asm volatile("s_branch 0");
                        if (GAP == 12 && nt == 1) asm volatile("s_branch 0");
                        if (GAP == 13 && kk == 3 && nt == 1) asm volatile("s_sleep 2");
                        if (GAP == 14 && kk == 3 && nt == 1) asm volatile("s_nop 11");
                        if (GAP == 9 && kk == 3 && nt == 1) { asm volatile("s_nop 15"); asm volatile("s_nop 15"); asm volatile("s_nop 15"); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                a = an; b[0] = bn0; b[1] = bn1;
            }
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int u = 0; u < NACC; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[u][r];
    } else {
        f32x4 x[12], v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const f32x4*>(in + tid * 4) * (float)(i + 1);
#pragma unroll
        for (int i = 0; i < 12; ++i) x[i] = v[i % 6];
        const f32x2 c = {0.5f, 0.25f};
        if (PRIO & 2) __builtin_amdgcn_s_setprio(3);
        t0 = __builtin_readcyclecounter();
#pragma unroll PU
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NR; ++i) x[i] = *reinterpret_cast<const f32x4*>(lds + 8192 + (wid - 4) * 1024 + lane * 16 + (i & 3) * 16 * 0 + (i >> 2) * 4096 * 0);
            if (NL) {                               // NL global requests per phase (1 KiB each, a 24 MiB window per CU walked line by line)
                const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 20, 0x00020000);
#pragma unroll
                for (int i = 0; i < NL; ++i)
                    x[i % 12] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, lane * 16, ((it * NL + i) & 1023) * 1024, 0));
            }
            if (NR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x2& d = reinterpret_cast<f32x2*>(&v[i % 6])[(i / 6) & 1];
                const f32x2 e = reinterpret_cast<f32x2*>(&x[i % 12])[(i / 12) & 1];
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(c), "v"(e));
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) *reinterpret_cast<f32x4*>(lds + 16384 + (wid - 4) * 6144 + i * 1024 + lane * 16) = v[i];
            if (NW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 6; ++i) s += v[i][0] + v[i][3];
#pragma unroll
        for (int i = 0; i < 12; ++i) s += x[i][1];
    }
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) stamps[gw] = t1 - t0;
}

static int g_iters = 2000;
template <int NACC, int ACCA, int OPER, int NR, int NV, int NW, int PRIO, int GAP = 0, int SWAP = 0, int PU = 1, int NL = 0>
void run(const float* in, float* out, unsigned long long* stamps, const char* tag) {
    const int iters = g_iters, piters = iters;
    hipFuncSetAttribute((const void*)probe<NACC, ACCA, OPER, NR, NV, NW, PRIO, GAP, SWAP, PU, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<NACC, ACCA, OPER, NR, NV, NW, PRIO, GAP, SWAP, PU, NL>), dim3(256), dim3(512), 65536, 0, in, out, stamps, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(256 * 8);
    hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
    double cm = 0, cp = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? cm : cp) += (double)st[b * 8 + w];
    cm /= 1024; cp /= 1024;
    printf("gap=%d swap=%d pu=%d nl=%d %-44s NACC=%2d acc=%s oper=%d | producer %2d rd %2d pk %d wr prio=%d | cyc/MFMA %6.1f | producer cyc/phase %7.1f  (MFMA waves ran %.0f, producers %.0f cycles)\n",
           GAP, SWAP, PU, NL, tag, NACC, ACCA == 2 ? "f16 " : ACCA ? "AGPR" : "VGPR", OPER, NR, NV, NW, PRIO, cm / (iters * NACC * 4.0), cp / piters, cm, cp);
    fflush(stdout);
}

// usage: pc_probe                 -- the table of partner-wave experiments (profiles/r05_pc_probe.txt)
//        pc_probe long [zero]     -- ~1 s of bare back-to-back fp32 MFMAs on every SIMD (random or zero operands), for a clock / power
//                                    trace sampled beside it (tools/probes/mfma_clock_trace.sh)
int main(int argc, char** argv) {
    const bool longrun = argc > 1 && std::string(argv[1]) == "long";
    const bool long16 = argc > 1 && std::string(argv[1]) == "long16";    // the same with fp16 MFMAs (32x32x16)
    const bool zero = argc > 2 && std::string(argv[2]) == "zero";
    float *in, *out; unsigned long long* stamps;
    std::vector<float> h(1 << 18);
    srand(1);
    for (auto& v : h) v = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    if (long16)                          // two proper fp16 values per 32-bit slot
        for (auto& v : h) {
            const _Float16 lo = (_Float16)v, hi = (_Float16)(zero ? 0.f : (float)rand() / RAND_MAX - 0.5f);
            unsigned short a, b; memcpy(&a, &lo, 2); memcpy(&b, &hi, 2);
            const unsigned u = a | ((unsigned)b << 16); memcpy(&v, &u, 4);
        }
    hipMalloc(&in, h.size() * 4); hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&stamps, 256 * 8 * 8);
    if (long16) {
        g_iters = 120000;
        for (int rep = 0; rep < 8; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            run<12, 2, 0, 0, 0, 0, 0, 0, 0, 1>(in, out, stamps, zero ? "bare fp16 MFMAs, zero operands" : "bare fp16 MFMAs, random operands");
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("   wall %.1f ms for 2 launches -> %.1f TFLOP/s issued\n", ms, 2.0 * 256 * 4 * (double)g_iters * 48 * 32768 / (ms * 1e-3) / 1e12);
        }
        return 0;
    }
    if (longrun) {
        g_iters = 60000;
        for (int rep = 0; rep < 8; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            run<12, 0, 0, 0, 0, 0, 0, 0, 0, 1>(in, out, stamps, zero ? "bare MFMAs, zero operands" : "bare MFMAs, random operands");
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            // two launches per run(); MFMAs per launch = 256 CUs x 4 waves x iters x 48, 4096 FLOP each
            printf("   wall %.1f ms for 2 launches -> %.1f TFLOP/s issued\n", ms, 2.0 * 256 * 4 * (double)g_iters * 48 * 4096 / (ms * 1e-3) / 1e12);
        }
        return 0;
    }
    // MFMA waves (0-3 of a 512-thread block: 12 accumulator tiles, own operand traffic) beside their SIMD partners (waves 4-7)
    run<12, 0, 1, 0, 0, 0, 0, 0, 0, 1>(in, out, stamps, "idle partners");
    run<12, 0, 1, 12, 0, 0, 0, 0, 0, 8>(in, out, stamps, "partners: 12 ds_read_b128 + wait per phase");
    run<12, 0, 1, 0, 0, 6, 0, 0, 0, 8>(in, out, stamps, "partners: 6 ds_write_b128 + wait per phase");
    run<12, 0, 1, 0, 36, 0, 0, 0, 0, 8>(in, out, stamps, "partners: 36 v_pk_fma_f32 per phase");
    run<12, 0, 1, 12, 36, 6, 0, 0, 0, 1>(in, out, stamps, "partners: a transform phase (12 rd, 36 pk, 6 wr)");
    run<12, 0, 1, 24, 72, 12, 0, 0, 0, 1>(in, out, stamps, "partners: twice that work per phase");
    run<12, 0, 1, 12, 36, 6, 2, 0, 0, 1>(in, out, stamps, "  ... partners at s_setprio 3");
    run<12, 0, 1, 12, 36, 6, 1, 0, 0, 1>(in, out, stamps, "  ... MFMA waves at s_setprio 3");
    run<12, 0, 1, 12, 36, 6, 0, 0, 1, 1>(in, out, stamps, "  ... MFMA waves are the younger half");
    run<4, 0, 1, 12, 36, 6, 0, 0, 0, 1>(in, out, stamps, "  ... 4 accumulator tiles (16-MFMA loop trips)");
    run<12, 0, 1, 12, 36, 6, 0, 5, 0, 1>(in, out, stamps, "  ... s_nop 7 (32 cycles) behind every MFMA");
    run<12, 0, 1, 12, 36, 6, 0, 1, 0, 1>(in, out, stamps, "  ... s_nop 15 (64 cycles) behind every MFMA");
    run<12, 0, 1, 12, 36, 6, 0, 10, 0, 1>(in, out, stamps, "  ... s_branch behind every MFMA");
    run<12, 0, 1, 12, 36, 6, 0, 13, 0, 1>(in, out, stamps, "  ... s_sleep 2 behind every 8th MFMA");
    run<12, 0, 1, 12, 36, 6, 0, 9, 0, 1>(in, out, stamps, "  ... 3 x s_nop 15 behind every 8th MFMA");
    // the same partners beside fp16 MFMAs (v_mfma_f32_32x32x16_f16): is their vector ALU free there?
    run<12, 2, 1, 0, 0, 0, 0, 0, 0, 1>(in, out, stamps, "fp16 MFMA waves, idle partners");
    run<12, 2, 1, 0, 36, 0, 0, 0, 0, 8>(in, out, stamps, "fp16 MFMA waves; partners: 36 v_pk_fma_f32 per phase");
    run<12, 2, 1, 12, 36, 6, 0, 0, 0, 1>(in, out, stamps, "fp16 MFMA waves; partners: a transform phase");
    run<12, 2, 1, 24, 72, 12, 0, 0, 0, 1>(in, out, stamps, "fp16 MFMA waves; partners: twice that");
    // vector-memory requests of the partners: do they slow the MFMA waves (which issue two requests per eight MFMAs themselves)?
    run<12, 0, 1, 0, 0, 0, 0, 0, 0, 8, 3>(in, out, stamps, "partners: 3 global requests per phase, nothing else");
    run<12, 0, 1, 0, 0, 0, 0, 0, 0, 8, 12>(in, out, stamps, "partners: 12 global requests per phase, nothing else");
    run<12, 0, 0, 0, 0, 0, 0, 0, 0, 8, 12>(in, out, stamps, "partners: 12 requests per phase; MFMA waves without operand traffic");
    run<12, 0, 1, 12, 0, 6, 0, 0, 0, 8, 12>(in, out, stamps, "partners: 12 requests + 12 LDS reads + 6 LDS writes per phase");
    printf("reading: 'producer cyc/phase' = (partner waves' total cycles) / 2000 phases; the partners keep running after the MFMA waves end,\n"
           "so a total only slightly above the MFMA waves' total means: (almost) nothing was done beside them.\n");
    return 0;
}
