// seam_pwhpc.hip -- pointwise (1x1, stride 1) convolution in fp16 (fp32 accumulate) with a LONG reduction (C >= 512, a multiple of 256)
// as a PRODUCER / CONSUMER block, persistent over its XCD's tiles (round 6): the fp16 twin of seam_pwpc.hip, built like
// conv3x3_f16pc with one tap.
//
// Why.  conv1x1_swh (seam_pwh.hip) keeps its weight slab in LDS: C <= 512, and at C = 512 the slab is 128 channels wide, so every
// activation row is read K / 128 times by independent waves (0.52 of the HBM roof on 512 -> 256).  The implicit GEMM
// (conv_igemm<_Float16,128,128>) runs the C >= 1024 reductions of layer3 / layer4 at 0.3-0.5 of the HBM roof: its four waves stage both
// operands through LDS themselves and meet at a barrier per 64-channel chunk.  Here:
//   * a block owns 256 pixels (consecutive rows of the [M, C] activation matrix) x 128 output channels per tile;
//   * waves 4..7 (producers) copy the tile's rows, 128 channels (one 256-byte run per pixel, 16 adjacent lanes per run) per chunk,
//     global -> registers -> LDS (two LDS buffers, two more chunks in registers), across tile boundaries; their per-lane offsets are
//     launch invariants, a tile changes one descriptor; every interval issues the same 16 loads (without a chunk: out-of-range
//     offsets), so hipcc's vmcnt bookkeeping is exact;
//   * waves 0..3 (consumers, one per SIMD) only multiply: wave w owns channels 32 w .. 32 w + 31 of the tile for all 256 pixels (8
//     accumulator tiles).  Per MFMA one `ds_read_b128` (the A fragment of one 32-pixel group: lane = pixel, 16 bytes = 8 consecutive
//     k, the upper lanes 8 k further), per 8 MFMAs one 1-KiB global load (the wave's B fragment: weights packed in fragment order,
//     a 16-deep register ring = two chunks ahead); addresses are one per-lane base + immediates: no VALU in the K loop.  One barrier
//     per chunk (64 MFMAs per wave);
//   * epilogue as in conv3x3_f16pc: the consumers finish in registers (scale / shift from an LDS row, fp32 FMA, one rounding to fp16,
//     ReLU), write the fp16 tile (64 KiB, swizzled) over the LDS buffer the tile's last chunk just left, one barrier; the producers
//     drain it to memory in 16-byte pieces beside the next tile's first chunk.  No residual operand (the layers it serves -- the
//     bottleneck reductions, the top FPN lateral -- have none; the C entry refuses one).
// Arithmetic: the products of seam_conv2d_f16 (fp16 operands, fp32 accumulation in k order, fp32 scale / shift, one rounding).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr unsigned kOob = 0x80000000u;
constexpr int BM = 256;                     // pixels per tile
constexpr int CK = 128;                     // channels per chunk
constexpr int ROWB = CK * 2 + 16;           // LDS bytes per pixel and chunk: 17 slots of 16 bytes -- the 16 lanes of a ds_read_b128 cycle
                                            // (16 consecutive pixels) meet 16 different bank groups
constexpr int ABUF = BM * ROWB;             // 69632
constexpr int NP = BM * 16 / 256;           // 16-byte pieces per producer thread and chunk: 16
constexpr int EX = ABUF;                    // the finished fp16 tile (256 rows x 256 bytes) lies over chunk buffer 1
constexpr int SS = 2 * ABUF;                // scale[128] | shift[128] fp32
constexpr int DC = SS + 1024;               // the producers' drain count
constexpr int LDS_BYTES = DC + 16;
static_assert(EX + 65536 <= SS && LDS_BYTES <= 160 * 1024, "LDS map");
constexpr int RB = 16;                      // B fragments in flight per consumer wave: two chunks (16 k-steps of 16) ahead

#ifndef SEAM_PWHPC_ABL
#define SEAM_PWHPC_ABL 0     // experiments: 1 no in-loop A reads, 2 no in-loop B loads, 4 no row staging
#endif
#define LDSQ __attribute__((address_space(3)))
#define PH_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)

struct PwhpcArgs {
    const void* x;         // [M, C] fp16
    const void* w;         // packed: [K/128][4 n-tiles][C/128 chunks][8 k-steps][64 lanes][8 fp16]
    const float* scale;    // [K] or null
    const float* shift;    // [K] or null
    void* y;               // [M, K] fp16
    int M, C, K, relu;
    int tiles_n, nchunks, total_tiles;
    unsigned m_tiles_n;
};

__device__ __forceinline__ int fdivu(int a, int d, unsigned m) { return d == 1 ? a : (int)__umulhi((unsigned)a, m); }

__global__ __launch_bounds__(512, 2) void conv1x1_f16pc(const PwhpcArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int n = p.nchunks;

    // ---- the block's tiles: XCD x (= blockIdx & 7) owns a contiguous range of the launch's tiles (tile = m-tile * tiles_n + n-tile:
    // the n-tiles of a row block are neighbours -- the second one finds the rows in the L2); its blocks walk it interleaved ----
    const int T = p.total_tiles, G = gridDim.x;
    const int xcd = blockIdx.x & 7, sl0 = blockIdx.x >> 3;
    const int q8 = T >> 3, rem8 = T & 7;
    const int cnt = q8 + (xcd < rem8 ? 1 : 0);
    const int start = xcd < rem8 ? xcd * (q8 + 1) : rem8 * (q8 + 1) + (xcd - rem8) * q8;
    const int S = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
    const int ntiles = sl0 < cnt ? (cnt - sl0 + S - 1) / S : 0;
    if (ntiles == 0) return;
    const int tile0 = start + sl0;
    const size_t row_bytes = (size_t)p.C * 2, out_row = (size_t)p.K * 2;
    LDSQ unsigned* const dcnt = reinterpret_cast<LDSQ unsigned*>((LDSQ char*)smem + DC);

    if (!consumer) {
        // =================================================== producer ===================================================
        const int ptid = tid - 256;
        unsigned goff[NP];                      // byte offset of piece (ptid & 15) of tile row (ptid >> 4) + 16 r: launch invariants
        LDSQ char* lp[NP];                      // its LDS address in chunk buffer 0
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const int row = (ptid >> 4) + 16 * r;
            goff[r] = (unsigned)(row * p.C * 2 + (ptid & 15) * 16);
            lp[r] = (LDSQ char*)smem + row * ROWB + (ptid & 15) * 16;
        }
        f32x4 rq[2][NP];                        // two chunks in registers (chunk parity)
        // the REQUEST stage: chunk ck of tile `tile`; rows past M have no records in the tile's descriptor and read as zeros; past the
        // block's last tile the descriptor is empty (the loads are issued all the same: see the header)
        int tile = tile0, ck = 0, tiles_left = ntiles;
        auto x_desc = [&](const int tl) {
            const int tm = fdivu(tl, p.tiles_n, p.m_tiles_n);
            const int row0 = tm * BM;
            const int rows = tiles_left > 0 ? min(BM, p.M - row0) : 0;
            return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + (size_t)row0 * row_bytes), 0, (int)(rows * row_bytes), 0x00020000);
        };
        __amdgpu_buffer_rsrc_t rs = x_desc(tile);
        auto request = [&](f32x4 (&dst)[NP]) {
#pragma unroll
            for (int r = 0; r < NP; ++r)
                dst[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (SEAM_PWHPC_ABL & 4) ? kOob : goff[r], ck * (CK * 2), 0));
            if (++ck == n) {                    // the next request belongs to the next tile
                ck = 0;
                tile += S;
                --tiles_left;
                rs = x_desc(tile);
            }
        };
        auto store_chunk = [&](const f32x4 (&src)[NP], const int buf) {
#pragma unroll
            for (int r = 0; r < NP; ++r)
                *reinterpret_cast<f32x4 LDSQ*>(lp[r] + buf * ABUF) = src[r];
        };
        if (ptid == 0) *dcnt = 0u;
        request(rq[0]);                         // chunk 0
        request(rq[1]);                         // chunk 1
        store_chunk(rq[0], 0);
        request(rq[0]);                         // chunk 2
        PH_BAR();                               // P: chunk 0 visible
        for (int k = 0; k < ntiles; ++k) {
            for (int t = 0; t < n; t += 2) {    // two chunks per trip: the register sets' parity is a compile-time constant
                if (t == n - 2 && ptid < 64) {  // the tile's epilogue vectors -> LDS (read by the consumers behind the last chunk's barrier)
                    const int i4 = (ptid & 31) * 4;
                    const float* src = ptid < 32 ? p.scale : p.shift;
                    const float dflt = ptid < 32 ? 1.f : 0.f;
                    const int tl = tile0 + k * S;
                    const int tn = tl - fdivu(tl, p.tiles_n, p.m_tiles_n) * p.tiles_n;
                    const f32x4 vv = src ? *reinterpret_cast<const f32x4*>(src + tn * 128 + i4) : f32x4{dflt, dflt, dflt, dflt};
                    *reinterpret_cast<f32x4 LDSQ*>((LDSQ char*)smem + SS + ptid * 16) = vv;
                }
                // chunk c (even position in the tile): chunk c + 1 registers (set 1) -> buffer 1; request chunk c + 3 into set 1
                store_chunk(rq[1], 1);
                request(rq[1]);
                PH_BAR();
                store_chunk(rq[0], 0);
                request(rq[0]);
                PH_BAR();
            }
            // the tile's epilogue: the consumers' finished fp16 tile -> memory, beside the next tile's first chunk
            PH_BAR();                           // E: the tile is in LDS
            {
                const int tl = tile0 + k * S;
                const int tm = fdivu(tl, p.tiles_n, p.m_tiles_n);
                const int tn = tl - tm * p.tiles_n;
                const int row0 = tm * BM;
                const int rows = min(BM, p.M - row0);
                const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)((char*)p.y + (size_t)row0 * out_row), 0, (int)(rows * out_row), 0x00020000);
                const int piece = ptid & 15;
                int ob = ptid >> 4;
                asm volatile("" : "+v"(ob));
#pragma unroll
                for (int hf = 0; hf < 4; ++hf) {
                    u32x4 v[4];
                    SB();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int o = (hf * 4 + i) * 16 + ob;
                        v[i] = *reinterpret_cast<const u32x4 LDSQ*>((LDSQ char*)smem + EX + o * 256 + ((piece ^ (o & 15)) << 4));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int it = hf * 4 + i;
                        const int o = it * 16 + ob;
                        // (rows past M: outside the descriptor, dropped)
                        const unsigned off = (unsigned)(o * p.K + tn * 128 + piece * 8) * 2u;
                        // rows with bit 4 set keep their two 8-byte halves swapped (the consumers' conflict-free write pattern)
                        const u32x4 w = (it & 1) ? u32x4{v[i][2], v[i][3], v[i][0], v[i][1]} : v[i];
                        __builtin_amdgcn_raw_buffer_store_b128(w, y_rsrc, off, 0, 0);
                    }
                    SB();
                }
            }
            // D: every producer wave is through with the tile's rows before any of them puts the next tile's chunk 1 over them (a count
            // in LDS among the four producer waves; the consumers are in the middle of their chunk 0 and take no part)
            if (k + 1 < ntiles) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(dcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned want = 4u * (unsigned)(k + 1);
                while (__hip_atomic_load(dcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(2);
                asm volatile("" ::: "memory");
            }
        }
    } else {
        // =================================================== consumer ===================================================
        const int wn = wave;                    // this wave's 32-channel n-tile of the block's 128
        f32x16 acc[8];
        // the lane's pixel (group 0) at k-half lane >> 5 in chunk buffer 0; groups and k-steps are immediates
        LDSQ char* const ab0 = (LDSQ char*)smem + (lane & 31) * ROWB + (lane >> 5) * 16;
        const int wtile_bytes = n * 8 * 1024;   // one n-tile's weights: n chunks x 8 k-steps x 1 KiB
        const int blane = lane * 16;
        f32x4 af[8];
        f32x4 bf[RB];
        int tile = tile0;
        PH_BAR();                               // P
        __amdgpu_buffer_rsrc_t w_rsrc;
        auto load_b = [&](const int slot, const int step) {         // step = chunk * 8 + ks of the tile
            bf[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, blane, step * 1024, 0));
        };
        auto ring_preload = [&](const int tl) {
            const int tn = tl - fdivu(tl, p.tiles_n, p.m_tiles_n) * p.tiles_n;
            w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w + (size_t)(tn * 4 + wn) * wtile_bytes), 0, wtile_bytes, 0x00020000);
#pragma unroll
            for (int s = 0; s < RB; ++s) { SB(); load_b(s, s); }
            SB();
        };
        ring_preload(tile);
        for (int k = 0; k < ntiles; ++k) {
            auto chunk = [&](const int t, const int par) {       // chunk t of the tile, in buffer par (= t & 1)
                LDSQ char* const ac = ab0 + par * ABUF;
                auto read_a = [&](const int m, const int ks) -> f32x4 {
                    return *reinterpret_cast<const f32x4 LDSQ*>(ac + m * (32 * ROWB) + ks * 32);
                };
#pragma unroll
                for (int m = 0; m < 8; ++m) af[m] = read_a(m, 0);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int slot = (par * 8 + ks) % RB;
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        SB();
                        if (par == 0 && ks == 0 && t == 0) {     // the tile's first step multiplies into a constant zero: no accumulator clears
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[slot]), __builtin_bit_cast(f16x8, af[m]), z, 0, 0, 0);
                        } else {
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[slot]), __builtin_bit_cast(f16x8, af[m]), acc[m], 0, 0, 0);
                        }
                        SB();
#if !(SEAM_PWHPC_ABL & 1)
                        if (ks + 1 < 8) af[m] = read_a(m, ks + 1);
#endif
                    }
                    SB();
#if !(SEAM_PWHPC_ABL & 2)
                    load_b(slot, min(t * 8 + ks + RB, n * 8 - 1));           // (past the tile's end: its last step again, never used)
#endif
                }
                SB();
                PH_BAR();                       // chunk t + 1 is in the other buffer; this one may be overwritten
            };
            for (int t = 0; t < n; t += 2) {    // (n is even: the buffer and ring phases are compile-time constants)
                chunk(t, 0);
                chunk(t + 1, 1);
            }
            // ---- epilogue ----
            tile += S;
            if (k + 1 < ntiles) ring_preload(tile);     // in flight under the finishing arithmetic
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            int le = lane;
            asm volatile("" : "+v"(le));
            auto finish_tile = [&](auto relu_c) {
                constexpr bool RELU = decltype(relu_c)::value;
                const f16x2 lo = f16x2{(_Float16)0.f, (_Float16)0.f};
                const int h = le >> 5, ol = le & 31;
                // row o = 32 m + ol: 16 pieces of 16 bytes, piece index XOR (o & 15); rows with bit 4 set swap the 8-byte halves of a piece
                LDSQ char* const row0 = (LDSQ char*)smem + EX + ol * 256 + (((h ^ (ol >> 4)) & 1) << 3);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {    // this lane's 16 channels: 8 qd + 4 (lane >> 5) + 0..3 of the wave's 32
                    const f32x4 sc = *reinterpret_cast<const f32x4 LDSQ*>((LDSQ char*)smem + SS + (wn * 32 + 8 * qd + 4 * h) * 4);
                    const f32x4 sh = *reinterpret_cast<const f32x4 LDSQ*>((LDSQ char*)smem + SS + 512 + (wn * 32 + 8 * qd + 4 * h) * 4);
                    const int pc = ((wn * 4 + qd) ^ (ol & 15)) << 4;
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const f32x2 a0 = f32x2{acc[m][4 * qd], acc[m][4 * qd + 1]} * f32x2{sc[0], sc[1]} + f32x2{sh[0], sh[1]};
                        const f32x2 a1 = f32x2{acc[m][4 * qd + 2], acc[m][4 * qd + 3]} * f32x2{sc[2], sc[3]} + f32x2{sh[2], sh[3]};
                        f16x2 h0 = __builtin_convertvector(a0, f16x2), h1 = __builtin_convertvector(a1, f16x2);
                        if (RELU) {
                            h0 = __builtin_elementwise_max(h0, lo);
                            h1 = __builtin_elementwise_max(h1, lo);
                        }
                        u32x2 pk;
                        pk[0] = __builtin_bit_cast(unsigned, h0);
                        pk[1] = __builtin_bit_cast(unsigned, h1);
                        *reinterpret_cast<u32x2 LDSQ*>(row0 + m * (32 * 256) + pc) = pk;
                    }
                }
            };
            if (p.relu) finish_tile(std::true_type{}); else finish_tile(std::false_type{});
            PH_BAR();                           // E: the producers take it from here
        }
    }
}

// [K, C] fp32 (row-major) -> fp16 fragments [K/128][4][C/128][8][64][8]:
//   element (tn, w, chunk, ks, lane, e) = W[n = 128 tn + 32 w + (lane & 31)][c = 128 chunk + 16 ks + 8 (lane >> 5) + e]
__global__ void pwhpc_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int K, int C) {
    const int nch = C / 128;
    const size_t total = (size_t)(K / 32) * nch * 8 * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        size_t rest = i >> 6;
        const int ks = (int)(rest & 7); rest >>= 3;
        const int chunk = (int)(rest % nch);
        const int nt32 = (int)(rest / nch);
        const int nn = nt32 * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = chunk * 128 + ks * 16 + (lane >> 5) * 8 + e;
            out[i * 8 + e] = (_Float16)w[(size_t)nn * C + c];
        }
    }
}

inline unsigned magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

int pwhpc_plan(PwhpcArgs& a, long long M, int C, int K) {
    if (M <= 0 || C < 512 || (C % 256) || K < 128 || (K % 128)) return 1;      // an even number of 128-channel chunks
    if (M * (long long)(C > K ? C : K) * 2 >= (1LL << 40)) return 1;
    a.M = (int)M; a.C = C; a.K = K;
    if (M >= (1LL << 31) || (size_t)BM * C * 2 >= kOob || (size_t)BM * K * 2 >= kOob) return 1;
    a.tiles_n = K / 128;
    a.nchunks = C / CK;
    const long long tiles = ((M + BM - 1) / BM) * a.tiles_n;
    if (tiles >= (1LL << 24)) return 1;
    a.total_tiles = (int)tiles;
    a.m_tiles_n = magic(a.tiles_n);
    return 0;
}

}  // namespace

extern "C" {

/* 1 when seam_conv1x1_f16pc takes this layer: C >= 512 and a multiple of 256, K a multiple of 128 */
int seam_conv1x1_f16pc_supported(long long M, int C, int K) {
    PwhpcArgs a;
    return pwhpc_plan(a, M, C, K) == 0 ? 1 : 0;
}

long long seam_conv1x1_f16pc_weight_halves(int K, int C) { return (long long)K * C; }

/* w: [K, C] fp32 row-major (a 1x1 OIHW weight) -> the kernel's fragment order, fp16 */
int seam_pack_conv1x1_weight_f16pc(const float* w, void* w_packed, int K, int C, void* stream) {
    if (K % 128 || C % 128) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(K / 32) * (C / 128) * 8 * 64;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pwhpc_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)w_packed, K, C);
    return (int)hipGetLastError();
}

/* y[M, K] = act(scale * (x[M, C] . w^T) + shift) in fp16 with fp32 accumulation; residual must be null */
int seam_conv1x1_f16pc(const void* x, const void* w_packed, const float* scale, const float* shift, const void* residual, void* y,
                       long long M, int C, int K, int relu, void* stream) {
    PwhpcArgs a;
    if (residual || pwhpc_plan(a, M, C, K)) return (int)hipErrorInvalidValue;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.y = y; a.relu = relu;
    static std::atomic<unsigned> attr_done{0};
    static std::atomic<int> cus[32];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv1x1_f16pc, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        cus[dev & 31].store(ncu, std::memory_order_relaxed);
        attr_done.fetch_or(bit, std::memory_order_release);
    }
    const int ncu = cus[dev & 31].load(std::memory_order_relaxed);
    const unsigned grid = (unsigned)(a.total_tiles > ncu ? ncu : a.total_tiles);
    hipLaunchKernelGGL(conv1x1_f16pc, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

}  // extern "C"
