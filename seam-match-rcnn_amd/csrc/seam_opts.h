// seam_opts.h -- variant selectors of the launchers.
//
// A launcher never reads the environment: the kernel-variant switches the parity tests and the A/B tools flip (which Winograd
// form, persistent or one tile per block, tile shape of the implicit GEMM, ...) are plain ints behind the C ABI
// (seam_set_option / seam_get_option, include/seam_hip.h).  The host side (seam-match-rcnn_amd/_native.py) applies the
// SEAM_* environment variables of the same names ONCE, when it loads the library.  Reading an option is one relaxed atomic load.
#pragma once
#include <atomic>

namespace seam_opt {

enum Id {
    W24_PC,            // 1 (default): NT = 2 launches of the F(2x4) Winograd conv run on conv3x3_wino24pc; 0: conv3x3_wino24<2>
    W24_NT,            // 0 (default): the launcher picks the n-tiles per block; 1 | 2 force a form
    W24_PERSIST,       // 1 (default): conv3x3_wino24pc as one persistent block per CU; 0: one tile per block
    W24_NSPLIT,        // 0 (default): n-tile split over XCD groups chosen by the launcher (1 today); > 0 forces it
    W24_DYNLDS,        // extra dynamic LDS of conv3x3_wino24<NT> (occupancy experiments), default 0
    F16PC_TILE16,      // 1 (default): conv3x3_f16pc tiles a large map in 16 x 16 patches where those leave fewer empty slots than
                       //   8 x 32 ones; 0: always 8 x 32 (the round-5 form; same results)
    CONV_TILE,         // 0 (default): the implicit GEMM picks its block tile; BM * 1000 + BN forces one
    F16_VEC_EPILOGUE,  // default 1
    EPI_PRIO,          // default 1
    CONV_SLOTS,        // resident 4-wave blocks the persistent implicit GEMM launches, default 512
    CONV_DYNLDS,       // default 0
    PW_BLOCKS,         // blocks of conv1x1_sw, default 256 (a multiple of 64)
    WINO_MT,           // 0 (default): F(2x2) launcher picks its m-tile form; 1 | 2 force it
    WINO_NSPLIT,       // 0 (default): the F(2x2) launcher splits the n-tiles over XCD groups by its weight-footprint rule; > 0 forces it
    ROIALIGN_LDS,      // 2 (default): LDS-staged ROI quadrant tiles; 1: row-staged tiles; 0: gather kernel (profiles/r03_roialign_ab.txt)
    COUNT
};

struct Entry { const char* name; int dflt; };
constexpr Entry kTable[COUNT] = {
    {"SEAM_W24_PC", 1}, {"SEAM_W24_NT", 0}, {"SEAM_W24_PERSIST", 1}, {"SEAM_W24_NSPLIT", 0}, {"SEAM_W24_DYNLDS", 0}, {"SEAM_F16PC_TILE16", 1},
    {"SEAM_CONV_TILE", 0}, {"SEAM_F16_VEC_EPILOGUE", 1}, {"SEAM_EPI_PRIO", 1}, {"SEAM_CONV_SLOTS", 512}, {"SEAM_CONV_DYNLDS", 0},
    {"SEAM_PW_BLOCKS", 256}, {"SEAM_WINO_MT", 0}, {"SEAM_WINO_NSPLIT", 0}, {"SEAM_ROIALIGN_LDS", 2},
};

extern std::atomic<int> g_value[COUNT];      // seam_abi.hip
inline int get(Id id) { return g_value[id].load(std::memory_order_relaxed); }

}  // namespace seam_opt
