// seam_abi.hip -- version / error helpers and the launchers' variant selectors of the C ABI (include/seam_hip.h).
#include <hip/hip_runtime.h>
#include <string.h>
#include "seam_opts.h"

namespace seam_opt {
std::atomic<int> g_value[COUNT];
namespace {
struct Init {
    Init() { for (int i = 0; i < COUNT; ++i) g_value[i].store(kTable[i].dflt, std::memory_order_relaxed); }
} g_init;
int find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < COUNT; ++i)
        if (!strcmp(name, kTable[i].name)) return i;
    return -1;
}
}  // namespace
}  // namespace seam_opt

extern "C" {

int seam_version(void) { return 1001; }   // 1.001

const char* seam_error_string(int code) { return hipGetErrorString((hipError_t)code); }

int seam_option_count(void) { return seam_opt::COUNT; }

const char* seam_option_name(int index) { return index >= 0 && index < seam_opt::COUNT ? seam_opt::kTable[index].name : nullptr; }

int seam_set_option(const char* name, int value) {
    const int i = seam_opt::find(name);
    if (i < 0) return (int)hipErrorInvalidValue;
    seam_opt::g_value[i].store(value, std::memory_order_relaxed);
    return 0;
}

int seam_get_option(const char* name, int* value) {
    const int i = seam_opt::find(name);
    if (i < 0 || !value) return (int)hipErrorInvalidValue;
    *value = seam_opt::g_value[i].load(std::memory_order_relaxed);
    return 0;
}

}  // extern "C"
