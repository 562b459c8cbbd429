// seam_abi.hip -- version / error helpers of the C ABI (include/seam_hip.h).
#include <hip/hip_runtime.h>

extern "C" {

int seam_version(void) { return 1000; }   // 1.000

const char* seam_error_string(int code) { return hipGetErrorString((hipError_t)code); }

}  // extern "C"
