// pf_status.h -- launch status shared by the kernel launchers and the C-ABI layer: a launcher that sees a HIP error
// keeps the code, so that pf_last_error() can name it (hipGetLastError() clears the error it returns: asking again in
// pf_api.hip used to read "no error").
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/pf_hip.h"

namespace pf {
extern thread_local int g_hip_error;       // last HIP error a launcher saw on this thread (hipError_t)
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return PF_OK;
    g_hip_error = static_cast<int>(e);
    return PF_ERR_HIP;
}
inline int hip_failed(hipError_t e) {       // for hipFuncSetAttribute and friends
    g_hip_error = static_cast<int>(e);
    return PF_ERR_HIP;
}
// opt a kernel in to more than 64 KiB of dynamic LDS; a failure is recorded for pf_last_error() (returning PF_ERR_HIP
// without it used to print whatever stale code g_hip_error held, possibly "no error")
inline bool opt_in_lds(const void* kernel, int bytes) {
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) return true;
    g_hip_error = static_cast<int>(e);
    (void)hipGetLastError();
    return false;
}
}  // namespace pf
