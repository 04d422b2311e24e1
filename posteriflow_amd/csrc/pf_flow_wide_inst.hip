// pf_flow_wide_inst.hip -- instantiates the large-batch forward kernel (pf_flow_wide_kernel.h) for ONE feature count;
// built once per shape with -DPF_WIDE_D=<D> (Makefile) so the shapes compile in parallel.
#include "pf_flow_wide_kernel.h"

#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

namespace pf {

int PF_CAT(launch_flow_wide_d, PF_WIDE_D)(const FwdParams& p, hipStream_t s) {
    constexpr int D = PF_WIDE_D, CKS = 18;
    if (p.plan.D != D || p.plan.CKM != CKS || !p.plan.wide) return PF_ERR_UNSUPPORTED;
    auto kern = flow_wide_kernel<D, CKS>;
    constexpr int lds = wide::lds_bytes();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return PF_ERR_HIP;
    const unsigned grid = (unsigned)((p.batch + wide::kRowsPerWG - 1) / wide::kRowsPerWG);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
    return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_HIP;
}

}  // namespace pf
