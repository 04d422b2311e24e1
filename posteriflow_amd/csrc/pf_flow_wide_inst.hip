// pf_flow_wide_inst.hip -- instantiates the large-batch forward kernel (pf_flow_wide_kernel.h) for ONE feature count;
// built once per shape with -DPF_WIDE_D=<D> (Makefile) so the shapes compile in parallel.
#include <cstdio>

#include "pf_flow_wide_kernel.h"

#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

namespace pf {

int PF_CAT(launch_flow_wide_d, PF_WIDE_D)(const FwdParams& p, hipStream_t s) {
    constexpr int D = PF_WIDE_D, CKS = 18;
    if (p.plan.D != D || p.plan.CKM != CKS || !p.plan.wide) return PF_ERR_UNSUPPORTED;
    auto kern = flow_wide_kernel<D, CKS>;
    constexpr int lds = wide::lds_bytes();
    if (!opt_in_lds(reinterpret_cast<const void*>(kern), lds))
        return PF_ERR_HIP;
    const unsigned grid = (unsigned)((p.batch + wide::kRowsPerWG - 1) / wide::kRowsPerWG);
#if PF_WIDE_TRACE
    // diagnostic build: per-stage s_memtime spans of wave 0 of workgroup 0, printed after a synchronous launch
    static unsigned long long* trace = nullptr;
    if (!trace && hipMalloc(&trace, 10 * sizeof(unsigned long long)) != hipSuccess) return PF_ERR_HIP;
    FwdParams q = p;
    q.fail_flags = reinterpret_cast<uint32_t*>(trace);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, q);
    unsigned long long host[10];
    if (hipMemcpy(host, trace, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return PF_ERR_HIP;
    static const char* names[10] = {"stage1", "W0", "W1+gate", "final GEMMs", "spline", "pad+bias", "DMA issue", "barrier A", "barrier B", "kernel"};
    fprintf(stderr, "[wide trace, %lld rows]", (long long)p.batch);
    for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.1f", names[i], host[i] * 1e-3);
    fprintf(stderr, " (kilo-ticks of s_memtime)\n");
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p);
#endif
    return launch_status();
}

}  // namespace pf
