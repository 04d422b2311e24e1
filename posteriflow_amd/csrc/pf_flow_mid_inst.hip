// pf_flow_mid_inst.hip -- instantiates the mid-batch forward kernel (pf_flow_mid_kernel.h) for ONE feature count;
// built once per shape with -DPF_MID_D=<D> (Makefile) so the shapes compile in parallel.
#include <cstdio>

#include "pf_flow_mid_kernel.h"

#define PF_CAT2(a, b) a##b
#define PF_CAT(a, b) PF_CAT2(a, b)

namespace pf {

int PF_CAT(launch_flow_mid_d, PF_MID_D)(const FwdParams& p, hipStream_t s) {
    constexpr int D = PF_MID_D, CKS = 18;
    if (p.plan.D != D || p.plan.CKM != CKS || !p.plan.wide) return PF_ERR_UNSUPPORTED;
    auto kern = flow_mid_kernel<D, CKS>;
    constexpr int lds = mid::lds_bytes(CKS);
    if (!opt_in_lds(reinterpret_cast<const void*>(kern), lds)) return PF_ERR_HIP;
    const unsigned grid = (unsigned)((p.batch + mid::kRowsPerWG - 1) / mid::kRowsPerWG);
#if PF_MID_TRACE
    // diagnostic build: per-stage s_memtime spans of one wave of workgroup 0, printed after a synchronous launch
    static unsigned long long* trace = nullptr;
    if (!trace && hipMalloc(&trace, 8 * sizeof(unsigned long long)) != hipSuccess) return PF_ERR_HIP;
    FwdParams q = p;
    q.fail_flags = reinterpret_cast<uint32_t*>(trace);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(mid::kThreads), lds, s, q);
    unsigned long long host[8];
    if (hipMemcpy(host, trace, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return PF_ERR_HIP;
    static const char* names[8] = {"stage1", "W0", "W1+gate", "final GEMMs", "splines", "layer end", "(barriers)", "kernel"};
    fprintf(stderr, "[mid trace, %lld rows, wave %d]", (long long)p.batch, PF_MID_TRACE_WAVE);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f", names[i], host[i] * 1e-3);
    fprintf(stderr, " (kilo-ticks of s_memtime, 100 MHz)\n");
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(mid::kThreads), lds, s, p);
#endif
    return launch_status();
}

}  // namespace pf
