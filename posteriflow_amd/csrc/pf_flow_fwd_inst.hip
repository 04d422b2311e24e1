// pf_flow_fwd_inst.hip -- instantiates the forward kernels of ONE (precision, NT); built
// once per combination with -DPF_INST_BF16=<0|1> -DPF_INST_NT=<4|8|12|16> -DPF_INST_INV=<0|1|2> (Makefile; 0 forward,
// 1 inverse, 2 training forward with dropout), so the variants compile in parallel.
#include <algorithm>

#include "pf_flow_fwd_kernel.h"

#define PF_CAT2(a, b, c, d) a##b##c##d
#define PF_CAT(a, b, c, d) PF_CAT2(a, b, c, d)

namespace pf {

#if PF_INST_INV == 2
#define PF_NAME launch_flow_train_p
#elif PF_INST_INV
#define PF_NAME launch_flow_inverse_p
#else
#define PF_NAME launch_flow_forward_p
#endif
int PF_CAT(PF_NAME, PF_INST_BF16, _nt, PF_INST_NT)(const FwdParams& p, int R, hipStream_t s) {
    constexpr bool BF = PF_INST_BF16 != 0;
    constexpr bool INV = PF_INST_INV == 1;
    constexpr bool DROP = PF_INST_INV == 2;
    constexpr int NT = PF_INST_NT;
    constexpr int SMALL = BF ? 3 : 6, MID = BF ? 9 : 18, LARGE = BF ? 18 : 36;
    const int ckm = p.plan.CKM;
    if (ckm == 0) return launch_ckm<BF, NT, 0, INV, DROP>(p, R, s);
    if (ckm == MID) return launch_ckm<BF, NT, MID, INV, DROP>(p, R, s);
    if constexpr (NT <= 8) { if (ckm == SMALL) return launch_ckm<BF, NT, SMALL, INV, DROP>(p, R, s); }
    if constexpr (NT == 16) { if (ckm == LARGE) return launch_ckm<BF, NT, LARGE, INV, DROP>(p, R, s); }
    return PF_ERR_UNSUPPORTED;
}

}  // namespace pf
