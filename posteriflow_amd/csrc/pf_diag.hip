// pf_diag.hip -- a measurement aid, not part of the path: the compute-free read that bounds the 16-row flow kernel.
//
// At 16 rows per workgroup every CU pulls the WHOLE packed weight stream of the flow through its own vector-memory path
// (256 workgroups x 7.9 MB per 4096-row launch).  This kernel reads a caller-given buffer the same way --
// one workgroup of 8 waves per CU, every wave streaming its own contiguous eighth with IN_FLIGHT one-KiB loads outstanding,
// an xor per load and nothing else -- so that bench.py can time, on the box it is running on and on the very buffer the
// flow kernel streams, the floor the kernel is measured against (`roofline.ceiling`).  scripts/micro/l2_ingest.cpp is the
// stand-alone sweep (sizes, workgroup counts, access patterns) this one pattern was chosen from.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pf_status.h"

namespace pf {

typedef __attribute__((ext_vector_type(4))) unsigned int dg_u32x4;

template <int U>
__global__ __launch_bounds__(512) void diag_stream_kernel(const dg_u32x4* __restrict__ src, int64_t n16, unsigned* __restrict__ sink) {
    dg_u32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = dg_u32x4{0u, 0u, 0u, 0u};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t per = n16 / 8;                                  // 16-byte vectors per wave region
    const dg_u32x4* reg = src + (int64_t)wave * per + lane;
    for (int64_t i = 0; i + (int64_t)(U - 1) * 64 < per; i += (int64_t)U * 64) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] ^= reg[i + (int64_t)u * 64];
    }
    unsigned r = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) r ^= acc[u][0] ^ acc[u][1] ^ acc[u][2] ^ acc[u][3];
    // the xor of a whole stream is data-dependent: the store keeps the loads alive; every workgroup writes its own word
    sink[blockIdx.x] = r;
}

int diag_stream_ingest(const void* buf, int64_t bytes, int workgroups, int in_flight, unsigned* sink, hipStream_t s) {
    const int64_t n16 = bytes / 16;
    if (n16 < 8 * 64 * 16) return PF_ERR_BAD_ARG;
    const dg_u32x4* src = reinterpret_cast<const dg_u32x4*>(buf);
    const dim3 grid((unsigned)workgroups), block(512);
    switch (in_flight) {
    case 2: hipLaunchKernelGGL(diag_stream_kernel<2>, grid, block, 0, s, src, n16, sink); break;
    case 4: hipLaunchKernelGGL(diag_stream_kernel<4>, grid, block, 0, s, src, n16, sink); break;
    case 8: hipLaunchKernelGGL(diag_stream_kernel<8>, grid, block, 0, s, src, n16, sink); break;
    case 16: hipLaunchKernelGGL(diag_stream_kernel<16>, grid, block, 0, s, src, n16, sink); break;
    default: return PF_ERR_BAD_ARG;
    }
    return launch_status();
}

}  // namespace pf
