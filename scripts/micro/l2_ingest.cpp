// Micro-benchmark behind DESIGN section 4.1's "per-CU ingest" ceiling: how fast can ONE workgroup per CU pull an L2-resident
// stream that every workgroup reads (the 16-row flow kernel's access pattern: each of the 256 workgroups streams the same
// 7.9 MB of packed weights for its 16 rows)?  8 waves per workgroup, 16-byte loads, U loads in flight per wave, no compute
// beyond an xor; sizes 1 / 4 / 7.9 MB; 1, 16, 128, 256 workgroups.
// build: hipcc --offload-arch=gfx950 -O3 l2_ingest.cpp -o l2_ingest ; run: ./l2_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int U>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* out) {
    u32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = u32x4{0, 0, 0, 0};
    // a wave reads 1 KiB per instruction (the packed-fragment pattern), the 8 waves interleave
    for (size_t i = threadIdx.x; i + (size_t)(U - 1) * 512 < n16; i += (size_t)U * 512) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const u32x4 v = src[i + (size_t)u * 512]; acc[u] ^= v; }
    }
    unsigned r = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) r ^= acc[u][0] ^ acc[u][1] ^ acc[u][2] ^ acc[u][3];
    if (r == 0x12345678u) out[blockIdx.x] = r;           // never true for the test data; keeps the loads
}

// Same stream, plus a software prefetch into the XCD's L2: workgroup q of its XCD (blockIdx.x / 8 mod 32, workgroups go
// round-robin over the 8 XCDs) touches every 32nd KiB of the stream AHEAD KiB before the workgroups get there (one dword
// per 16 bytes of a lane = all 8 lines of the KiB in one wave instruction, through an LDS-DMA load: no register, no wait).
template <int U, int AHEAD>
__global__ __launch_bounds__(512) void stream_prefetch_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* out) {
    __shared__ unsigned dummy[64 * 8];
    u32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = u32x4{0, 0, 0, 0};
    const int q = (blockIdx.x >> 3) & 31, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t nkb = n16 / 64;                         // KiB in the stream
    size_t kb = wave;                                    // KiB this wave reads first (the 8 waves interleave)
    for (size_t i = threadIdx.x; i + (size_t)(U - 1) * 512 < n16; i += (size_t)U * 512) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t pk = kb + (size_t)u * 8 + AHEAD;
            if (((pk >> 3) & 31) == (size_t)q && pk < nkb)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(src + pk * 64 + lane),
                                                 (__attribute__((address_space(3))) unsigned*)(dummy + 64 * wave), 4, 0, 0);
            const u32x4 v = src[i + (size_t)u * 512]; acc[u] ^= v;
        }
        kb += (size_t)U * 8;
    }
    unsigned r = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) r ^= acc[u][0] ^ acc[u][1] ^ acc[u][2] ^ acc[u][3];
    if (r == 0x12345678u) out[blockIdx.x] = r + dummy[lane];
}

template <int U, int AHEAD>
static void run_pf(const u32x4* d, size_t bytes, int grid, unsigned* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) stream_prefetch_kernel<U, AHEAD><<<grid, 512>>>(d, bytes / 16, out);
    const int reps = 50;
    (void)hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) stream_prefetch_kernel<U, AHEAD><<<grid, 512>>>(d, bytes / 16, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("  %3d workgroups, %d in flight, prefetch %4d KiB ahead: %7.1f us  = %6.1f GB/s per CU\n", grid, U, AHEAD, us, bytes / us * 1e-3);
}

// The 16-row flow kernel's own pattern: every wave streams ITS OWN contiguous region (1/8 of the stream), U KiB in flight.
template <int U>
__global__ __launch_bounds__(512) void stream_regions_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* out) {
    u32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = u32x4{0, 0, 0, 0};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per = n16 / 8;                          // 16-byte vectors per wave region
    const u32x4* reg = src + (size_t)wave * per + lane;
    for (size_t i = 0; i + (size_t)(U - 1) * 64 < per; i += (size_t)U * 64) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const u32x4 v = reg[i + (size_t)u * 64]; acc[u] ^= v; }
    }
    unsigned r = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) r ^= acc[u][0] ^ acc[u][1] ^ acc[u][2] ^ acc[u][3];
    if (r == 0x12345678u) out[blockIdx.x] = r;
}
template <int U>
static void run_regions(const u32x4* d, size_t bytes, int grid, unsigned* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) stream_regions_kernel<U><<<grid, 512>>>(d, bytes / 16, out);
    const int reps = 50;
    (void)hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) stream_regions_kernel<U><<<grid, 512>>>(d, bytes / 16, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("  %3d workgroups, per-wave regions, %2d KiB in flight per wave: %7.1f us  = %6.1f GB/s per CU\n", grid, U, us, bytes / us * 1e-3);
}

template <int U>
static void run(const u32x4* d, size_t bytes, int grid, unsigned* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) stream_kernel<U><<<grid, 512>>>(d, bytes / 16, out);
    const int reps = 50;
    hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) stream_kernel<U><<<grid, 512>>>(d, bytes / 16, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("  %3d workgroups, %d loads in flight per wave: %7.1f us  = %6.1f GB/s per CU, %6.2f TB/s chip\n", grid, U, us,
           bytes / us * 1e-3, (double)grid * bytes / us * 1e-6);
}

int main() {
    const size_t cap = 8u << 20;
    std::vector<unsigned> h(cap / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
    u32x4* d; unsigned* out;
    hipMalloc(&d, cap); hipMalloc(&out, 4096);
    hipMemcpy(d, h.data(), cap, hipMemcpyHostToDevice);
    for (size_t bytes : {(size_t)1 << 20, (size_t)4 << 20, (size_t)7900 * 1024}) {
        printf("stream of %.1f MB read by every workgroup:\n", bytes / 1048576.0);
        for (int grid : {1, 16, 128, 256}) { run<4>(d, bytes, grid, out); run<16>(d, bytes, grid, out); }
        run<1>(d, bytes, 256, out); run<2>(d, bytes, 256, out); run<8>(d, bytes, 256, out);
        run_regions<2>(d, bytes, 256, out); run_regions<4>(d, bytes, 256, out); run_regions<8>(d, bytes, 256, out);
        run_regions<16>(d, bytes, 256, out); run_regions<16>(d, bytes, 1, out);
        run_pf<4, 64>(d, bytes, 256, out); run_pf<4, 256>(d, bytes, 256, out); run_pf<4, 1024>(d, bytes, 256, out);
        run_pf<2, 256>(d, bytes, 256, out);
    }
    return 0;
}
