// Micro-benchmark for the wide kernel's inner loop (round 2): cycles per v_mfma_f32_32x32x16_bf16 in a dependent
// chain fed by one ds_read_b128 per MFMA, 4 waves per workgroup (one per SIMD), one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 mfma_lds.cpp -o mfma_lds ; run: ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;

template <int MODE, int P, int NT = 256>
__global__ __launch_bounds__(NT) void k(const char* g, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 72 * 1024 / 16; i += NT) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(g)[i];
    __syncthreads();
    uint32_t rl = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + lane * 16;
    bf16x8 b[16];
    for (int i = 0; i < 16; ++i) {
        u32x4 t = reinterpret_cast<const u32x4*>(g)[lane + 64 * i];
        b[i] = __builtin_bit_cast(bf16x8, t);
        if (MODE & 1) asm volatile("" : "+a"(b[i]));       // B operand in AGPRs
    }
    f32x16 acc, acc2;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    u32x4 aq[P];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(rl));
#pragma unroll
        for (int e = 0; e < P; ++e) aq[e] = *reinterpret_cast<const lds_u32x4_t*>(rl + e * 1024);
#pragma unroll
        for (int e = 0; e < 64; ++e) {
            const u32x4 a = aq[e % P];
            if (!(MODE & 2)) {
                aq[e % P] = *reinterpret_cast<const lds_u32x4_t*>(rl + ((e + P) % 72) * 1024);
                if (MODE & 16) asm volatile("" : "+a"(aq[e % P]));   // A fragments land in AGPRs
            }
            if (MODE & 4) {                                  // two independent chains, alternating
                if (e & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b[e % 16], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b[e % 16], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), b[e % 16], acc, 0, 0, 0);
            }
            if (MODE & 8) {                                  // 4 independent VALU ops per MFMA
                asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32 %3, %3, %3"
                             : "+v"(acc2[0]), "+v"(acc2[1]), "+v"(acc2[2]), "+v"(acc2[3]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int P, int NT = 256>
void run(const char* name, const char* g, float* out, unsigned long long* cyc) {
    const int iters = 200, grid = 256;
    auto kern = k<MODE, P, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 80 * 1024, 0, g, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 80 * 1024, 0, g, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double tflops = (double)grid * (NT / 64) * iters * 64.0 * 32768.0 / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    printf("%-48s %7.1f ticks per MFMA   %7.3f ms  %7.1f TFLOP/s  (%.2f GHz if tick = cycle)\n", name, s / grid / (iters * 64.0), ms, tflops,
           s / grid / (ms * 1e6));
}

int main() {
    char* g; float* out; unsigned long long* cyc;
    hipMalloc(&g, 1 << 20); hipMemset(g, 0x3c, 1 << 20);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    run<2, 4>("chain, no LDS reads", g, out, cyc);
    run<0, 4>("chain + ds_read_b128, P=4, B in VGPR", g, out, cyc);
    run<1, 4>("chain + ds_read_b128, P=4, B in AGPR", g, out, cyc);
    run<1, 8>("chain + ds_read_b128, P=8, B in AGPR", g, out, cyc);
    run<5, 4>("two chains + ds_read, P=4, B in AGPR", g, out, cyc);
    run<9, 4>("chain + ds_read + 4 VALU, P=4, B in AGPR", g, out, cyc);
    run<13, 4>("two chains + ds_read + 4 VALU, P=4", g, out, cyc);
    run<17, 4>("chain + ds_read, P=4, A and B in AGPR", g, out, cyc);
    run<25, 4>("chain + ds_read + 4 VALU, A and B in AGPR", g, out, cyc);
    run<1, 4, 512>("8 waves: chain + ds_read, B in AGPR (per wave)", g, out, cyc);
    run<9, 4, 512>("8 waves: chain + ds_read + 4 VALU (per wave)", g, out, cyc);
    run<2, 4, 512>("8 waves: chain, no LDS reads (per wave)", g, out, cyc);
    run<10, 4>("chain + 4 VALU, no LDS reads", g, out, cyc);
    return 0;
}
