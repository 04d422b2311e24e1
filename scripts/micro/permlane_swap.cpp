// what v_permlane32_swap_b32 does on gfx950, and the mid-batch kernel's accumulator -> (row, feature) hand-over built on it:
// tile values encode (tile, unit, row); after the swaps lane (n, hf) must hold units 0..31 of tile A (hf = 0) or B (hf = 1) of row n
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int N> struct ic { static constexpr int value = N; };
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) { if constexpr (B < E) { f(ic<B>{}); static_for<B + 1, E>(f); } }
__global__ void k(float* out) {
    const int lane = threadIdx.x, n = lane & 31, hf = lane >> 5;
    f32x16 tA, tB;
    for (int j = 0; j < 16; ++j) {
        const int u = 8 * (j >> 2) + 4 * hf + (j & 3);
        tA[j] = 1000.f + 32.f * u + n;          // tile A, unit u, row n
        tB[j] = 5000.f + 32.f * u + n;
    }
    float p[32];
    static_for<0, 16>([&](auto jj) {
        constexpr int j = decltype(jj)::value, u0 = 8 * (j >> 2) + (j & 3);
        const float ea = tA[j], eb = tB[j];
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ea), __builtin_bit_cast(unsigned, eb), false, false);
        const unsigned s0 = sw[0], s1 = sw[1];      // (__builtin_bit_cast applied to a vector ELEMENT reads element 0: clang 19)
        p[u0] = __builtin_bit_cast(float, s0);
        p[u0 + 4] = __builtin_bit_cast(float, s1);
    });
    for (int u = 0; u < 32; ++u) out[lane * 32 + u] = p[u];
}
int main() {
    float* d; hipMalloc(&d, 64 * 32 * 4);
    k<<<1, 64>>>(d);
    static float h[64 * 32]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int u = 0; u < 32; ++u) {
            const float want = (lane >> 5 ? 5000.f : 1000.f) + 32.f * u + (lane & 31);
            if (h[lane * 32 + u] != want) { if (bad < 8) printf("lane %d unit %d: got %.0f want %.0f\n", lane, u, h[lane * 32 + u], want); ++bad; }
        }
    printf("%d mismatches\n", bad);
    return bad != 0;
}
