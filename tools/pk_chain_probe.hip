// v_pk_fma_f32 issue rate against the number of independent accumulator chains per wavefront and the number of
// wavefronts per SIMD (weights as an SGPR pair, activation broadcast by op_sel, as in pong_policy.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CHAINS, int LDSKB>
__global__ __launch_bounds__(256) void k(float* out, const f2* wsrc, int iters) {
    __shared__ float pad[LDSKB * 256];
    if (threadIdx.x == 1023) pad[0] = 1.f;
    f2 a[CHAINS];
    for (int i = 0; i < CHAINS; i++) a[i] = f2{(float)threadIdx.x, (float)i};
    f2 x = f2{1.0001f, 0.9999f};
    f2 w = wsrc[0];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / CHAINS; r++)
#pragma unroll
            for (int i = 0; i < CHAINS; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(w), "v"(x));
    }
    float s = 0;
    for (int i = 0; i < CHAINS; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; f2* w;
    (void)hipMalloc(&out, 64 << 20); (void)hipMalloc(&w, 8); (void)hipMemset(w, 0, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    auto run = [&](const char* name, auto launch, int grid) {
        launch(); (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.3f ms  %6.1f TFLOP/s\n", name, ms, (double)grid * 256 * iters * 64 * 4 / ms / 1e9);
    };
    // 256 CUs: grid 256 x 256 threads = 1 wave per SIMD (LDS 100 KB keeps a second workgroup off the CU), 512 = 2 per SIMD
#define R(C, KB, G, NAME) run(NAME, [&] { k<C, KB><<<G, 256>>>(out, w, iters); }, G)
    R(2, 100, 256, "2 chains, 1 wave/SIMD");
    R(4, 100, 256, "4 chains, 1 wave/SIMD");
    R(8, 100, 256, "8 chains, 1 wave/SIMD");
    R(2, 60, 512, "2 chains, 2 waves/SIMD");
    R(4, 60, 512, "4 chains, 2 waves/SIMD");
    R(8, 60, 512, "8 chains, 2 waves/SIMD");
    return 0;
}
