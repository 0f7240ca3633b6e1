#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const f2* wsrc, int iters) {
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x, (float)i};
    f2 x = f2{1.0001f, 0.9999f};
    f2 w = wsrc[0];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(w), "v"(x));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(x));
                if (MODE == 2) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "s"(w.x), "v"(x.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "s"(w.y), "v"(x.y));
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; f2* w;
    (void)hipMalloc(&out, 4 << 20 << 2); (void)hipMalloc(&w, 8);
    (void)hipMemset(w, 0, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, grid = 2048;
    for (int mode = 0; mode < 3; mode++)
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (mode == 0) k<0><<<grid, 512>>>(out, w, iters);
            if (mode == 1) k<1><<<grid, 512>>>(out, w, iters);
            if (mode == 2) k<2><<<grid, 512>>>(out, w, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)grid * 512 * iters * 64 * 2 * 2;
            if (rep) printf("mode %d (%s): %.3f ms  %.1f TFLOP/s\n", mode, mode == 0 ? "pk sgpr op_sel" : mode == 1 ? "pk vgpr" : "2x v_fma sgpr", ms, flop / ms / 1e9);
        }
}
