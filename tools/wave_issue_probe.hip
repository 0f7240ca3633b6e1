// Per-wavefront issue interval on gfx950: independent vs dependent v_add_f32, all 64 lanes vs the lower 32 active, one or two wavefronts per SIMD.
// Measured: 5.0-5.4 cycles independent, 8.5-8.7 dependent, whatever the active lanes and the partner: a wavefront issues one vector instruction per
// ~5 cycles at best, and a half-empty wave64 is not faster.
//   hipcc --offload-arch=gfx950 -O3 tools/wave_issue_probe.hip -o wave_issue_probe && ./wave_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ACTIVE>
__global__ void k(float *out, int iters) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = (float)threadIdx.x + i;
    float x = 1.0001f;
    if ((threadIdx.x & 63) < ACTIVE) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// dependent chain: one register, so the per-wave issue interval of DEPENDENT instructions shows
template <int ACTIVE>
__global__ void kd(float *out, int iters) {
    float a = (float)threadIdx.x;
    float x = 1.0001f;
    if ((threadIdx.x & 63) < ACTIVE) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 64; r++) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(x));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    float *out; (void)hipMalloc(&out, 64 << 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 4000; int cus = 256;
    for (int wps = 1; wps <= 2; wps++)
      for (int mode = 0; mode < 4; mode++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<64>, dim3(cus), dim3(256 * wps), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<32>, dim3(cus), dim3(256 * wps), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(kd<64>, dim3(cus), dim3(256 * wps), 0, 0, out, iters);
            if (mode == 3) hipLaunchKernelGGL(kd<32>, dim3(cus), dim3(256 * wps), 0, 0, out, iters);
            (void)hipEventRecord(e1), (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%d wave(s)/SIMD  %s  active lanes %d: %.2f cycles per wave-instruction per wave\n", wps, mode < 2 ? "independent" : "dependent  ", mode & 1 ? 32 : 64,
               ms * 1e-3 * 2.4e9 / ((double)iters * 64));
      }
}
