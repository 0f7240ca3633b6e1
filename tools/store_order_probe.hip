// Why does a wavefront that issues more than one 1-KiB store run slower than the fill rate?  Pure-store variants on
// the fused-84 tensor (3.70 GB): same bytes, different order of a wavefront's stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int T>
__global__ __launch_bounds__(T) void lin1(uint4* out, int64_t total) {
    const int64_t g = (int64_t)blockIdx.x * T + threadIdx.x;
    if (g < total) out[g] = make_uint4((unsigned)g, 1, 2, 3);
}
// K stores per thread, workgroup-contiguous span (wave's stores are 4 KB apart)
template <int K>
__global__ __launch_bounds__(256) void lin_wg(uint4* out, int64_t total) {
    const int64_t g0 = (int64_t)blockIdx.x * 256 * K + threadIdx.x;
#pragma unroll
    for (int i = 0; i < K; i++) if (g0 + i * 256 < total) out[g0 + i * 256] = make_uint4(i, 1, 2, 3);
}
// K stores per thread, wave-contiguous span (wave's stores are adjacent: 1 KB apart)
template <int K>
__global__ __launch_bounds__(256) void lin_wave(uint4* out, int64_t total) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t g0 = w * 64 * K + (threadIdx.x & 63);
#pragma unroll
    for (int i = 0; i < K; i++) if (g0 + i * 64 < total) out[g0 + i * 64] = make_uint4(i, 1, 2, 3);
}
// grid-stride: K stores per thread, each a full grid apart (every "round" is an address-linear sweep)
template <int K>
__global__ __launch_bounds__(256) void lin_stride(uint4* out, int64_t total) {
    const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
#pragma unroll
    for (int i = 0; i < K; i++) if (g0 + i * stride < total) out[g0 + i * stride] = make_uint4(i, 1, 2, 3);
}
// K stores per thread with a wait for the previous one (one store in flight per wave)
template <int K>
__global__ __launch_bounds__(256) void lin_wg_wait(uint4* out, int64_t total) {
    const int64_t g0 = (int64_t)blockIdx.x * 256 * K + threadIdx.x;
#pragma unroll
    for (int i = 0; i < K; i++) {
        if (g0 + i * 256 < total) out[g0 + i * 256] = make_uint4(i, 1, 2, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
int main() {
    const int64_t total = (int64_t)65536 * 8 * 441;
    uint4* out; (void)hipMalloc(&out, total * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, total * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
#define G(per) (unsigned)((total + (per) - 1) / (per))
    run("1 store/thread, 64-thread workgroups", [&] { lin1<64><<<G(64), 64>>>(out, total); });
    run("1 store/thread, 256-thread workgroups", [&] { lin1<256><<<G(256), 256>>>(out, total); });
    run("1 store/thread, 1024-thread workgroups", [&] { lin1<1024><<<G(1024), 1024>>>(out, total); });
    run("2 stores/thread, 4 KB apart (workgroup span 8 KB)", [&] { lin_wg<2><<<G(512), 256>>>(out, total); });
    run("2 stores/thread, adjacent (wave span 2 KB)", [&] { lin_wave<2><<<G(512), 256>>>(out, total); });
    run("8 stores/thread, adjacent (wave span 8 KB)", [&] { lin_wave<8><<<G(2048), 256>>>(out, total); });
    run("2 stores/thread, a grid apart (two linear sweeps)", [&] { lin_stride<2><<<G(512), 256>>>(out, total); });
    run("8 stores/thread, a grid apart (eight linear sweeps)", [&] { lin_stride<8><<<G(2048), 256>>>(out, total); });
    run("2 stores/thread, 4 KB apart, wait between", [&] { lin_wg_wait<2><<<G(512), 256>>>(out, total); });
    return 0;
}
