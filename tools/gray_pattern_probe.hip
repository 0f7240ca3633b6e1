// Which part of the fused-84 kernel's 752 us is the STORE PATTERN?  Same tensor (65 536 x 2 x 4 x 84 x 84 u8 = 3.70 GB),
// constant data, different assignments of the 16-byte chunks to wavefronts.  38 KB of LDS per 256-thread workgroup
// (as the production kernel) => 4 workgroups = 16 waves per CU.
//   hipcc --offload-arch=gfx950 -O3 gray_pattern_probe.hip -o gray_pattern_probe && ./gray_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int kTile = 441;          // chunks per plane
constexpr int kEnv = 8 * kTile;     // chunks per env
template <int LDSKB>
__device__ inline void touch_lds() {
    __shared__ uint4 pad[LDSKB * 64];
    if (threadIdx.x == 1023) pad[0] = make_uint4(1, 2, 3, 4);  // never true: keeps the allocation
}
// P1: one wavefront per env, its 56 KB in ascending order (production pattern)
template <int LDSKB>
__global__ __launch_bounds__(256) void p_wave_env(uint4* out, int64_t n) {
    touch_lds<LDSKB>();
    const int64_t env = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= n) return;
    uint4* o = out + env * kEnv;
    const int lane = threadIdx.x & 63;
    for (int c = lane; c < kEnv; c += 64) o[c] = make_uint4(c, 1, 2, 3);
}
// P2: one wavefront per plane (7 KB), planes in address order
template <int LDSKB>
__global__ __launch_bounds__(256) void p_wave_plane(uint4* out, int64_t tiles) {
    touch_lds<LDSKB>();
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= tiles) return;
    uint4* o = out + t * kTile;
    const int lane = threadIdx.x & 63;
    for (int c = lane; c < kTile; c += 64) o[c] = make_uint4(c, 1, 2, 3);
}
// P3: one workgroup (4 waves) per env, the four waves sweep the env's 56 KB together (4 KB per sweep)
template <int LDSKB>
__global__ __launch_bounds__(256) void p_wg_env(uint4* out, int64_t n) {
    touch_lds<LDSKB>();
    uint4* o = out + (int64_t)blockIdx.x * kEnv;
    for (int c = threadIdx.x; c < kEnv; c += 256) o[c] = make_uint4(c, 1, 2, 3);
}
// P4: one workgroup per 4 envs, swept together in address order (4 KB per sweep over 226 KB)
template <int LDSKB>
__global__ __launch_bounds__(256) void p_wg_4env(uint4* out, int64_t n) {
    touch_lds<LDSKB>();
    uint4* o = out + (int64_t)blockIdx.x * 4 * kEnv;
    for (int c = threadIdx.x; c < 4 * kEnv; c += 256) o[c] = make_uint4(c, 1, 2, 3);
}
// P3b: one 1 024-thread workgroup (16 waves) per env: 16 KB per sweep
__global__ __launch_bounds__(1024) void p_wg1024_env(uint4* out, int64_t n) {
    touch_lds<38>();
    uint4* o = out + (int64_t)blockIdx.x * kEnv;
    for (int c = threadIdx.x; c < kEnv; c += 1024) o[c] = make_uint4(c, 1, 2, 3);
}
// P3c: one 1 024-thread workgroup per 4 envs (the production kernel's envs per workgroup), 16 KB per sweep over 226 KB
__global__ __launch_bounds__(1024) void p_wg1024_4env(uint4* out, int64_t n) {
    touch_lds<38>();
    uint4* o = out + (int64_t)blockIdx.x * 4 * kEnv;
    for (int c = threadIdx.x; c < 4 * kEnv; c += 1024) o[c] = make_uint4(c, 1, 2, 3);
}
// P5: address-linear, 1 chunk per thread
__global__ __launch_bounds__(256) void p_linear(uint4* out, int64_t total) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < total) out[g] = make_uint4((unsigned)g, 1, 2, 3);
}
int main() {
    const int64_t n = 65536, tiles = n * 8, total = tiles * kTile;
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
    run("address-linear, 1 chunk/thread", [&] { p_linear<<<(unsigned)((total + 255) / 256), 256>>>(out, total); });
    run("wave per env (56 KB each), 16 waves/CU", [&] { p_wave_env<38><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("wave per env, 32 waves/CU", [&] { p_wave_env<16><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("wave per plane (7 KB each), 16 waves/CU", [&] { p_wave_plane<38><<<(unsigned)(tiles / 4), 256>>>(out, tiles); });
    run("wave per plane, 32 waves/CU", [&] { p_wave_plane<16><<<(unsigned)(tiles / 4), 256>>>(out, tiles); });
    run("workgroup per env (4 waves sweep 56 KB), 16 waves/CU", [&] { p_wg_env<38><<<(unsigned)n, 256>>>(out, n); });
    run("workgroup per env, 32 waves/CU", [&] { p_wg_env<16><<<(unsigned)n, 256>>>(out, n); });
    run("workgroup per 4 envs (sweep 226 KB), 16 waves/CU", [&] { p_wg_4env<38><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("1 024-thread workgroup per env (16 waves sweep 56 KB)", [&] { p_wg1024_env<<<(unsigned)n, 1024>>>(out, n); });
    run("1 024-thread workgroup per 4 envs (sweep 226 KB)", [&] { p_wg1024_4env<<<(unsigned)(n / 4), 1024>>>(out, n); });
    return 0;
}
