// What costs an address-linear writer its bandwidth on the fused-84 tensor (3.70 GB)?  Pure zero stores, one 1-KiB block per
// wavefront, variants: aligned 1-KiB blocks vs 7 056-byte tiles cut into 6 x 64 + 57 chunks (unaligned pieces, partial
// wavefronts), static LDS in the workgroup, a 256-byte kernel-argument block read piecemeal, one dependent header load.
//   hipcc --offload-arch=gfx950 -O3 sweep_overhead_probe.hip -o sweep_overhead_probe && ./sweep_overhead_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct Big { int a[60]; const uint2 *hdr; };
template <int MODE>
__global__ __launch_bounds__(256) void k(uint4 *__restrict__ out, int n_tiles, Big big) {
    __shared__ uint8_t lds[(MODE & 2) ? 6144 : 16];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (MODE & 2) { lds[threadIdx.x] = (uint8_t)b; if (b == -1) v.x = lds[(threadIdx.x + 1) & 255]; }
    if (MODE & 1) {  // tiles of 441 chunks, 7 blocks each
        const int tile = b / 7, blk = b - tile * 7;
        if (tile >= n_tiles) return;
        const int c = blk * 64 + lane;
        if (MODE & 8) { const uint2 h = big.hdr[(int64_t)tile * 8]; if ((h.y & 255u) == 7u) v.y = h.x; }
        if (MODE & 4) { if (big.a[blk * 7] == c) v.z = big.a[blk * 7 + 1]; if (big.a[blk + 50] == lane) v.w = 1; }
        if (c < 441) out[(int64_t)tile * 441 + c] = v;
    } else {
        if ((int64_t)b * 64 + lane < (int64_t)n_tiles * 441) out[(int64_t)b * 64 + lane] = v;
    }
}
int main() {
    const int n_tiles = 65536 * 8;
    const int64_t total = (int64_t)n_tiles * 441;
    uint4 *out; uint2 *hdr;
    (void)hipMalloc(&out, total * 16 + 4096); (void)hipMalloc(&hdr, (size_t)n_tiles * 64); (void)hipMemset(hdr, 0, (size_t)n_tiles * 64);
    Big big; for (int i = 0; i < 60; i++) big.a[i] = -5 - i; big.hdr = hdr;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-70s %8.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, total * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
    const unsigned g_lin = (unsigned)((total + 255) / 256), g_tile = (unsigned)(((int64_t)n_tiles * 7 + 3) / 4);
#define L(M, G) [&] { k<M><<<G, 256>>>(out, n_tiles, big); }
    run("aligned 1-KiB blocks", L(0, g_lin));
    run("aligned + 6 KB static LDS per workgroup", L(2, g_lin));
    run("7 056-byte tiles as 6 x 64 + 57 chunks", L(1, g_tile));
    run("tiles + LDS", L(3, g_tile));
    run("tiles + kernarg block read piecemeal", L(5, g_tile));
    run("tiles + one dependent 8-byte header load per wavefront", L(9, g_tile));
    run("tiles + LDS + kernargs + header", L(15, g_tile));
    return 0;
}
