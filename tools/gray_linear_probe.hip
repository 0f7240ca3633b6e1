// Probe for the next round (DESIGN 4.3 open question): what does an ADDRESS-LINEAR writer sustain on the fused-84
// output tensor (N, 2, 4, 84, 84) u8 = 3.70 GB at 65 536 envs?  Workgroup b writes chunks [b*256*ITERS, ...) of the
// whole tensor; every 16-byte chunk = template chunk (7 KB table, L2 resident) + a cheap per-tile overlay test
// against a 64-byte tile header (6 boxes), as a two-kernel design (patch kernel + linear writer) would do.
//   hipcc --offload-arch=gfx950 -O3 gray_linear_probe.hip -o gray_linear_probe && ./gray_linear_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
struct TileHdr { uint8_t x0[6], x1[6], y0[6], y1[6]; uint32_t patch_ofs; uint32_t pad; };  // 32 B
template <int ITERS, int MODE>
__global__ __launch_bounds__(256) void writer(uint4* __restrict__ out, const uint4* __restrict__ tmpl, const TileHdr* __restrict__ hdr,
                                              const uint8_t* __restrict__ patch, int64_t total) {
    const int64_t g0 = (int64_t)blockIdx.x * (256 * ITERS);
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const int64_t g = g0 + i * 256 + threadIdx.x;
        if (g >= total) break;
        const int64_t tile = g / 441;
        const int c = (int)(g - tile * 441);
        uint4 v = MODE >= 1 ? tmpl[c] : make_uint4(0, 0, 0, 0);
        if (MODE >= 2) {
            const TileHdr h = hdr[tile];
            const int b0 = c * 16, r0 = b0 / 84, r1 = (b0 + 15) / 84;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (r1 >= h.y0[k] && r0 < h.y1[k]) {  // rare: fetch the patch bytes of this box row
                    const uint8_t* p = patch + h.patch_ofs + k * 32;
                    uint8_t* vb = reinterpret_cast<uint8_t*>(&v);
                    for (int j = 0; j < 16; j++) {
                        const int b = b0 + j, r = b / 84, x = b - r * 84;
                        if (r >= h.y0[k] && r < h.y1[k] && x >= h.x0[k] && x < h.x1[k]) vb[j] = p[(r - h.y0[k]) * 4 + (x - h.x0[k])];
                    }
                }
            }
        }
        out[g] = v;
    }
}
int main() {
    const int64_t n = 65536, tiles = n * 8, total = tiles * 441;
    uint4 *out, *tmpl; TileHdr* hdr; uint8_t* patch;
    (void)hipMalloc(&out, total * 16); (void)hipMalloc(&tmpl, 441 * 16); (void)hipMalloc(&hdr, tiles * sizeof(TileHdr)); (void)hipMalloc(&patch, tiles * 192);
    (void)hipMemset(tmpl, 7, 441 * 16); (void)hipMemset(patch, 9, tiles * 192);
    TileHdr* hh = (TileHdr*)malloc(tiles * sizeof(TileHdr));
    for (int64_t t = 0; t < tiles; t++) {
        for (int k = 0; k < 6; k++) {
            const int x = (int)((t * 7 + k * 13) % 78), y = (int)(14 + (t * 3 + k * 11) % 60);
            hh[t].x0[k] = x, hh[t].x1[k] = x + 4, hh[t].y0[k] = y, hh[t].y1[k] = y + (k < 2 ? 3 : 7);
        }
        hh[t].patch_ofs = (uint32_t)(t * 192); hh[t].pad = 0;
    }
    (void)hipMemcpy(hdr, hh, tiles * sizeof(TileHdr), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %8.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, total * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
#define L(I, M) [&] { writer<I, M><<<(unsigned)((total + 256 * I - 1) / (256 * I)), 256>>>(out, tmpl, hdr, patch, total); }
    run("hipMemsetAsync", [&] { (void)hipMemsetAsync(out, 1, total * 16, 0); });
    run("linear 1 chunk/thread, constant", L(1, 0));
    run("linear 2 chunks/thread, constant", L(2, 0));
    run("linear 4 chunks/thread, constant", L(4, 0));
    run("linear 2 chunks/thread, template", L(2, 1));
    run("linear 2 chunks/thread, template+overlay", L(2, 2));
    run("linear 4 chunks/thread, template+overlay", L(4, 2));
    return 0;
}
