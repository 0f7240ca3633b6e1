// Does an address-linear writer of the fused-84 tensor (3.70 GB, tiles of 441 16-byte chunks) reach the fill rate when its per-tile
// metadata comes in through the SCALAR cache (s_load: not queued behind the CU's vector stores) instead of a vector load?  Round 2 / 6
// measured the skeleton with a dependent VECTOR read in front of every store at 975 us (the env kernel: 760; pure aligned stores: 540).
// One wavefront per 1-KiB-aligned block of the tensor (NB blocks per wavefront, a grid apart); a block holds the tail of one tile and the
// head of the next: both tiles' 8-byte records in ONE s_load_dwordx4; template chunks (score band rows of the tile's score pair, the
// white bottom rows) by vector loads whose addresses depend on scalar data only; court rows are zeros.
//   hipcc --offload-arch=gfx950 -O3 tools/sweep_sload_probe.hip -o tools/sweep_sload_probe && ./tools/sweep_sload_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

constexpr int kChunks = 441, kBand = 74, kZero0 = 74, kZero1 = 404;  // chunks per tile; band rows 0..13; zero rows up to row 77

// MODE 0: pure stores.  1: metadata by vector load (one 8-byte record per lane's tile).  2: metadata by scalar load.
template <int MODE, int NB>
__global__ __launch_bounds__(256) void sweep(uint4 *__restrict__ out, const uint2 *__restrict__ meta, const uint4 *__restrict__ band,
                                             const uint4 *__restrict__ rest, int64_t total, int64_t stride_blocks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t b0 = (int64_t)blockIdx.x * 4 + wave;
    uint4 v[NB];
    int64_t g[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int64_t b = b0 + i * stride_blocks;
        g[i] = b * 64 + lane;
        v[i] = make_uint4(0, 0, 0, 0);
        if (MODE == 0) continue;
        const int64_t g0 = b * 64;                            // wave-uniform
        const int tile0 = (int)(g0 / kChunks);
        const int split = (int)((int64_t)(tile0 + 1) * kChunks - g0);  // lanes >= split belong to tile0 + 1
        uint2 m;
        if (MODE == 1) {
            m = meta[tile0 + (lane >= split ? 1 : 0)];
        } else {
            const uint2 m0 = meta[tile0], m1 = meta[tile0 + 1];   // uniform addresses: s_load
            m.x = lane >= split ? m1.x : m0.x, m.y = lane >= split ? m1.y : m0.y;
        }
        const int c = lane >= split ? lane - split : (int)(g0 - (int64_t)tile0 * kChunks) + lane;
        if (m.y != 0) {
            if (c < kBand) v[i] = band[(int64_t)m.x * kBand + c];
            else if (c >= kZero1) v[i] = rest[c];
        }
    }
#pragma unroll
    for (int i = 0; i < NB; i++)
        if (g[i] < total) out[g[i]] = v[i];
}

int main() {
    const int n_tiles = 65536 * 8;
    const int64_t total = (int64_t)n_tiles * kChunks;
    uint4 *out, *band, *rest;
    uint2 *meta;
    (void)hipMalloc(&out, total * 16 + (1 << 20));
    (void)hipMalloc(&meta, (size_t)(n_tiles + 8) * 8);
    (void)hipMalloc(&band, (size_t)968 * 3 * kBand * 16);
    (void)hipMalloc(&rest, (size_t)kChunks * 16);
    (void)hipMemset(band, 7, (size_t)968 * 3 * kBand * 16);
    (void)hipMemset(rest, 9, (size_t)kChunks * 16);
    std::vector<uint2> h(n_tiles + 8);
    for (int t = 0; t < n_tiles + 8; t++) h[t] = make_uint2(((uint32_t)(t / 8) * 2654435761u >> 12) % (968 * 3), 1);  // an env's 8 tiles share a score pair
    (void)hipMemcpy(meta, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int64_t blocks = (total + 63) / 64;
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; r++) launch();
        (void)hipEventRecord(e1), (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %8.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, total * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
#define L(M, NBK) [&] { const int64_t per = (blocks + NBK - 1) / NBK; const unsigned grid = (unsigned)((per + 3) / 4); \
                        sweep<M, NBK><<<grid, 256>>>(out, meta, band, rest, total, (int64_t)grid * 4); }
    for (int rep = 0; rep < 2; rep++) {
        run("pure stores, 1 block / wavefront", L(0, 1));
        run("pure stores, 4 blocks / wavefront a grid apart", L(0, 4));
        run("vector metadata + templates, 1 block / wavefront", L(1, 1));
        run("vector metadata + templates, 4 blocks", L(1, 4));
        run("SCALAR metadata + templates, 1 block / wavefront", L(2, 1));
        run("SCALAR metadata + templates, 2 blocks", L(2, 2));
        run("SCALAR metadata + templates, 4 blocks", L(2, 4));
        run("SCALAR metadata + templates, 8 blocks", L(2, 8));
    }
    return 0;
}
