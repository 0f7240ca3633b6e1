// How much of the fused-84 kernel's store time is the per-tile stream-out shape?  One wavefront per env (56 448 B),
// pure stores, 38 KB LDS per workgroup as in production:
//   A: the env as ONE aligned stream (56 full-wave stores of 1 KiB, 128-byte aligned)
//   B: tile by tile as the kernel does: 8 x (6 full stores + one 57-lane store), tiles start at multiples of 7 056 B
//   C: B with the data read from LDS first (ds_read_b128 -> global_store_dwordx4), 7 reads issued before the stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int kTile = 441, kEnv = 8 * kTile;
template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* out, int64_t n) {
    __shared__ uint4 lds[4][608];  // 38 KB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t env = (int64_t)blockIdx.x * 4 + wave;
    if (MODE == 2) for (int i = lane; i < kTile; i += 64) lds[wave][i] = make_uint4(i, 1, 2, 3);
    if (env >= n) return;
    uint4* o = out + env * kEnv;
    if (MODE == 0) {
        for (int c = lane; c < kEnv; c += 64) o[c] = make_uint4(c, 1, 2, 3);
    } else {
#pragma unroll 1
        for (int t = 0; t < 8; t++) {
            uint4* ot = o + t * kTile;
            uint4 v[7];
#pragma unroll
            for (int i = 0; i < 7; i++) v[i] = MODE == 2 ? lds[wave][(lane + 64 * i) % 608] : make_uint4(t, i, 2, 3);
#pragma unroll
            for (int i = 0; i < 7; i++) if (lane + 64 * i < kTile) ot[lane + 64 * i] = v[i];
        }
    }
}
int main() {
    const int64_t n = 65536, total = n * kEnv;
    uint4* out; (void)hipMalloc(&out, total * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 20; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %8.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, total * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
    run("A: env as one aligned stream", [&] { k<0><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("B: tile by tile (6 full + one 57-lane store, 7 056-B tiles)", [&] { k<1><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("C: B, data through LDS (ds_read_b128 first)", [&] { k<2><<<(unsigned)(n / 4), 256>>>(out, n); });
    return 0;
}
