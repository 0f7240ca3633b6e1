// One wavefront per env (56 KB), pure stores: does it help if neighbouring wavefronts do NOT write the same offset of
// their env at the same time?  (a) in order; (b) start rotated by the env index (tile granularity: 8 phases);
// (c) rotated at store granularity (56 phases); (d) every wavefront walks its env in a bit-reversed tile order.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int kTile = 441, kEnv = 8 * kTile;
template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* out, int64_t n) {
    __shared__ uint4 lds[4][608];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t env = (int64_t)blockIdx.x * 4 + wave;
    if (threadIdx.x == 1023) lds[0][0] = make_uint4(0, 0, 0, 0);
    if (env >= n) return;
    uint4* o = out + env * kEnv;
    if (MODE == 0 || MODE == 2) {
        const int rot = MODE == 2 ? (int)((env * 37) % 56) : 0;
        for (int s = 0; s < 56; s++) {
            int ss = s + rot; ss -= ss >= 56 ? 56 : 0;
            const int c = ss * 64 + lane;
            if (c < kEnv) o[c] = make_uint4(c, 1, 2, 3);
        }
    } else {
        const int rot = MODE == 1 ? (int)((env * 5) & 7) : 0;
        for (int t = 0; t < 8; t++) {
            int tt = MODE == 3 ? ((t & 1) << 2 | (t & 2) | (t >> 2)) : ((t + rot) & 7);
            uint4* ot = o + tt * kTile;
            for (int i = 0; i < 7; i++) if (lane + 64 * i < kTile) ot[lane + 64 * i] = make_uint4(t, i, 2, 3);
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
        printf("%-56s %8.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, total * 16.0 / (ms / 20 * 1e-3) / 1e12);
    };
    run("(a) in order", [&] { k<0><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("(b) tile order rotated by env (8 phases)", [&] { k<1><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("(c) store order rotated by env (56 phases)", [&] { k<2><<<(unsigned)(n / 4), 256>>>(out, n); });
    run("(d) bit-reversed tile order, same for every env", [&] { k<3><<<(unsigned)(n / 4), 256>>>(out, n); });
    return 0;
}
