// How fast can ONE wavefront run the 180 velocity iterations of a car's four revolute joints?  (the per-car
// solve and the touching solve are dependent-instruction chains of lone wavefronts.)  Runs the production loop body
// (car_solver.h) and restructured variants on the same synthetic cars and prints time per iteration and a checksum of the
// resulting state: a variant only counts if its checksum equals V0's bit for bit
// (the FM = true variants: V3's).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I include tools/solve_chain_probe.hip -o tools/solve_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../competitive_rl_amd/csrc/car_solver.h"

using namespace crl;

// Variants (round 5: the loop bodies are car_solver.h's own, template parameter FM = fused multiply-adds):
//   V0 isl_joints_vel<false> (branches per joint)   V1 isl_joints_vel_in<false> (no limit code)   V2 isl_joints_vel_sel<false> (selects)
//   V3 isl_joints_vel_in<true>                      V4 isl_joints_vel_sel<true>                   V5 isl_joints_vel<true>
// The checksum of V3-V5 differs from V0-V2's by design (one rounding per a*b+c instead of two); within a group it must agree.
template <int V>
__global__ __launch_bounds__(64) void probe(const CarConsts *Kp, int iters, int with_limits, uint32_t *sum, unsigned long long *ticks) {
    const CarConsts K = *Kp;
    const int lane = threadIdx.x, id = blockIdx.x * 64 + lane;
    CarRegs c;
    const float f = (float)((id * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    c.H.cx = 10.0f * f, c.H.cy = 3.0f - f, c.H.a = 6.0f * f - 3.0f, c.H.vx = 20.0f * f - 7.0f, c.H.vy = 9.0f * f, c.H.w = f - 0.5f;
    for (int w = 0; w < 4; w++) {
        c.W[w].cx = c.H.cx + 0.1f * w, c.W[w].cy = c.H.cy - 0.1f * w, c.W[w].a = c.H.a + (w < 2 ? 0.8f * f - 0.4f : 0.0f);
        c.W[w].vx = c.H.vx + 0.3f * w * f, c.W[w].vy = c.H.vy - 0.2f * f, c.W[w].w = 30.0f * f + w;
        c.imp[w][0] = c.imp[w][1] = c.imp[w][2] = 0.f, c.motor_imp[w] = 0.f, c.motor_speed[w] = w < 2 ? 3.0f * f - 1.5f : 0.f;
        c.lim[w] = LIM_INACTIVE, c.fx[w] = 100.f * f, c.fy[w] = -50.f * f;
    }
    JointTmp j;
    const float h = 1.0f / 50.0f;
    isl_integrate_vel<false>(c, K, h);
    isl_joints_init(c, j, K, 1.0f);
    if (!with_limits)
        for (int w = 0; w < 4; w++) c.lim[w] = LIM_INACTIVE, c.imp[w][2] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (V == 0) isl_joints_vel<false>(c, j, K, h);
        if (V == 1) isl_joints_vel_in<false>(c, j, K, h);
        if (V == 2) isl_joints_vel_sel<false>(c, j, K, h);
        if (V == 3) isl_joints_vel_in<true>(c, j, K, h);
        if (V == 4) isl_joints_vel_sel<true>(c, j, K, h);
        if (V == 5) isl_joints_vel<true>(c, j, K, h);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t x = __float_as_uint(c.H.vx) ^ (__float_as_uint(c.H.vy) * 3u) ^ (__float_as_uint(c.H.w) * 5u);
    for (int w = 0; w < 4; w++)
        x ^= __float_as_uint(c.W[w].vx) * (7u + w) ^ __float_as_uint(c.W[w].w) * (11u + w) ^ __float_as_uint(c.imp[w][0]) * (13u + w) ^
             __float_as_uint(c.imp[w][2]) * (17u + w) ^ __float_as_uint(c.motor_imp[w]) * (19u + w);
    atomicXor(sum, x * (uint32_t)(id + 1));
    if (id == 0) *ticks = t1 - t0;
}

int main() {
    CarConsts K{};
    K.hull_inv_mass = 0.24f, K.hull_inv_I = 0.36f, K.hull_lc[0] = 0.0f, K.hull_lc[1] = 0.2f;
    K.wheel_inv_mass = 2.3f, K.wheel_inv_I = 134.0f;
    const float ax[4] = {-1.1f, 1.1f, -1.1f, 1.1f}, ay[4] = {1.6f, 1.6f, -1.64f, -1.64f};
    for (int w = 0; w < 4; w++) K.anchor[w][0] = ax[w], K.anchor[w][1] = ay[w];
    CarConsts *Kd;
    uint32_t *sum;
    unsigned long long *ticks;
    hipMalloc(&Kd, sizeof(K)), hipMalloc(&sum, 4), hipMalloc(&ticks, 8);
    hipMemcpy(Kd, &K, sizeof(K), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 1800;
    for (int lim = 0; lim < 2; lim++)
        for (int v = 0; v < 6; v++) {
            if ((v == 1 || v == 3) && lim) continue;
            for (int blocks : {512, 16}) {
                float best = 1e9f;
                uint32_t hs = 0;
                unsigned long long ht = 0;
                for (int rep = 0; rep < 3; rep++) {
                    hipMemset(sum, 0, 4);
                    hipEventRecord(e0);
                    if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64), 0, 0, Kd, iters, lim, sum, ticks);
                    if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64), 0, 0, Kd, iters, lim, sum, ticks);
                    if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(64), 0, 0, Kd, iters, lim, sum, ticks);
                    if (v == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(64), 0, 0, Kd, iters, lim, sum, ticks);
                    if (v == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(64), 0, 0, Kd, iters, lim, sum, ticks);
                    if (v == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(64), 0, 0, Kd, iters, lim, sum, ticks);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                    hipMemcpy(&hs, sum, 4, hipMemcpyDeviceToHost), hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
                }
                printf("limits=%d V%d blocks=%4d: %8.1f us total, %7.1f ns / iteration, %8.1f counter ticks / iteration, checksum %08x\n", lim, v,
                       blocks, best * 1e3f, best * 1e6f / iters, (double)ht / iters, hs);
            }
        }
    return 0;
}
