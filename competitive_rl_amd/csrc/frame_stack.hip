// frame_stack.hip -- FrameStackTensor.update as ONE pass over the stack.
//
// Reference (competitive_rl/utils/utils.py:158-170): `current_obs *= mask`, `roll(-C, dim=1)`, then the newest
// observation goes into the last C planes -- three full passes over a float32 (N, C*k, H, W) tensor plus a host ->
// device copy of the observation.  Here the observation is already in HBM (u8 from the env kernels, or f32) and the
// update is a single in-place sweep: every thread owns one 16-byte column position of one env and walks down the
// planes (load plane p+C, scale by the env's mask, store plane p; then convert and store the new planes), so the
// in-place shift has no cross-thread hazard and each wavefront moves 1 KiB contiguous per plane.
// HBM bytes per env (k planes of hw floats): (k-1)*4*hw read + k*4*hw written + the observation; torch's three ops
// move ~2.5x that.  Bound: HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "crl_internal.h"

namespace crl {

// src == stack: the in-place shift.  src != stack (round 5, crl_frame_stack_update_to): the same planes read from ANOTHER tensor -- a plain
// streaming copy with no load that has to stay ahead of a store to the same rows.
template <int VEC, bool OBS_F32>
__global__ __launch_bounds__(256) void frame_stack_update_kernel(float *stack, const float *src_stack, const void *__restrict__ obs,
                                                                 int64_t obs_env_stride, const float *__restrict__ mask,
                                                                 int64_t n, int c, int k, int64_t hw) {
    const int64_t per_env = hw / VEC;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * per_env) return;
    const int64_t env = t / per_env, x = (t - env * per_env) * VEC;
    const int planes = c * k, keep = planes - c;
    const float m = mask ? mask[env] : 1.0f;
    float *base = stack + env * planes * hw + x;
    const float *sbase = src_stack + env * planes * hw + x;
    if (VEC == 4) {
        // (round 5) the kept planes are read in batches of up to eight with ALL of a batch's loads issued before its first store (the in-place
        // shift reads and writes one array, so the compiler otherwise orders load p+1 behind store p-1), streaming loads and stores.  A batch's
        // source planes lie above every plane the batch writes, and batches walk upwards, so no store can overtake a load of a later batch.
        // Measured (tools/frame_stack_time.py, 65 536 x (4, 84, 84) float32 + a u8 plane = 13.4 GB): 2.99 ms before, 2.99 batched, 2.94 with
        // the streaming hints = 4.56 TB/s, 0.57 of the HBM peak: the mixed read / write stream of an in-place shift, not the load order, is
        // what holds it below a plain copy's 6.3 TB/s.
        constexpr int kB = 8;
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int p0 = 0; p0 < keep; p0 += kB) {
            f4 v[kB];
#pragma unroll
            for (int j = 0; j < kB; j++)
                if (p0 + j < keep) v[j] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(sbase + (int64_t)(p0 + j + c) * hw));
#pragma unroll
            for (int j = 0; j < kB; j++)
                if (p0 + j < keep) {
                    if (mask) v[j] *= m;
                    __builtin_nontemporal_store(v[j], reinterpret_cast<f4 *>(base + (int64_t)(p0 + j) * hw));
                }
        }
        for (int q = 0; q < c; q++) {
            float4 v;
            if (OBS_F32) {
                v = *reinterpret_cast<const float4 *>(static_cast<const float *>(obs) + env * obs_env_stride + (int64_t)q * hw + x);
            } else {
                const uchar4 u = *reinterpret_cast<const uchar4 *>(static_cast<const uint8_t *>(obs) + env * obs_env_stride + (int64_t)q * hw + x);
                v = make_float4((float)u.x, (float)u.y, (float)u.z, (float)u.w);
            }
            f4 w;
            w.x = v.x, w.y = v.y, w.z = v.z, w.w = v.w;
            __builtin_nontemporal_store(w, reinterpret_cast<f4 *>(base + (int64_t)(keep + q) * hw));
        }
    } else {
        for (int p = 0; p < keep; p++) {
            float v = sbase[(int64_t)(p + c) * hw];
            if (mask) v *= m;
            base[(int64_t)p * hw] = v;
        }
        for (int q = 0; q < c; q++) {
            const int64_t o = env * obs_env_stride + (int64_t)q * hw + x;
            base[(int64_t)(keep + q) * hw] = OBS_F32 ? static_cast<const float *>(obs)[o] : (float)static_cast<const uint8_t *>(obs)[o];
        }
    }
}

// Out of place and FLAT (round 5): per env the kept planes are ONE contiguous run in the source (planes c .. c k - 1) and one in the
// destination (planes 0 .. c (k - 1) - 1), so the shift is a linear copy displaced by c planes -- two address streams per env instead of
// the column walk's 2 k -- followed by the widened observation.  Four 16-byte chunks per thread, all loads before the stores.
#ifndef CRL_FS_U
#define CRL_FS_U 1  // 16-byte chunks per thread (measured at 65 536 x (4, 84, 84): 1: 2.16 ms, 2: 2.27, 4: 2.42, 8: 2.52)
#endif
#ifndef CRL_FS_NT
#define CRL_FS_NT 3  // bit 0: streaming loads, bit 1: streaming stores (measured: 0: 2.36 ms, 1: 2.38, 2: 2.17, 3: 2.15 -- the stores are what matters; the float32 observation kernel of pong_raster_gray.hip, which stores from LDS tiles, LOSES 5 % with streaming stores)
#endif
template <bool OBS_F32>
__global__ __launch_bounds__(256) void frame_stack_copy_kernel(float *__restrict__ dst, const float *__restrict__ src, const void *__restrict__ obs,
                                                               int64_t obs_env_stride, const float *__restrict__ mask, int64_t n, int c, int k, int64_t hw) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int U = CRL_FS_U;
    const int64_t shiftc = (int64_t)c * hw / 4, totc = shiftc * k, keepc = totc - shiftc;
    const int64_t per_env = (totc + 256 * U - 1) / (256 * U);
    const int64_t env = blockIdx.x / per_env;
    if (env >= n) return;
    const int64_t q0 = (blockIdx.x - env * per_env) * (256 * U) + threadIdx.x;
    const float m = mask ? mask[env] : 1.0f;
    const f4 *s4 = reinterpret_cast<const f4 *>(src) + env * totc + shiftc;
    f4 *d4 = reinterpret_cast<f4 *>(dst) + env * totc;
    f4 v[U];
#pragma unroll
    for (int j = 0; j < U; j++) {
        const int64_t q = q0 + 256 * j;
        if (q < keepc) {
            v[j] = CRL_FS_NT & 1 ? __builtin_nontemporal_load(s4 + q) : s4[q];
        } else if (q < totc) {
            const int64_t e = (q - keepc) * 4;  // element inside the env's observation (c planes of hw)
            if (OBS_F32) {
                v[j] = *reinterpret_cast<const f4 *>(static_cast<const float *>(obs) + env * obs_env_stride + e);
            } else {
                const uchar4 u = *reinterpret_cast<const uchar4 *>(static_cast<const uint8_t *>(obs) + env * obs_env_stride + e);
                v[j].x = (float)u.x, v[j].y = (float)u.y, v[j].z = (float)u.z, v[j].w = (float)u.w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < U; j++) {
        const int64_t q = q0 + 256 * j;
        if (q < keepc) {
            if (mask) v[j] *= m;
            if (CRL_FS_NT & 2) __builtin_nontemporal_store(v[j], d4 + q);
            else d4[q] = v[j];
        } else if (q < totc) {
            if (CRL_FS_NT & 2) __builtin_nontemporal_store(v[j], d4 + q);
            else d4[q] = v[j];
        }
    }
}

// The generic update of a uint8 stack (FrameStackTensor(dtype=torch.uint8): opt-in, a quarter of the bytes; its hot path is the stack drawn
// by the step, this is what serves foreign observations, masks of the caller's own, skipped updates).  Same column walk as
// frame_stack_update_kernel: a thread owns one 16-byte (or one-byte) column position of one env and walks up the planes, so the shift is
// safe in place (dst == src) and out of place alike.  A mask entry of 0 erases the env's history, anything else keeps it (bytes cannot be
// scaled); a float32 observation holds 0..255 integers and is truncated like a tensor cast.
template <int VEC, bool OBS_F32>
__global__ __launch_bounds__(256) void frame_stack_update_u8_kernel(uint8_t *dst, const uint8_t *src, const void *__restrict__ obs, int64_t obs_env_stride,
                                                                    const float *__restrict__ mask, int64_t n, int c, int k, int64_t hw) {
    const int64_t per_env = hw / VEC;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * per_env) return;
    const int64_t env = t / per_env, x = (t - env * per_env) * VEC;
    const int planes = c * k, keep = planes - c;
    const bool erase = mask && mask[env] == 0.0f;
    uint8_t *base = dst + env * planes * hw + x;
    const uint8_t *sbase = src + env * planes * hw + x;
    if (VEC == 16) {
        for (int p = 0; p < keep; p++) {
            uint4 v = *reinterpret_cast<const uint4 *>(sbase + (int64_t)(p + c) * hw);
            if (erase) v = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4 *>(base + (int64_t)p * hw) = v;
        }
        for (int q = 0; q < c; q++) {
            uint4 v;
            if (OBS_F32) {
                const float4 *f = reinterpret_cast<const float4 *>(static_cast<const float *>(obs) + env * obs_env_stride + (int64_t)q * hw + x);
                uint32_t wds[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float4 a = f[i];
                    wds[i] = (uint32_t)(uint8_t)(int)a.x | (uint32_t)(uint8_t)(int)a.y << 8 | (uint32_t)(uint8_t)(int)a.z << 16 | (uint32_t)(uint8_t)(int)a.w << 24;
                }
                v = make_uint4(wds[0], wds[1], wds[2], wds[3]);
            } else {
                v = *reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(obs) + env * obs_env_stride + (int64_t)q * hw + x);
            }
            *reinterpret_cast<uint4 *>(base + (int64_t)(keep + q) * hw) = v;
        }
    } else {
        for (int p = 0; p < keep; p++) {
            const uint8_t v = sbase[(int64_t)(p + c) * hw];
            base[(int64_t)p * hw] = erase ? (uint8_t)0 : v;
        }
        for (int q = 0; q < c; q++) {
            const int64_t o = env * obs_env_stride + (int64_t)q * hw + x;
            base[(int64_t)(keep + q) * hw] = OBS_F32 ? (uint8_t)(int)static_cast<const float *>(obs)[o] : static_cast<const uint8_t *>(obs)[o];
        }
    }
}

}  // namespace crl

static int frame_stack_update_impl(float *stack_dev, const float *src_dev, const void *obs_dev, int32_t obs_dtype, int64_t obs_env_stride,
                                   const float *mask_dev, int64_t n, int32_t c, int32_t k, int64_t hw, void *stream) {
    crl_fail_no_ctx();
    if (!stack_dev || !obs_dev || !src_dev) return crl_fail(CRL_EINVAL, "frame_stack_update: null tensor");
    if (n <= 0 || c <= 0 || k <= 0 || hw <= 0) return crl_fail(CRL_EINVAL, "frame_stack_update: bad shape n=%lld c=%d k=%d hw=%lld", (long long)n, c, k, (long long)hw);
    if (obs_dtype != CRL_OBS_U8 && obs_dtype != CRL_OBS_F32) return crl_fail(CRL_EINVAL, "frame_stack_update: obs_dtype %d", obs_dtype);
    if (obs_env_stride < (int64_t)c * hw) return crl_fail(CRL_EINVAL, "frame_stack_update: obs_env_stride %lld < c*hw", (long long)obs_env_stride);
    if (src_dev != stack_dev) {  // two tensors: they must not overlap at all
        const int64_t bytes = n * c * k * hw * (int64_t)sizeof(float);
        const uintptr_t a = (uintptr_t)stack_dev, b = (uintptr_t)src_dev;
        if (a < b + (uintptr_t)bytes && b < a + (uintptr_t)bytes) return crl_fail(CRL_EINVAL, "frame_stack_update_to: destination and source stacks overlap");
    }
    hipStream_t st = (hipStream_t)stream;
    const bool f32 = obs_dtype == CRL_OBS_F32;
    const size_t obs_align = f32 ? 16 : 4;
    const bool vec = hw % 4 == 0 && ((uintptr_t)stack_dev % 16) == 0 && ((uintptr_t)src_dev % 16) == 0 && ((uintptr_t)obs_dev % obs_align) == 0 &&
                     (obs_env_stride * (f32 ? 4 : 1)) % (int64_t)obs_align == 0;
    const int64_t threads = n * (vec ? hw / 4 : hw);
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    using namespace crl;
    static const bool flat_off = getenv("CRL_FRAME_STACK_COLUMNS") != nullptr;  // (A/B: the out-of-place update as the column walk)
    if (vec && src_dev != stack_dev && !flat_off) {
        const int64_t totc = (int64_t)c * k * hw / 4, per_env = (totc + 256 * CRL_FS_U - 1) / (256 * CRL_FS_U);
        if (n * per_env < (int64_t)1 << 31) {
            const dim3 g2((unsigned)(n * per_env));
            if (f32) hipLaunchKernelGGL((frame_stack_copy_kernel<true>), g2, block, 0, st, stack_dev, src_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
            else hipLaunchKernelGGL((frame_stack_copy_kernel<false>), g2, block, 0, st, stack_dev, src_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
            const hipError_t e2 = hipGetLastError();
            if (e2 != hipSuccess) return crl_fail(CRL_EHIP, "frame_stack_update_to: %s", hipGetErrorString(e2));
            return CRL_OK;
        }
    }
    if (vec && f32) hipLaunchKernelGGL((frame_stack_update_kernel<4, true>), grid, block, 0, st, stack_dev, src_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    else if (vec) hipLaunchKernelGGL((frame_stack_update_kernel<4, false>), grid, block, 0, st, stack_dev, src_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    else if (f32) hipLaunchKernelGGL((frame_stack_update_kernel<1, true>), grid, block, 0, st, stack_dev, src_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    else hipLaunchKernelGGL((frame_stack_update_kernel<1, false>), grid, block, 0, st, stack_dev, src_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "frame_stack_update: %s", hipGetErrorString(e));
    return CRL_OK;
}

extern "C" int crl_frame_stack_update(float *stack_dev, const void *obs_dev, int32_t obs_dtype, int64_t obs_env_stride,
                                      const float *mask_dev, int64_t n, int32_t c, int32_t k, int64_t hw, void *stream) {
    return frame_stack_update_impl(stack_dev, stack_dev, obs_dev, obs_dtype, obs_env_stride, mask_dev, n, c, k, hw, stream);
}
extern "C" int crl_frame_stack_update_to(float *dst_stack_dev, const float *src_stack_dev, const void *obs_dev, int32_t obs_dtype, int64_t obs_env_stride,
                                         const float *mask_dev, int64_t n, int32_t c, int32_t k, int64_t hw, void *stream) {
    return frame_stack_update_impl(dst_stack_dev, src_stack_dev, obs_dev, obs_dtype, obs_env_stride, mask_dev, n, c, k, hw, stream);
}

extern "C" int crl_frame_stack_update_u8(uint8_t *dst_stack_dev, const uint8_t *src_stack_dev, const void *obs_dev, int32_t obs_dtype, int64_t obs_env_stride,
                                         const float *mask_dev, int64_t n, int32_t c, int32_t k, int64_t hw, void *stream) {
    crl_fail_no_ctx();
    if (!dst_stack_dev || !src_stack_dev || !obs_dev) return crl_fail(CRL_EINVAL, "frame_stack_update_u8: null tensor");
    if (n <= 0 || c <= 0 || k <= 0 || hw <= 0) return crl_fail(CRL_EINVAL, "frame_stack_update_u8: bad shape n=%lld c=%d k=%d hw=%lld", (long long)n, c, k, (long long)hw);
    if (obs_dtype != CRL_OBS_U8 && obs_dtype != CRL_OBS_F32) return crl_fail(CRL_EINVAL, "frame_stack_update_u8: obs_dtype %d", obs_dtype);
    if (obs_env_stride < (int64_t)c * hw) return crl_fail(CRL_EINVAL, "frame_stack_update_u8: obs_env_stride %lld < c*hw", (long long)obs_env_stride);
    if (src_stack_dev != dst_stack_dev) {  // two tensors: they must not overlap at all (the same tensor = the in-place update)
        const int64_t bytes = n * c * k * hw;
        const uintptr_t a = (uintptr_t)dst_stack_dev, b = (uintptr_t)src_stack_dev;
        if (a < b + (uintptr_t)bytes && b < a + (uintptr_t)bytes) return crl_fail(CRL_EINVAL, "frame_stack_update_u8: destination and source stacks overlap");
    }
    const bool f32 = obs_dtype == CRL_OBS_F32;
    const bool vec = hw % 16 == 0 && ((uintptr_t)dst_stack_dev % 16) == 0 && ((uintptr_t)src_stack_dev % 16) == 0 && ((uintptr_t)obs_dev % 16) == 0 &&
                     (obs_env_stride * (f32 ? 4 : 1)) % 16 == 0;
    const int64_t threads = n * (vec ? hw / 16 : hw);
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    using namespace crl;
    if (vec && f32) hipLaunchKernelGGL((frame_stack_update_u8_kernel<16, true>), grid, block, 0, st, dst_stack_dev, src_stack_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    else if (vec) hipLaunchKernelGGL((frame_stack_update_u8_kernel<16, false>), grid, block, 0, st, dst_stack_dev, src_stack_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    else if (f32) hipLaunchKernelGGL((frame_stack_update_u8_kernel<1, true>), grid, block, 0, st, dst_stack_dev, src_stack_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    else hipLaunchKernelGGL((frame_stack_update_u8_kernel<1, false>), grid, block, 0, st, dst_stack_dev, src_stack_dev, obs_dev, obs_env_stride, mask_dev, n, c, k, hw);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return crl_fail(CRL_EHIP, "frame_stack_update_u8: %s", hipGetErrorString(e));
    return CRL_OK;
}
