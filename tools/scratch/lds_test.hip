#include <hip/hip_runtime.h>
__global__ void k(const uint4* src, uint4* dst) {
    __shared__ __attribute__((aligned(16))) uint4 sh[256];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + blockIdx.x * 256 + threadIdx.x),
                                     (__attribute__((address_space(3))) void*)sh, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dst[blockIdx.x * 256 + threadIdx.x] = sh[threadIdx.x ^ 1];
}
