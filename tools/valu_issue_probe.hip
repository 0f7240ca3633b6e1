// Issue rate of common vector instructions on gfx950, in cycles per wave64 instruction per SIMD, at 1 / 2 / 4 wavefronts per SIMD.
// Eight independent register chains per lane, 64 instructions per loop iteration; the clock is taken as 2.4 GHz (rocm-smi under load).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_probe.hip -o valu_issue_probe && ./valu_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define BODY(INSN)                                                        \
    for (int it = 0; it < iters; it++) {                                  \
        _Pragma("unroll") for (int r = 0; r < 8; r++)                     \
            _Pragma("unroll") for (int i = 0; i < 8; i++) { INSN; }       \
    }

template <int MODE>
__global__ void k(float *out, int iters) {
    float a[8];
    unsigned u[8];
    for (int i = 0; i < 8; i++) a[i] = (float)threadIdx.x + i, u[i] = threadIdx.x * 7 + i;
    float x = 1.0001f;
    unsigned y = 12345u + threadIdx.x;
    if (MODE == 0) BODY(asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(x)))
    if (MODE == 1) BODY(asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x)))
    if (MODE == 2) BODY(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x)))
    if (MODE == 3) BODY(asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(y)))
    if (MODE == 4) BODY(asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(y)))
    if (MODE == 5) BODY(asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i])))
    if (MODE == 6) BODY(asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(u[i])))
    if (MODE == 7) BODY(asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc"))
    if (MODE == 8) BODY(asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x)))
    if (MODE == 9) BODY(asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u[i]) : "v"(y)))
    if (MODE == 10) BODY(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x)))
    if (MODE == 11) BODY(asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(y)))
    if (MODE == 12) BODY(asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(y)))
    if (MODE == 13) BODY(asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double *)&u[i & 6])))
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out;
    (void)hipMalloc(&out, 64 << 20);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const char *names[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_cvt_f32_ubyte0", "v_cmp + v_cndmask (2)",
                           "v_max_f32", "v_mul_lo_u32", "v_cvt_pk_bf16_f32", "v_mov_b32", "v_perm_b32", "v_pk_mul_f32"};
    const int iters = 4000;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int wps = 1; wps <= 4; wps *= 2) {  // wavefronts per SIMD: one workgroup of 4 * wps wavefronts per CU
        printf("%d wavefront(s) per SIMD: cycles per wave-instruction per SIMD at 2.4 GHz\n", wps);
        for (int mode = 0; mode < 14; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
#define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(cus), dim3(256 * wps), 0, 0, out, iters);
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13)
                (void)hipEventRecord(e1), (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double insts_per_simd = (double)iters * 64 * wps * (mode == 7 ? 2 : 1);
            printf("  %-24s %6.2f\n", names[mode], ms * 1e-3 * 2.4e9 / insts_per_simd);
        }
    }
}
