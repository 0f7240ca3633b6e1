#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const f2* w, const f2* in, f2* out) {
    f2 ws = w[0];            // uniform -> sgpr pair
    f2 wv = w[threadIdx.x & 0];
    asm volatile("v_mov_b32 %0, %0" : "+v"(wv.x));  // force VGPR
    f2 x = in[threadIdx.x];
    f2 c = (f2)(1.f, 2.f);
    f2 r[6];
    for (int i = 0; i < 6; i++) r[i] = c;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[0]) : "v"(wv), "v"(x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(r[1]) : "v"(wv), "v"(x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(r[2]) : "v"(wv), "v"(x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[3]) : "s"(ws), "v"(x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(r[4]) : "s"(ws), "v"(x));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(r[5]) : "s"(ws), "v"(x));
    for (int i = 0; i < 6; i++) out[6 * threadIdx.x + i] = r[i];
}
int main() {
    f2 *w, *in, *out;
    (void)hipMallocManaged(&w, 8); (void)hipMallocManaged(&in, 64 * 8); (void)hipMallocManaged(&out, 6 * 64 * 8);
    w[0] = (f2){10.f, 100.f};
    for (int i = 0; i < 64; i++) in[i] = (f2){i + 1.f, -(i + 1.f) * 3};
    k<<<1, 64>>>(w, in, out);
    (void)hipDeviceSynchronize();
    const char* nm[6] = {"vgpr default", "vgpr bcast lo", "vgpr bcast hi", "sgpr default", "sgpr bcast lo", "sgpr bcast hi"};
    const float e[6][2] = {{11, -298}, {11, 102}, {-29, -298}, {11, -298}, {11, 102}, {-29, -298}};
    for (int i = 0; i < 6; i++) printf("%-14s got (%g,%g) expect (%g,%g)\n", nm[i], out[i].x, out[i].y, e[i][0], e[i][1]);
}
