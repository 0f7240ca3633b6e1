// What does a cross-stream wait cost the kernel BEHIND it?  (docs/LAB_NOTES_r03.md: kernels of the CarRacing step that follow a
// hipStreamWaitEvent run ~40 us longer than the same kernel alone.)
// Producer kernel on stream A dirties a large buffer; consumer kernel on stream B (latency-bound: every lane gathers from a 24 MB
// table it has read before) runs (a) behind nothing, (b) behind a wait on an event recorded after the producer, for several event
// flag combinations, on created streams and on the legacy default stream.  Consumer durations from its own hipEvents with
// hipEventDisableSystemFence (they bracket only the consumer).
//   hipcc --offload-arch=gfx950 -O3 tools/wait_cost_probe.hip -o tools/wait_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void producer(uint4 *buf, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = make_uint4(i, 1, 2, 3);
}
// 32 768 lanes, each follows a short chain through the table (like a SoA state kernel: dependent loads, little arithmetic)
__global__ void consumer(const uint32_t *__restrict__ table, uint32_t *out, uint32_t mask) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, acc = 0;
    for (int k = 0; k < 24; k++) {
        const uint32_t v = table[(i * 2654435761u + k * 40503u) & mask];
        acc += v, i = i * 31u + v;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const size_t big = (size_t)300 << 20;  // what the frame kernels leave behind
    uint4 *buf;
    uint32_t *table, *out;
    const uint32_t tn = 6u << 20;  // 24 MB of uint32
    hipMalloc(&buf, big), hipMalloc(&table, tn * 4), hipMalloc(&out, 32768 * 4);
    hipMemset(table, 1, tn * 4);
    hipStream_t A, B;
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking), hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
    hipEvent_t t0, t1;
    hipEventCreateWithFlags(&t0, hipEventDisableSystemFence), hipEventCreateWithFlags(&t1, hipEventDisableSystemFence);
    struct Case { const char *name; unsigned flags; bool wait; bool null_stream; bool produce; };
    const Case cases[] = {
        {"consumer alone (no producer)", 0, false, false, false},
        {"producer on A, consumer on B, no wait", 0, false, false, true},
        {"wait, default event", hipEventDefault, true, false, true},
        {"wait, DisableTiming", hipEventDisableTiming, true, false, true},
        {"wait, DisableTiming|DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence, true, false, true},
        {"wait, DisableTiming|ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice, true, false, true},
        {"wait, DisableTiming|DisableSystemFence|ReleaseToDevice", hipEventDisableTiming | hipEventDisableSystemFence | hipEventReleaseToDevice, true, false, true},
        {"consumer on the DEFAULT stream, wait, DisableTiming|DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence, true, true, true},
        {"consumer on the DEFAULT stream, no wait", 0, false, true, true},
    };
    for (const Case &c : cases) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, c.flags) != hipSuccess) {
            printf("%-78s (flag combination rejected)\n", c.name);
            (void)hipGetLastError();
            continue;
        }
        hipStream_t S = c.null_stream ? nullptr : B;
        std::vector<float> us, gap;
        for (int rep = 0; rep < 30; rep++) {
            hipLaunchKernelGGL(consumer, dim3(512), dim3(64), 0, S, table, out, tn - 1);  // warm the table
            hipDeviceSynchronize();
            if (c.produce) hipLaunchKernelGGL(producer, dim3(4096), dim3(256), 0, A, buf, big / 16);
            if (c.wait) {
                hipEventRecord(ev, A);
                hipStreamWaitEvent(S, ev, 0);
            } else {
                hipStreamSynchronize(A);  // same cache state, but no dependency in the consumer's queue
            }
            hipEventRecord(t0, S);
            hipLaunchKernelGGL(consumer, dim3(512), dim3(64), 0, S, table, out, tn - 1);
            hipEventRecord(t1, S);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, t0, t1);
            us.push_back(ms * 1e3f);
        }
        std::sort(us.begin(), us.end());
        printf("%-78s consumer (t0 -> t1): median %7.1f us  min %7.1f\n", c.name, us[us.size() / 2], us[0]);
        hipEventDestroy(ev);
    }
    return 0;
}
