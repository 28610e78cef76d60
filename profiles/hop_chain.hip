// hop_chain.hip -- what a cross-stream dependency costs on MI355X: a chain of tiny kernels alternating between two streams,
// ordered by (a) hipEventRecord + hipStreamWaitEvent, (b) hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory,
// against (c) the same kernels on ONE stream.       hipcc --offload-arch=gfx950 -O2 -o /tmp/hop_chain profiles/hop_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int hops = argc > 1 ? atoi(argv[1]) : 400, n = argc > 2 ? atoi(argv[2]) : 65536;
    float* buf; CK(hipMalloc(&buf, n * sizeof(float))); CK(hipMemset(buf, 0, n * sizeof(float)));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    uint32_t* sig = nullptr; const char* kind = "signal memory";
    hipError_t se = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
    if (se != hipSuccess) {                                   // (this pool: "invalid argument") -- plain device memory, then pinned host memory
        fprintf(stderr, "hipMallocSignalMemory: %s (hipDeviceAttributeCanUseStreamWaitValue = %d)\n", hipGetErrorString(se), can);
        (void)hipGetLastError();
        kind = "device memory";
        if (hipMalloc((void**)&sig, 64) != hipSuccess) sig = nullptr;
    }
    if (sig) {                                                 // does the pair work on this memory at all?
        (void)hipMemset(sig, 0, 64);
        hipError_t w = hipStreamWriteValue32(s[0], sig, 1u, 0);
        hipError_t q = w == hipSuccess ? hipStreamWaitValue32(s[1], sig, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu) : w;
        if (q != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            fprintf(stderr, "write / wait value32 on %s: %s\n", kind, hipGetErrorString(q)); (void)hipGetLastError();
            (void)hipFree(sig); sig = nullptr; kind = "pinned host memory";
            if (hipHostMalloc((void**)&sig, 64, hipHostMallocCoherent) == hipSuccess) {
                *sig = 0;
                w = hipStreamWriteValue32(s[0], sig, 1u, 0);
                q = w == hipSuccess ? hipStreamWaitValue32(s[1], sig, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu) : w;
                if (q != hipSuccess || hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "... on %s: %s\n", kind, hipGetErrorString(q)); (void)hipGetLastError(); sig = nullptr; }
            } else sig = nullptr;
        }
        if (sig) fprintf(stderr, "write / wait value32 works on %s\n", kind);
    }
    dim3 g((n + 255) / 256), b(256);
    for (int rep = 0; rep < 3; rep++) {
        // (c) one stream
        CK(hipDeviceSynchronize());
        double t0 = now();
        for (int h = 0; h < hops; h++) hipLaunchKernelGGL(k_touch, g, b, 0, s[0], buf, n);
        CK(hipDeviceSynchronize());
        const double one = (now() - t0) / hops * 1e6;
        // (a) events
        t0 = now();
        for (int h = 0; h < hops; h++) {
            const int a = h & 1, o = a ^ 1;
            hipLaunchKernelGGL(k_touch, g, b, 0, s[a], buf, n);
            CK(hipEventRecord(ev[a], s[a]));
            CK(hipStreamWaitEvent(s[o], ev[a], 0));
        }
        CK(hipDeviceSynchronize());
        const double evt = (now() - t0) / hops * 1e6;
        double val = -1;
        if (sig) {
            CK(hipDeviceSynchronize()); const uint32_t base = 16u + (uint32_t)rep * (uint32_t)hops;
            t0 = now();
            for (int h = 0; h < hops; h++) {
                const int a = h & 1, o = a ^ 1;
                hipLaunchKernelGGL(k_touch, g, b, 0, s[a], buf, n);
                CK(hipStreamWriteValue32(s[a], sig, base + (uint32_t)h, 0));
                CK(hipStreamWaitValue32(s[o], sig, base + (uint32_t)h, hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
            CK(hipDeviceSynchronize());
            val = (now() - t0) / hops * 1e6;
        }
        printf("n = %d floats, %d kernels: one stream %.2f us per kernel | two streams, event record + wait %.2f | write / wait value32 %.2f\n", n, hops, one, evt, val);
    }
    return 0;
}
