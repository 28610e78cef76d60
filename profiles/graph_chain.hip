// graph_chain.hip -- what a HIP graph would buy the whole-domain step at small N ("capture what remains of a
// whole-domain step in a HIP graph").  A step at 131,072 ... 262,144 particles is a chain of 6 DEPENDENT kernels on one stream, each
// a few microseconds long; the question is whether replaying the chain as a graph shortens the time per kernel.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/graph_chain profiles/graph_chain.hip && scratch/graph_chain
// Three ways to run `steps` x 6 kernels over n floats (every kernel reads what the one before wrote: a true dependency chain):
//   eager   hipLaunchKernelGGL in a loop (what sph_step does)
//   graph6  one captured step (6 kernel nodes), hipGraphLaunch per step
//   graph60 ten captured steps (60 nodes), hipGraphLaunch per ten steps
// and, to separate the host's launch cost from the device's boundary cost, the host time spent queueing each variant.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out, unsigned n, float a) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = in[i] * a + 1.0f;
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000, K = 6;
    hipStream_t s; CK(hipStreamCreate(&s));
    for (unsigned n : {1024u, 131072u, 262144u, 2097152u}) {
        float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
        const dim3 grid((n + 255) / 256), block(256);
        auto step = [&]() { for (int k = 0; k < K; k++) hipLaunchKernelGGL(k_touch, grid, block, 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n, 0.5f); };
        for (int w = 0; w < 50; w++) step();
        CK(hipStreamSynchronize(s));
        double t0 = now(); for (int i = 0; i < steps; i++) step(); double tq = now() - t0; CK(hipStreamSynchronize(s)); const double eager = now() - t0;
        double res[2] = {0, 0}, resq[2] = {0, 0};
        int v = 0;
        for (int per : {1, 10}) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int p = 0; p < per; p++) step();
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 20; w++) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            t0 = now(); for (int i = 0; i < steps / per; i++) CK(hipGraphLaunch(ge, s)); resq[v] = now() - t0; CK(hipStreamSynchronize(s)); res[v] = now() - t0;
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            v++;
        }
        printf("n %8u (%5u blocks): us per kernel  eager %.2f (host queueing %.2f)   graph of 6 nodes %.2f (host %.2f)   graph of 60 nodes %.2f (host %.2f)\n",
               n, grid.x, eager / (steps * K), tq / (steps * K), res[0] / (steps * K), resq[0] / (steps * K), res[1] / (steps * K), resq[1] / (steps * K));
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
