// Micro-benchmark: cost of per-lane LDS reads on gfx950 for the address patterns of the pair kernels.
// A wave reads `iters` x 16 times from its slice; lane l reads entry e(l) + k (k = 0..15), entry stride S dwords.
// e(l) comes from the host: lanes of one cell share a start, cells are n(c) entries apart.
// Reported: LDS-pipe time per wave-instruction per CU (all 4 SIMDs busy, W waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x

// MODE 0: ds_read_b64 (8-byte aligned)   1: ds_read2_b32 off, off+1 (4-byte aligned)   2: two ds_read_b32
// MODE 3: ds_read_b32 only               4: ds_read_b128 (16-byte aligned)             5: ds_read_b96
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ entry, int stride, int iters, float* out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)((wave * 150 + entry[lane]) * stride * 4);   // bytes
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const unsigned sb = (unsigned)stride * 4u;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int it = 0; it < iters; it++) {
        unsigned a0 = addr, a1 = addr + sb, a2 = addr + 2 * sb, a3 = addr + 3 * sb;
        for (int k = 0; k < 16; k += 4) {          // four reads in flight per wave, then one wait
            if (MODE == 0) {
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            } else if (MODE == 1) {
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %5 offset1:1\n ds_read2_b32 %2, %6 offset1:1\n ds_read2_b32 %3, %7 offset1:1\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            } else if (MODE == 2) {
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %5\n ds_read_b32 %3, %5 offset:4\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(acc0), "=&v"(acc1), "=&v"(acc2), "=&v"(acc3) : "v"(a0), "v"(a1) : "memory");
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %5\n ds_read_b32 %3, %5 offset:4\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(acc0), "=&v"(acc1), "=&v"(acc2), "=&v"(acc3) : "v"(a2), "v"(a3) : "memory");
            } else if (MODE == 4) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 q0, q1, q2, q3;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
                acc0 += q0.x + q1.y + q2.z + q3.w;
            } else if (MODE == 5) {
                typedef float f3 __attribute__((ext_vector_type(3)));
                f3 q0, q1, q2, q3;
                asm volatile("ds_read_b96 %0, %4\n ds_read_b96 %1, %5\n ds_read_b96 %2, %6\n ds_read_b96 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
                acc0 += q0.x + q1.y + q2.z + q3.x;
            } else if (MODE == 3) {
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(acc0), "=&v"(acc1), "=&v"(acc2), "=&v"(acc3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            }
            a0 += 4 * sb; a1 += 4 * sb; a2 += 4 * sb; a3 += 4 * sb;
        }
    }
    acc0 += (float)(d0 + d1 + d2 + d3);
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}

// The pair kernels' mix: R ds_read_b64 per candidate + V fp32 VALU instructions that use what was read.
// READS = 0: the arithmetic alone (operands stay in registers); VALU = 0: the reads alone.
template <int READS, int VALU>
__global__ __launch_bounds__(256) void kmix(const int* __restrict__ entry, int stride, int iters, float* out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4864; i += 256) lds[i] = 1.0f + (float)i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned addr = (unsigned)((wave * 100 + entry[lane]) * stride * 4);
    const unsigned sb = (unsigned)stride * 4u;
    float a0 = 1.f + lane, a1 = 2.f, a2 = 3.f, a3 = 4.f, a4 = 5.f, a5 = 6.f;
    double d0 = 1.0, d1 = 2.0, d2 = 3.0, d3 = 4.0;
    for (int it = 0; it < iters; it++) {
        unsigned a = addr;
        for (int k = 0; k < 16; k++) {              // 16 candidates per row walk
            if (READS == 4)
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a) : "memory");
            if (READS == 42) {     // round 5: the force candidate's 32 bytes as TWO aligned ds_read_b128 (stride must be a multiple of 4 dwords)
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 qa, qb;
                asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(qa), "=&v"(qb) : "v"(a & ~15u) : "memory");
                d0 = ((double*)&qa)[0]; d1 = ((double*)&qa)[1]; d2 = ((double*)&qb)[0]; d3 = ((double*)&qb)[1];
            }
            if (READS == 2)
                asm volatile("ds_read_b64 %0, %2\n ds_read_b32 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(d0), "=&v"(a5) : "v"(a) : "memory");
            if (READS == 22)       // two density candidates: b64 + b32 each
                asm volatile("ds_read_b64 %0, %3\n ds_read_b32 %2, %3 offset:8\n ds_read_b64 %1, %3 offset:40\n ds_read_b32 %2, %3 offset:48\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(d0), "=&v"(d1), "=&v"(a5) : "v"(a) : "memory");
            if (READS == 21) {     // two density candidates in one b128 ({x0,y0,x1,y1}) + one b64 ({z0,z1}); 16-byte aligned
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 q;
                asm volatile("ds_read_b128 %0, %2\n ds_read_b64 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(q), "=&v"(d1) : "v"(a & ~15u) : "memory");
                d0 = *(double*)&q;
            }
            a += sb;
            const float x0 = ((float*)&d0)[0], x1 = ((float*)&d0)[1], x2 = ((float*)&d1)[0], x3 = ((float*)&d1)[1];
#pragma unroll
            for (int v = 0; v < VALU / 6; v++) {    // six independent fp32 fmas per round, fed by the loaded values
                asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %7, %8\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %9, %6\n v_fmac_f32 %4, %6, %8\n v_fmac_f32 %5, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + (float)(d2 + d3);
}

// software-pipelined form of the force-like mix: the reads of candidate k+1 are in flight while candidate k's
// arithmetic runs (two register sets, loop unrolled by two; s_waitcnt lgkmcnt(4) = "all but the newest four")
__global__ __launch_bounds__(256) void kpipe(const int* __restrict__ entry, int stride, int iters, float* out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4864; i += 256) lds[i] = 1.0f + (float)i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned addr = (unsigned)((wave * 100 + entry[lane]) * stride * 4);
    const unsigned sb = (unsigned)stride * 4u;
    float a0 = 1.f + lane, a1 = 2.f, a2 = 3.f, a3 = 4.f, a4 = 5.f, a5 = 6.f;
    double p0, p1, p2, p3, q0, q1, q2, q3;
#define ISSUE(r0, r1, r2, r3, ad) asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n" \
                                               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(ad) : "memory")
#define WORK(r0, r1) do { asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                                  \
        const float x0 = ((float*)&r0)[0], x1 = ((float*)&r0)[1], x2 = ((float*)&r1)[0], x3 = ((float*)&r1)[1];          \
        _Pragma("unroll") for (int v = 0; v < 4; v++)                                                                          \
            asm volatile("v_fmac_f32 %0, %6, %7\n v_fmac_f32 %1, %7, %8\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %9, %6\n v_fmac_f32 %4, %6, %8\n v_fmac_f32 %5, %7, %9\n" \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(x0), "v"(x1), "v"(x2), "v"(x3)); } while (0)
    for (int it = 0; it < iters; it++) {
        unsigned a = addr;
        ISSUE(p0, p1, p2, p3, a); a += sb;
        for (int k = 0; k < 16; k += 2) {
            ISSUE(q0, q1, q2, q3, a); a += sb;
            WORK(p0, p1);
            ISSUE(p0, p1, p2, p3, a); a += sb;      // (one candidate past the end on the last round: still inside the slice)
            WORK(q0, q1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef ISSUE
#undef WORK
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + (float)(p2 + p3 + q2 + q3);
}

template <int READS, int VALU>
void runmix(const char* name, const std::vector<int>& entry) {
    int* d_entry; float* out;
    hipMalloc(&d_entry, 64 * sizeof(int));
    hipMemcpy(d_entry, entry.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 300;
    for (int w : {3, 5, 8}) {                   // blocks of 4 waves per CU = waves per SIMD (k_force runs 5, k_density 6)
        const int blocks = 256 * w;
        hipMalloc(&out, blocks * 256 * sizeof(float));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((kmix<READS, VALU>), dim3(blocks), dim3(256), 19456, 0, d_entry, 10, 4, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((kmix<READS, VALU>), dim3(blocks), dim3(256), 19456, 0, d_entry, 10, iters, out);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s : %6.2f ns per candidate per SIMD at %d waves per SIMD\n", name, ms * 1e6 / ((double)iters * 16 * w), w);
        hipFree(out);
    }
    hipFree(d_entry);
}

template <int READS, int VALU>
void runmix_stride(const char* name, const std::vector<int>& entry, int stride) {
    int* d_entry; float* out;
    hipMalloc(&d_entry, 64 * sizeof(int));
    hipMemcpy(d_entry, entry.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 300;
    for (int w : {3, 5, 8}) {                   // blocks of 4 waves per CU = waves per SIMD (k_force runs 5, k_density 6)
        const int blocks = 256 * w;
        hipMalloc(&out, blocks * 256 * sizeof(float));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((kmix<READS, VALU>), dim3(blocks), dim3(256), 19456, 0, d_entry, stride, 4, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((kmix<READS, VALU>), dim3(blocks), dim3(256), 19456, 0, d_entry, stride, iters, out);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s : %6.2f ns per candidate per SIMD at %d waves per SIMD\n", name, ms * 1e6 / ((double)iters * 16 * w), w);
        hipFree(out);
    }
    hipFree(d_entry);
}

// The s_waitcnt after every read would measure latency, not throughput, for one wave; with W >= 4 waves per SIMD the
// LDS pipe is kept full by the other waves and the wall time is the pipe's.
template <int MODE>
void run(const char* name, const std::vector<int>& entry, int stride, const char* pat) {
    int* d_entry; float* out;
    hipMalloc(&d_entry, 64 * sizeof(int));
    hipMemcpy(d_entry, entry.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 400;
    for (int w : {3, 4}) {                     // blocks per CU (4 waves each): 3-4 waves per SIMD, 32 KB LDS per block
        const int blocks = 256 * w;
        hipMalloc(&out, blocks * 256 * sizeof(float));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, d_entry, stride, 4, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 32768, 0, d_entry, stride, iters, out);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double reads_per_cu = (double)iters * 16 * 4 * w;     // wave-level read steps per CU
        printf("%-28s stride %2d  %-26s blocks/CU %d : %6.2f ns per wave-read per CU\n", name, stride, pat, w, ms * 1e6 / reads_per_cu);
        hipFree(out);
    }
    hipFree(d_entry);
}

static std::vector<int> cells(const std::vector<int>& lanes_per_cell, const std::vector<int>& count) {
    // lanes_per_cell[c] lanes read from the start of cell c; cell c+1 starts count[c] entries later
    std::vector<int> e;
    int start = 0;
    for (size_t c = 0; c < lanes_per_cell.size() && e.size() < 64; c++) {
        for (int l = 0; l < lanes_per_cell[c] && e.size() < 64; l++) e.push_back(start);
        start += count[c];
    }
    while (e.size() < 64) e.push_back(start);
    return e;
}

int main(int argc, char** argv) {
    const std::vector<int> eight(8, 8);
    const auto rest = cells(eight, eight);                                         // 8 lanes per cell, cells 8 entries apart
    if (argc > 1 && std::string(argv[1]) == "b128") {
        // round 5: would the force candidate's four ds_read_b64 be cheaper as two ds_read_b128?  Strides of 8 and 12 dwords (32 / 48 bytes),
        // lanes as at rest (8 per cell) and in a flow (9 per cell, random cells)
        const auto flow2 = cells({3, 9, 9, 9, 9, 9, 9, 7}, {9, 9, 9, 9, 9, 9, 9, 9});
        runmix<4, 24>("4 ds_read_b64 + 24 VALU, stride 10, rest", rest);
        runmix<4, 24>("4 ds_read_b64 + 24 VALU, stride 10, flow (9 per cell)", flow2);
        runmix_stride<42, 24>("2 ds_read_b128 + 24 VALU, stride 8, rest", rest, 8);
        runmix_stride<42, 24>("2 ds_read_b128 + 24 VALU, stride 8, flow (9 per cell)", flow2, 8);
        runmix_stride<42, 24>("2 ds_read_b128 + 24 VALU, stride 12, rest", rest, 12);
        runmix_stride<42, 24>("2 ds_read_b128 + 24 VALU, stride 12, flow (9 per cell)", flow2, 12);
        runmix_stride<42, 0>("2 ds_read_b128, no VALU, stride 8, rest", rest, 8);
        runmix_stride<42, 0>("2 ds_read_b128, no VALU, stride 8, flow", flow2, 8);
        runmix<4, 0>("4 ds_read_b64, no VALU, stride 10, rest", rest);
        runmix<0, 24>("24 VALU, no LDS", rest);
        return 0;
    }
    if (argc > 1 && std::string(argv[1]) == "two") {
        // round 4: what would two targets per lane buy k_force?  The candidate's four reads feed 48 VALU instead of 24.
        runmix<4, 24>("4 ds_read_b64 + 24 VALU (one target)", rest);
        runmix<4, 48>("4 ds_read_b64 + 48 VALU (two targets: per PAIR)", rest);
        runmix<0, 24>("24 VALU, no LDS", rest);
        runmix<0, 48>("48 VALU, no LDS", rest);
        return 0;
    }
    runmix<0, 24>("24 VALU, no LDS", rest);
    runmix<4, 0>("4 ds_read_b64, no VALU", rest);
    runmix<4, 24>("4 ds_read_b64 + 24 VALU (force-like)", rest);
    runmix<0, 12>("12 VALU, no LDS", rest);
    runmix<2, 0>("b64 + b32, no VALU", rest);
    runmix<2, 12>("b64 + b32 + 12 VALU (density-like)", rest);
    {
        int* d_entry; float* out;
        hipMalloc(&d_entry, 64 * sizeof(int));
        hipMemcpy(d_entry, rest.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
        for (int w : {3, 5, 8}) {
            const int blocks = 256 * w, iters = 300;
            hipMalloc(&out, blocks * 256 * sizeof(float));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kpipe, dim3(blocks), dim3(256), 19456, 0, d_entry, 10, 4, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kpipe, dim3(blocks), dim3(256), 19456, 0, d_entry, 10, iters, out);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-44s : %6.2f ns per candidate per SIMD at %d waves per SIMD\n", "4 ds_read_b64 + 24 VALU, software-pipelined", ms * 1e6 / ((double)iters * 16 * w), w);
            hipFree(out);
        }
        hipFree(d_entry);
    }
    runmix<22, 18>("2 x (b64 + b32) + 18 VALU (2 density cand.)", rest);
    runmix<21, 18>("b128 + b64 + 18 VALU (2 density cand.)", rest);
    runmix<21, 0>("b128 + b64, no VALU", rest);
    runmix<22, 0>("2 x (b64 + b32), no VALU", rest);
    return 0;
    const auto flow = cells({5, 9, 7, 10, 8, 9, 7, 9}, {7, 9, 7, 10, 8, 9, 7, 9});  // cells c, c+2 are 16 apart inside a 16-lane group
    const auto flow2 = cells({3, 9, 9, 9, 9, 9, 9, 7}, {9, 9, 9, 9, 9, 9, 9, 9});   // 9 per cell: no pair is 16 or 32 apart
    for (int stride : {10, 11}) {
        if (stride % 2 == 0) {
            run<0>("ds_read_b64", rest, stride, "rest (8 per cell)");
            run<0>("ds_read_b64", flow, stride, "flow (7+9 = 16 apart)");
            run<0>("ds_read_b64", flow2, stride, "flow (9 per cell)");
        }
        run<1>("ds_read2_b32 off,off+1", rest, stride, "rest (8 per cell)");
        run<1>("ds_read2_b32 off,off+1", flow, stride, "flow (7+9 = 16 apart)");
        run<1>("ds_read2_b32 off,off+1", flow2, stride, "flow (9 per cell)");
        run<2>("2 x ds_read_b32", rest, stride, "rest (8 per cell)");
        run<2>("2 x ds_read_b32", flow, stride, "flow (7+9 = 16 apart)");
    }
    {   // cells of a flowing dam: 5..12 particles per cell (mean 8.7); targets and candidates are different cells
        unsigned seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (int)(5 + ((seed >> 16) % 8)); };
        for (int t = 0; t < 6; t++) {
            std::vector<int> lanes, cnt;
            for (int c = 0; c < 16; c++) { lanes.push_back(rnd()); cnt.push_back(rnd()); }
            char nm[64]; snprintf(nm, sizeof nm, "random cells #%d", t);
            run<0>("ds_read_b64", cells(lanes, cnt), 10, nm);
            run<0>("ds_read_b64", cells(lanes, cnt), 2, nm);
            run<3>("ds_read_b32", cells(lanes, cnt), 1, nm);
        }
        // 5 cells inside a half-wave, first and last exactly 32 entries apart
        run<0>("ds_read_b64", cells({4, 7, 7, 7, 7, 8, 8, 8, 8}, {8, 8, 8, 8, 8, 8, 8, 8, 8}), 10, "cells c, c+4 32 apart");
        run<0>("ds_read_b64", cells({4, 7, 7, 7, 7, 8, 8, 8, 8}, {8, 8, 8, 8, 8, 8, 8, 8, 8}), 2, "cells c, c+4 32 apart");
    }
    for (int stride : {4, 12, 3, 5, 10}) {           // 3, 5, 10: not 16-byte aligned, 20+ ns per read
        run<5>("ds_read_b96", rest, stride, "rest (8 per cell)");
        run<5>("ds_read_b96", flow2, stride, "flow (9 per cell)");
        run<4>("ds_read_b128", rest, stride, "rest (8 per cell)");
        run<4>("ds_read_b128", flow2, stride, "flow (9 per cell)");
    }
    run<3>("ds_read_b32", rest, 3, "rest (8 per cell)");
    run<3>("ds_read_b32", flow, 3, "flow (7+9 = 16 apart)");
    run<3>("ds_read_b32", rest, 1, "rest (8 per cell)");
    run<3>("ds_read_b32", flow, 1, "flow (7+9 = 16 apart)");
    run<1>("ds_read2_b32 off,off+1", rest, 3, "rest (8 per cell)");
    run<1>("ds_read2_b32 off,off+1", flow, 3, "flow (7+9 = 16 apart)");
    run<0>("ds_read_b64", rest, 2, "rest (8 per cell)");
    run<0>("ds_read_b64", flow, 2, "flow (7+9 = 16 apart)");
    return 0;
}
