// Micro-benchmark: the inner loop of the FORCE pass in fp32 (the product's instruction sequence) against the same pair
// arithmetic in PACKED fp16 -- two candidates per lane-instruction, BASELINE config 5's "fp16 neighbour accumulators" --
// on the bare hardware: a wave walks rows of 24 candidates out of a wave-private LDS slice (per-lane start entries in
// the pattern of the pair kernels: 8 lanes per cell, cells 8 entries apart), 4 waves per block, W blocks per CU.
// Decides whether a packed-fp16 force kernel is worth building.  No global traffic in the loop.
//
//   fp32 : candidate = 4 ds_read_b64 {x,y} {z,vx} {vy,vz} {cp,w} at a 40-byte stride; 23 VALU + v_rsq_f32
//   fp16 : PAIR of candidates = 4 ds_read_b64 {x01,y01} {z01,vx01} {vy01,vz01} {cp01,w01} (halves), 32-byte... 40-byte
//          stride as well; ~21 v_pk_* + 2 v_rsq_f16 + a pack; row sums go to fp32 once per row (as k_density_h does)
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize profiles/force_mix_f16.hip -o /tmp/force_mix && /tmp/force_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;

constexpr int ROW = 24;          // candidates per row walk (a lane's three cells at rest density)
constexpr int ROWS = 9;
constexpr int SLICE = 160;       // entries per wave slice

__device__ __forceinline__ float in_vgpr(float x) { asm volatile("" : "+v"(x)); return x; }

__global__ __launch_bounds__(256, 5) void k_f32(const int* __restrict__ entry, int iters, float* out) {
    __shared__ float2 s_e[4 * SLICE * 5 + 64];
    for (int i = threadIdx.x; i < 4 * SLICE * 5 + 64; i += 256) s_e[i] = make_float2(0.01f * (float)(i % 97), 0.02f * (float)(i % 89));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 pi = make_float4(0.3f + 0.001f * lane, 0.4f, 0.5f, 0.f);
    const float cpi = in_vgpr(1.5f), h_v = in_vgpr(0.9f), cn = in_vgpr(0.01f);
    float fpx = 0, fpy = 0, fpz = 0, fvx = 0, fvy = 0, fvz = 0, sw = 0;
    unsigned near = 0;
    for (int it = 0; it < iters; it++)
        for (int r = 0; r < ROWS; r++) {
            unsigned idx = wave * SLICE + entry[lane] + r;
#pragma unroll 2
            for (int t = 0; t < ROW; t++) {
                const lds_v2f_ptr e = (lds_v2f_ptr)s_e + (idx + t) * 5;
                const v2f qa = e[0], qb = e[1];
                const float dx = pi.x - qa.x, dy = pi.y - qa.y, dz = pi.z - qb.x;
                const float r2 = fmaf(dz, dz, fmaf(dx, dx, fmaf(dy, dy, 1e-30f)));
                const v2f qc = e[2], qd = e[3];
                const float rinv = __builtin_amdgcn_rsqf(r2);
                const float hr = fmaxf(fmaf(-r2, rinv, h_v), 0.f);
                const float w = qd.y * hr;
                const float s = (cpi + qd.x) * w * (hr * rinv);
                fpx += s * dx; fpy += s * dy; fpz += s * dz;
                fvx += w * qb.y; fvy += w * qc.x; fvz += w * qc.y;
                sw += w;
                near = __builtin_amdgcn_alignbit(near, __float_as_uint(r2 - cn), 31);
            }
        }
    out[blockIdx.x * 256 + threadIdx.x] = fpx + fpy + fpz + fvx + fvy + fvz + sw + (float)near;
}

__device__ __forceinline__ h2 h2s(float x) { const _Float16 v = (_Float16)x; return h2{v, v}; }

// PAIRS: 12 pairs per row; entry p of a slice holds candidates 2p, 2p+1 as halves
__global__ __launch_bounds__(256, 5) void k_f16(const int* __restrict__ entry, int iters, float* out) {
    struct P { h2 a, b; };
    __shared__ P s_e[4 * SLICE * 5 + 64];
    for (int i = threadIdx.x; i < 4 * SLICE * 5 + 64; i += 256) {
        const _Float16 u = (_Float16)(0.01f * (float)(i % 97)), v = (_Float16)(0.02f * (float)(i % 89));
        s_e[i].a = h2{u, v}; s_e[i].b = h2{v, u};
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const h2 tx = h2s(0.3f + 0.001f * lane), ty = h2s(0.4f), tz = h2s(0.5f);
    const h2 cpi = h2s(1.5f), one = h2s(1.0f), zero = h2s(0.f), eps = h2s(6e-8f), cn = h2s(0.01f);
    float fpx = 0, fpy = 0, fpz = 0, fvx = 0, fvy = 0, fvz = 0, sw = 0;
    unsigned near = 0;
    typedef const volatile __attribute__((address_space(3))) v2f* lp;
    for (int it = 0; it < iters; it++)
        for (int r = 0; r < ROWS; r++) {
            unsigned idx = wave * SLICE + entry[lane] / 2 + r;
            h2 rpx = zero, rpy = zero, rpz = zero, rvx = zero, rvy = zero, rvz = zero, rsw = zero;
#pragma unroll 2
            for (int t = 0; t < ROW / 2; t++) {
                const lp e = (lp)s_e + (idx + t) * 5;
                const v2f qa = e[0], qb = e[1];                   // raw 64-bit reads, reinterpreted as two h2 each
                const h2 X = __builtin_bit_cast(h2, qa.x), Y = __builtin_bit_cast(h2, qa.y);
                const h2 Z = __builtin_bit_cast(h2, qb.x), VX = __builtin_bit_cast(h2, qb.y);
                const h2 dx = tx - X, dy = ty - Y, dz = tz - Z;
                h2 r2 = dy * dy + eps;
                r2 = dx * dx + r2;
                r2 = dz * dz + r2;
                const v2f qc = e[2], qd = e[3];
                const h2 VY = __builtin_bit_cast(h2, qc.x), VZ = __builtin_bit_cast(h2, qc.y);
                const h2 CP = __builtin_bit_cast(h2, qd.x), W = __builtin_bit_cast(h2, qd.y);
                const h2 rinv = h2{(_Float16)__builtin_amdgcn_rsqh((__fp16)r2.x), (_Float16)__builtin_amdgcn_rsqh((__fp16)r2.y)};
                h2 hr = one - r2 * rinv;
                hr = __builtin_elementwise_max(hr, zero);
                const h2 w = W * hr;
                const h2 s = (cpi + CP) * w * (hr * rinv);
                rpx = s * dx + rpx; rpy = s * dy + rpy; rpz = s * dz + rpz;
                rvx = w * VX + rvx; rvy = w * VY + rvy; rvz = w * VZ + rvz;
                rsw = rsw + w;
                const h2 c = r2 - cn;                              // two collision bits per pair: sign bits of the halves
                const unsigned cb = __builtin_bit_cast(unsigned, c);
                near = (near << 2) | ((cb >> 30) & 2u) | ((cb >> 15) & 1u);
            }
            fpx += (float)rpx.x + (float)rpx.y; fpy += (float)rpy.x + (float)rpy.y; fpz += (float)rpz.x + (float)rpz.y;
            fvx += (float)rvx.x + (float)rvx.y; fvy += (float)rvy.x + (float)rvy.y; fvz += (float)rvz.x + (float)rvz.y;
            sw += (float)rsw.x + (float)rsw.y;
        }
    out[blockIdx.x * 256 + threadIdx.x] = fpx + fpy + fpz + fvx + fvy + fvz + sw + (float)near;
}

template <class K>
void run(const char* name, K kern, const int* d_entry, int cand_per_iter) {
    for (int w : {3, 5}) {
        const int blocks = 256 * w, iters = 40;
        float* out; hipMalloc(&out, blocks * 256 * sizeof(float));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_entry, 2, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_entry, iters, out);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: w waves, each iters * cand_per_iter candidates
        printf("%-52s : %6.2f ns per candidate per SIMD at %d blocks per CU\n", name, ms * 1e6 / ((double)iters * cand_per_iter * w), w);
        hipFree(out);
    }
}

int main() {
    std::vector<int> e(64);
    for (int l = 0; l < 64; l++) e[l] = (l / 8) * 8;               // 8 lanes per cell, cells 8 entries apart
    int* d_entry; hipMalloc(&d_entry, 64 * sizeof(int));
    hipMemcpy(d_entry, e.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    run("fp32 force pair loop (product sequence)", k_f32, d_entry, ROWS * ROW);
    run("packed-fp16 force pair loop (2 candidates/instr)", k_f16, d_entry, ROWS * ROW);
    return 0;
}
