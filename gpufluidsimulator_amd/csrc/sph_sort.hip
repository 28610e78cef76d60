// sph_sort.hip -- cell hash, on-device LSD radix sort of (cell key, slot) pairs, reorder.
//
// Replaces cudaMapZIndex + cudaSortParticles (particleSystem.cu:491-501): the reference
// thrust::sort()s the whole 88-byte AoS array with a comparator (a merge sort moving 88 B per
// element per level, 90 % of its step at 131k particles).  Here only 8-byte (key, slot) pairs are
// sorted, 8 bits per pass over the significant key bits, and the SoA payload is gathered once.
// The sort is stable, so the order of particles inside a cell is deterministic.
#include "sph_device.hpp"

namespace sph {

constexpr int SORT_THREADS = 256;            // 4 waves
constexpr int SORT_KPT = 16;                 // keys per thread
constexpr int SORT_WAVE_TILE = WAVE * SORT_KPT;          // 1024 consecutive keys per wave
constexpr int SORT_TILE = SORT_THREADS * SORT_KPT;       // 4096 keys per block

// ---- hash: key per owned particle ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash(const float4* __restrict__ posi, uint32_t n, GridDesc g,
                                              uint32_t* __restrict__ keys) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = posi[i];
    keys[i] = cell_key(g, p.x, p.y, p.z);
}

// ---- pass 1 of 3: per-block digit histogram ----------------------------------------------------
// BITS = 8 or 9 bits per pass: 27 significant key bits (512^3 cells) sort in 3 passes of 9.
template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint32_t* __restrict__ keys, uint32_t n,
                                                            uint32_t shift, uint32_t nblocks,
                                                            uint32_t* __restrict__ hist) {
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t h[RADIX];
    for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) h[d] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (int t = 0; t < SORT_KPT; t++) {
        uint32_t i = base + t * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) hist[(size_t)d * nblocks + blockIdx.x] = h[d];   // digit-major
}

// ---- pass 2 of 3: one block per digit scans its row of per-block counts --------------------------
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t* __restrict__ hist, uint32_t nblocks,
                                                   uint32_t* __restrict__ digit_tot) {
    __shared__ uint32_t part[256];
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t per = (nblocks + 255u) / 256u;
    uint32_t lo = min(threadIdx.x * per, nblocks);
    uint32_t hi = min(lo + per, nblocks);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 partial sums
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = threadIdx.x >= (uint32_t)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;   // exclusive
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    if (threadIdx.x == 255) digit_tot[blockIdx.x] = part[255];
}

// ---- pass 3 of 3: stable scatter -------------------------------------------------------------------
// Each wave owns 1024 consecutive keys of the block's tile and walks them 64 at a time in order.
// Rank of a key among equal digits inside one 64-key row: ballot-based match-any + popcount of the
// lower lanes; running per-wave, per-digit counters live in LDS.
template <int BITS, bool FIRST>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const uint32_t* __restrict__ kin,
                                                               const uint32_t* __restrict__ vin,
                                                               uint32_t* __restrict__ kout,
                                                               uint32_t* __restrict__ vout, uint32_t n,
                                                               uint32_t shift, uint32_t nblocks,
                                                               const uint32_t* __restrict__ hist,
                                                               const uint32_t* __restrict__ digit_tot) {
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX / SORT_THREADS;       // digits per thread: 1 or 2 (consecutive digits)
    __shared__ uint32_t wh[4][RADIX];     // per-wave digit counters -> running offsets
    __shared__ uint32_t part[SORT_THREADS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (int w = 0; w < 4; w++)
        for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) wh[w][d] = 0;
    // exclusive scan of the digit totals (global base of every digit): thread t owns digits t*DPT ..
    uint32_t tot[DPT], sum = 0;
#pragma unroll
    for (int k = 0; k < DPT; k++) { tot[k] = digit_tot[threadIdx.x * DPT + k]; sum += tot[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < SORT_THREADS; off <<= 1) {
        uint32_t v = threadIdx.x >= (uint32_t)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t my_base[DPT];
    {
        uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
        for (int k = 0; k < DPT; k++) {
            my_base[k] = run + hist[(size_t)(threadIdx.x * DPT + k) * nblocks + blockIdx.x];
            run += tot[k];
        }
    }

    // load this wave's keys (registers) and count digits per wave
    const uint32_t wbase = blockIdx.x * SORT_TILE + wave * SORT_WAVE_TILE;
    uint32_t key[SORT_KPT];
#pragma unroll
    for (int t = 0; t < SORT_KPT; t++) {
        uint32_t i = wbase + t * WAVE + lane;
        key[t] = i < n ? kin[i] : 0xFFFFFFFFu;
        if (i < n) atomicAdd(&wh[wave][(key[t] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    // per digit: exclusive scan over the 4 waves, plus the global base
#pragma unroll
    for (int k = 0; k < DPT; k++) {
        const uint32_t d = threadIdx.x * DPT + k;
        uint32_t o = my_base[k];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t cnt = wh[w][d];
            wh[w][d] = o;
            o += cnt;
        }
    }
    __syncthreads();

    volatile uint32_t* cnt = wh[wave];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int t = 0; t < SORT_KPT; t++) {
        uint32_t i = wbase + t * WAVE + lane;
        bool valid = i < n;
        uint32_t d = (key[t] >> shift) & (RADIX - 1);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            bool bit = (d >> b) & 1u;
            uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        uint32_t base = 0;
        if (valid) base = cnt[d];                       // every peer reads the same word
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) cnt[d] = base + (uint32_t)__popcll(peers);   // LDS is in order per wave
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            uint32_t dst = base + rank;
            kout[dst] = key[t];
            vout[dst] = FIRST ? i : vin[i];
        }
    }
}

// ---- reorder: gather the SoA payload into sorted order; optionally build the cell table ------------
// CELLS: the {start, end} entries of kernelConstructBGrid (.cu:311-329) fall out of the same pass
// (boundary flags on the sorted keys, no atomics) when no ghost layers will be added afterwards.
template <bool CELLS>
__global__ __launch_bounds__(256) void k_reorder(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ vs,
                                                 uint32_t n, const float4* __restrict__ posi,
                                                 const float4* __restrict__ velr, float4* __restrict__ posi_out,
                                                 float4* __restrict__ velr_out, uint32_t* __restrict__ key_out,
                                                 uint2* __restrict__ cells, uint32_t slot0) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t src = vs[i];
    uint32_t k = ks[i];
    posi_out[i] = posi[src];
    velr_out[i] = velr[src];
    key_out[i] = k;
    if (CELLS) {
        if (i == 0 || ks[i - 1] != k) cells[k].x = slot0 + i;
        if (i + 1 == n || ks[i + 1] != k) cells[k].y = slot0 + i + 1;
    }
}

// ---- initial conditions on the device: twin of sph_ic_dam_break (csrc/particleSystem.cpp) ------------------
__device__ __forceinline__ uint32_t ic_hash(uint32_t x) {   // lowbias32
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

struct LatticeDesc {
    uint32_t nx, ny;
    float box[3], jdim[3];
    float spacing, radius, jit;
    uint32_t seed;
    int jitter;
};

__global__ __launch_bounds__(256) void k_reset_lattice(float4* __restrict__ posi, float4* __restrict__ velr,
                                                       float2* __restrict__ dp, float4* __restrict__ pos_by_index,
                                                       uint64_t start, uint32_t count, LatticeDesc L) {
#pragma clang fp contract(off)      // one rounding per operation: no multiply-add fusion in this kernel
    uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= count) return;
    const uint64_t idx = start + k;
    const float ia[3] = {(float)(idx % L.nx), (float)((idx / L.nx) % L.ny), (float)(idx / ((uint64_t)L.nx * L.ny))};
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // one rounding per operation, in the host twin's order
        const float bmin = -L.box[a] / 2.0f;
        float base = (L.spacing * ia[a] + L.radius) + bmin;
        if (L.jitter) {
            uint32_t h = ic_hash((uint32_t)idx * 3u + (uint32_t)a + L.seed * 0x9E3779B9u);
            h = ic_hash(h ^ 0x85EBCA6Bu);
            const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
            const float w = L.jdim[a];
            base = base + (w * u - w / 2.0f) * L.jit;
        }
        p[a] = base;
    }
    posi[k] = make_float4(p[0], p[1], p[2], __uint_as_float((uint32_t)idx));
    velr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    dp[k] = make_float2(0.f, 0.f);
    if (pos_by_index) pos_by_index[idx] = make_float4(p[0], p[1], p[2], 1.0f);
}

int launch_reset_lattice(sph_ctx* c, const uint32_t lattice[3], int jitter, const float jitter_dims[3], uint64_t start,
                         uint32_t count) {
    LatticeDesc L;
    L.nx = lattice[0]; L.ny = lattice[1];
    for (int a = 0; a < 3; a++) {
        L.box[a] = c->params.box_max[a] - c->params.box_min[a];
        L.jdim[a] = jitter_dims ? jitter_dims[a] : L.box[a];
    }
    L.radius = c->params.particle_radius;
    L.spacing = 2.0f * L.radius;
    L.jit = L.radius * 0.01f;
    L.seed = 1973u;
    L.jitter = jitter;
    if (count)
        hipLaunchKernelGGL(k_reset_lattice, dim3(ceil_div(count, 256)), dim3(256), 0, c->stream, c->posi + c->own_off,
                           c->velr + c->own_off, c->dp + c->own_off, c->slab ? nullptr : c->pos_out, start, count, L);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

int launch_hash(sph_ctx* c) {
    if (c->n == 0 || c->keys_fresh) return SPH_OK;    // keys_fresh: the integrate epilogue already hashed
    hipLaunchKernelGGL(k_hash, dim3(ceil_div(c->n, 256)), dim3(256), 0, c->stream, c->posi + c->own_off, c->n,
                       c->grid, c->k0);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

template <int BITS>
static void sort_pass(sph_ctx* c, uint32_t n, uint32_t nblocks, uint32_t shift, bool first, const uint32_t* kin,
                      const uint32_t* vin, uint32_t* kout, uint32_t* vout) {
    hipLaunchKernelGGL(k_sort_hist<BITS>, dim3(nblocks), dim3(SORT_THREADS), 0, c->stream, kin, n, shift, nblocks, c->hist);
    hipLaunchKernelGGL(k_sort_scan, dim3(1 << BITS), dim3(256), 0, c->stream, c->hist, nblocks, c->digit_tot);
    if (first)
        hipLaunchKernelGGL((k_sort_scatter<BITS, true>), dim3(nblocks), dim3(SORT_THREADS), 0, c->stream, kin, vin, kout,
                           vout, n, shift, nblocks, c->hist, c->digit_tot);
    else
        hipLaunchKernelGGL((k_sort_scatter<BITS, false>), dim3(nblocks), dim3(SORT_THREADS), 0, c->stream, kin, vin, kout,
                           vout, n, shift, nblocks, c->hist, c->digit_tot);
}

int launch_sort(sph_ctx* c) {
    const uint32_t n = c->n;
    if (n == 0) return SPH_OK;
    const uint32_t nblocks = ceil_div(n, SORT_TILE);
    SPH_REQUIRE(nblocks <= c->sort_blocks_cap, SPH_E_CAPACITY, "sort: %u blocks > capacity %u", nblocks,
                c->sort_blocks_cap);
    uint32_t* kin = c->k0; uint32_t* vin = c->v0;
    uint32_t* kout = c->k1; uint32_t* vout = c->v1;
    // 9-bit digits when they save a pass over 8-bit ones (27 bits: 3 x 9), else 8-bit digits.  Measured
    // per pass at 16.7 M keys: 124 us (8 bits), 150 us (9 bits), 220 us (10 bits: never worth it).
    const uint32_t p8 = (c->key_bits + 7) / 8, p9 = (c->key_bits + 8) / 9;
    const uint32_t bits = p9 < p8 ? 9u : 8u;
    const uint32_t passes = bits == 9u ? p9 : p8;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = p * bits;
        if (bits == 8) sort_pass<8>(c, n, nblocks, shift, p == 0, kin, vin, kout, vout);
        else sort_pass<9>(c, n, nblocks, shift, p == 0, kin, vin, kout, vout);
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    // (kin, vin) now hold the sorted pairs; gather the payload to the canonical offset gcap.  A
    // whole-domain context gets no ghosts later, so its cell table is built in the same pass.
    const bool cells = !c->slab;
    if (cells)
        hipLaunchKernelGGL(k_reorder<true>, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, kin, vin, n,
                           c->posi + c->own_off, c->velr + c->own_off, c->posi2 + c->gcap, c->velr2 + c->gcap,
                           c->keyS + c->gcap, c->cells, c->gcap);
    else
        hipLaunchKernelGGL(k_reorder<false>, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, kin, vin, n,
                           c->posi + c->own_off, c->velr + c->own_off, c->posi2 + c->gcap, c->velr2 + c->gcap,
                           c->keyS + c->gcap, c->cells, c->gcap);
    SPH_HIP(hipGetLastError());
    float4* t4;
    t4 = c->posi; c->posi = c->posi2; c->posi2 = t4;
    t4 = c->velr; c->velr = c->velr2; c->velr2 = t4;
    c->own_off = c->gcap;
    c->last_perm = vin;
    if (cells) { c->cells_lo = c->gcap; c->cells_hi = c->gcap + n; c->cells_valid = true; }
    return SPH_OK;
}

// sorted slot -> slot before the sort (valid until the next sph_hash); used by the compat seam to
// move the caller's AoS structs the way thrust::sort would
const uint32_t* last_sort_permutation(sph_ctx* c) { return c->last_perm; }

}  // namespace sph
