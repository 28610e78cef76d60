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
// All three kernels walk "virtual blocks" (tiles of SORT_TILE keys) grid-stride, and take the element
// count either by value or from device memory (n_dev, clamped to n): the main sort launches one block
// per tile; the sort of the movers (see launch_sort_merge) is sized from a stale estimate and is
// correct for any count.
__device__ __forceinline__ uint32_t sort_count(uint32_t n, const uint32_t* __restrict__ n_dev) {
    return n_dev ? min(*n_dev, n) : n;
}

template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint32_t* __restrict__ keys, uint32_t n_arg,
                                                            const uint32_t* __restrict__ n_dev, uint32_t shift,
                                                            uint32_t nblocks, uint32_t* __restrict__ hist) {
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t h[RADIX];
    const uint32_t n = sort_count(n_arg, n_dev);
    const uint32_t nvb = (n + SORT_TILE - 1) / SORT_TILE;
    for (uint32_t vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) h[d] = 0;
        __syncthreads();
        uint32_t base = vb * SORT_TILE;
#pragma unroll
        for (int t = 0; t < SORT_KPT; t++) {
            uint32_t i = base + t * SORT_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&h[(keys[i] >> shift) & (RADIX - 1)], 1u);
        }
        __syncthreads();
        for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) hist[(size_t)d * nblocks + vb] = h[d];   // digit-major
        __syncthreads();
    }
}

// ---- pass 2 of 3: one block per digit scans its row of per-block counts --------------------------
__global__ __launch_bounds__(256) void k_sort_scan(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t n_arg,
                                                   const uint32_t* __restrict__ n_dev,
                                                   uint32_t* __restrict__ digit_tot) {
    __shared__ uint32_t part[256];
    const uint32_t n = sort_count(n_arg, n_dev);
    const uint32_t nvb = (n + SORT_TILE - 1) / SORT_TILE;      // entries of the row in use (<= nblocks, the stride)
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    uint32_t per = (nvb + 255u) / 256u;
    uint32_t lo = min(threadIdx.x * per, nvb);
    uint32_t hi = min(lo + per, nvb);
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
                                                               uint32_t* __restrict__ vout, uint32_t n_arg,
                                                               const uint32_t* __restrict__ n_dev,
                                                               uint32_t shift, uint32_t nblocks,
                                                               const uint32_t* __restrict__ hist,
                                                               const uint32_t* __restrict__ digit_tot) {
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX / SORT_THREADS;       // digits per thread: 1 or 2 (consecutive digits)
    __shared__ uint32_t wh[4][RADIX];     // per-wave digit counters -> running offsets
    __shared__ uint32_t part[SORT_THREADS];
    const uint32_t n = sort_count(n_arg, n_dev);
    const uint32_t nvb = (n + SORT_TILE - 1) / SORT_TILE;
    if (blockIdx.x >= nvb) return;                  // block-uniform
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
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
    const uint32_t digit_base = part[threadIdx.x] - sum;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    for (uint32_t vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        for (int w = 0; w < 4; w++)
            for (int d = threadIdx.x; d < RADIX; d += SORT_THREADS) wh[w][d] = 0;
        uint32_t my_base[DPT];
        {
            uint32_t run = digit_base;
#pragma unroll
            for (int k = 0; k < DPT; k++) {
                my_base[k] = run + hist[(size_t)(threadIdx.x * DPT + k) * nblocks + vb];
                run += tot[k];
            }
        }
        __syncthreads();

        // load this wave's keys (registers) and count digits per wave
        const uint32_t wbase = vb * SORT_TILE + wave * SORT_WAVE_TILE;
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
        __syncthreads();                                    // wh is re-zeroed by the next tile
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
static void sort_pass(sph_ctx* c, uint32_t n, const uint32_t* n_dev, uint32_t grid, uint32_t nblocks, uint32_t shift,
                      bool first, const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout) {
    hipLaunchKernelGGL(k_sort_hist<BITS>, dim3(grid), dim3(SORT_THREADS), 0, c->stream, kin, n, n_dev, shift, nblocks,
                       c->hist);
    hipLaunchKernelGGL(k_sort_scan, dim3(1 << BITS), dim3(256), 0, c->stream, c->hist, nblocks, n, n_dev, c->digit_tot);
    if (first)
        hipLaunchKernelGGL((k_sort_scatter<BITS, true>), dim3(grid), dim3(SORT_THREADS), 0, c->stream, kin, vin, kout,
                           vout, n, n_dev, shift, nblocks, c->hist, c->digit_tot);
    else
        hipLaunchKernelGGL((k_sort_scatter<BITS, false>), dim3(grid), dim3(SORT_THREADS), 0, c->stream, kin, vin, kout,
                           vout, n, n_dev, shift, nblocks, c->hist, c->digit_tot);
}

// LSD radix sort of (key, value) pairs over the context's significant key bits.  `first`: the values of
// the first pass are the element indices (vin unused).  n_dev != null: the count lives on the device
// (<= n), `grid` blocks walk the tiles.  Returns through kin/vin the buffers that hold the result.
static void radix_sort_pairs(sph_ctx* c, uint32_t n, const uint32_t* n_dev, uint32_t grid, bool first, uint32_t*& kin,
                             uint32_t*& vin, uint32_t*& kout, uint32_t*& vout) {
    const uint32_t nblocks = ceil_div(n, SORT_TILE);          // row stride of the histogram
    // 9-bit digits when they save a pass over 8-bit ones (27 bits: 3 x 9), else 8-bit digits.  Measured
    // per pass at 16.7 M keys: 124 us (8 bits), 150 us (9 bits), 220 us (10 bits: never worth it).
    const uint32_t p8 = (c->key_bits + 7) / 8, p9 = (c->key_bits + 8) / 9;
    const uint32_t bits = p9 < p8 ? 9u : 8u;
    const uint32_t passes = bits == 9u ? p9 : p8;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = p * bits;
        if (bits == 8) sort_pass<8>(c, n, n_dev, grid, nblocks, shift, first && p == 0, kin, vin, kout, vout);
        else sort_pass<9>(c, n, n_dev, grid, nblocks, shift, first && p == 0, kin, vin, kout, vout);
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
}

// ---- the sort as a merge: only the particles whose cell changed are sorted -----------------------------------
// Between two steps a particle moves a small fraction of a cell, so after the integrate almost every key
// equals the key its slot was sorted under.  With A = the keys of the current (sorted) order and B = the
// new keys, the non-movers (A[i] == B[i]) are already in order; the movers are compacted (stable), radix
// sorted on their own, and both sequences get their merged positions by rank:
//   non-mover i :  (i - #movers before i) + #movers with (key, slot) < (A[i], i)
//   mover r     :  r + #non-movers with (key, slot) < (B, slot)
// The result is the (key, slot) sequence of the full stable sort, element for element, for ANY number of
// movers; launch_sort only prefers the full sort when the last known mover count makes it cheaper.

__global__ __launch_bounds__(256) void k_mm_mark(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                 uint32_t n, uint64_t* __restrict__ mask,
                                                 uint32_t* __restrict__ tile_cnt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool mv = i < n && A[i] != B[i];
    const uint64_t m = __ballot(mv);
    if ((threadIdx.x & 63u) == 0 && (i & ~63u) < n) {
        const uint32_t chunk = i >> 6;
        mask[chunk] = m;
        if (m) atomicAdd(&tile_cnt[chunk / MM_TILE_CHUNKS], (uint32_t)__popcll(m));   // sparse
    }
}

// one block: exclusive scan of the per-tile mover counts; re-zeroes the counts for the next step
__global__ __launch_bounds__(1024) void k_mm_tilescan(uint32_t* __restrict__ tile_cnt, uint32_t nt,
                                                      uint32_t* __restrict__ tile_off, uint32_t* __restrict__ m_dev,
                                                      volatile uint32_t* __restrict__ m_host,
                                                      unsigned long long* __restrict__ m_total) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nt + 1023u) / 1024u;
    const uint32_t lo = min(threadIdx.x * per, nt), hi = min(lo + per, nt);
    uint32_t s = 0;
    for (uint32_t t = lo; t < hi; t++) s += tile_cnt[t];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= (uint32_t)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t t = lo; t < hi; t++) {
        uint32_t v = tile_cnt[t];
        tile_off[t] = run;
        tile_cnt[t] = 0;
        run += v;
    }
    if (threadIdx.x == 1023) {
        *m_dev = part[1023]; *m_host = part[1023];
        if (m_total) *m_total += part[1023];          // one block: no atomic needed (sph_sort_stats: movers_total)
    }
}

// thread per chunk: movers before the chunk (M64) and the stable list of movers (new key, slot)
__global__ __launch_bounds__(256) void k_mm_compact(const uint64_t* __restrict__ mask, uint32_t nchunks,
                                                    const uint32_t* __restrict__ tile_off,
                                                    const uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                    uint32_t* __restrict__ M64, uint32_t* __restrict__ mk,
                                                    uint32_t* __restrict__ mi, uint2* __restrict__ cells) {
    __shared__ uint32_t part[256];
    const uint32_t chunk = blockIdx.x * MM_TILE_CHUNKS + threadIdx.x;
    uint64_t m = chunk < nchunks ? mask[chunk] : 0ull;
    const uint32_t cnt = (uint32_t)__popcll(m);
    part[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = threadIdx.x >= (uint32_t)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    if (chunk >= nchunks) return;
    uint32_t at = tile_off[blockIdx.x] + part[threadIdx.x] - cnt;
    M64[chunk] = at;
    while (m) {
        const uint32_t i = chunk * 64u + (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        mk[at] = B[i];
        mi[at] = i;
        at++;
        // only a cell a mover left can have become empty: clear those, the reorder pass rewrites every
        // cell that is still occupied (replaces the walk over all old keys, k_cells_clear)
        if (cells) cells[A[i]] = make_uint2(0u, 0u);
    }
}

// first r in [lo, hi) with (mk[r], mi[r]) >= (key, slot)
__device__ __forceinline__ uint32_t mm_lower_bound(const uint32_t* __restrict__ mk, const uint32_t* __restrict__ mi,
                                                   uint32_t lo, uint32_t hi, uint32_t key, uint32_t slot) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t k = mk[mid];
        const bool less = k < key || (k == key && mi[mid] < slot);
        if (less) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// non-movers: one thread per slot.  The bracket [L0, L1] of a whole wave is found with wave-uniform
// (scalar) searches; it is a single point unless a mover lands inside the wave's key span.
__global__ __launch_bounds__(256) void k_mm_place(const uint32_t* __restrict__ A, uint32_t n,
                                                  const uint64_t* __restrict__ mask, const uint32_t* __restrict__ M64,
                                                  const uint32_t* __restrict__ mk, const uint32_t* __restrict__ mi,
                                                  const uint32_t* __restrict__ m_dev, uint32_t* __restrict__ ks,
                                                  uint32_t* __restrict__ vs) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t first = __builtin_amdgcn_readfirstlane(i);          // slot of lane 0
    if (first >= n) return;                                            // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = first >> 6;
    const uint32_t m = *m_dev;
    const uint32_t last = min(first + 63u, n - 1u);
    const uint32_t L0 = mm_lower_bound(mk, mi, 0u, m, A[first], first);
    const uint32_t L1 = mm_lower_bound(mk, mi, L0, m, A[last], last);
    if (i >= n) return;
    const uint64_t bits = mask[chunk];
    if ((bits >> lane) & 1ull) return;                                 // a mover: placed by k_mm_place_movers
    const uint32_t key = A[i];
    const uint32_t before = M64[chunk] + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));
    const uint32_t L = L0 == L1 ? L0 : mm_lower_bound(mk, mi, L0, L1, key, i);
    const uint32_t dst = i - before + L;
    ks[dst] = key;
    vs[dst] = i;
}

__global__ __launch_bounds__(256) void k_mm_place_movers(const uint32_t* __restrict__ A, uint32_t n,
                                                         const uint64_t* __restrict__ mask,
                                                         const uint32_t* __restrict__ M64,
                                                         const uint32_t* __restrict__ mk,
                                                         const uint32_t* __restrict__ mi,
                                                         const uint32_t* __restrict__ m_dev, uint32_t* __restrict__ ks,
                                                         uint32_t* __restrict__ vs) {
    const uint32_t m = *m_dev;
    for (uint32_t r = blockIdx.x * 256u + threadIdx.x; r < m; r += gridDim.x * 256u) {
        const uint32_t key = mk[r], slot = mi[r];
        uint32_t lo = 0, hi = n;                       // s = first slot with A >= key
        while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (A[mid] < key) lo = mid + 1; else hi = mid; }
        const uint32_t s = lo;
        hi = n;                                        // e = first slot with A > key
        while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (A[mid] <= key) lo = mid + 1; else hi = mid; }
        const uint32_t e = lo;
        const uint32_t j = min(max(slot, s), e);       // non-movers of cell `key` below `slot` end here
        uint32_t before = m;                           // movers among the slots [0, j)
        if (j < n) before = M64[j >> 6] + (uint32_t)__popcll(mask[j >> 6] & ((1ull << (j & 63u)) - 1ull));
        const uint32_t dst = r + (j - before);
        ks[dst] = key;
        vs[dst] = slot;
    }
}

static uint32_t merge_grid_for(uint32_t movers_hint, uint32_t n) {
    const uint32_t want = ceil_div(2u * movers_hint + 1u, SORT_TILE) + 15u;    // head-room: the hint is stale
    return min(want, ceil_div(n, SORT_TILE));
}

// `counted`: the movers belong to a sort (they add to the running total), not to marks being dropped
static void mm_tilescan(sph_ctx* c, uint32_t n, bool counted) {
    const uint32_t nt = ceil_div(ceil_div(n, 64u), MM_TILE_CHUNKS);
    hipLaunchKernelGGL(k_mm_tilescan, dim3(1), dim3(1024), 0, c->stream, c->mm_tile_cnt, nt, c->mm_tile_off, c->mm_count,
                       c->mm_count_host_dev, counted ? c->mm_total : (unsigned long long*)nullptr);
}

// forget the marks the integrate epilogue left (the scan re-zeroes the tile counts they added to)
void mm_drop_marks(sph_ctx* c) {
    if (!c->mm_marked) return;
    mm_tilescan(c, c->mm_marked_n, false);
    c->mm_marked = false;
}

// (ks, vs) of the stable sort by B, from the current order (sorted by A); see the block comment above
// step 1 of the merge: the movers are marked (by the integrate epilogue, or here) and counted
static void launch_merge_count(sph_ctx* c, uint32_t n) {
    if (c->mm_marked && !(c->mm_marked_off == c->own_off && c->mm_marked_n == n)) mm_drop_marks(c);   // another range
    if (!c->mm_marked)                           // else: the fused integrate epilogue compared the keys already
        hipLaunchKernelGGL(k_mm_mark, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, c->keyS + c->own_off, c->k0, n,
                           c->mm_mask, c->mm_tile_cnt);
    c->mm_marked = false;
    mm_tilescan(c, n, true);
}

static void launch_sort_merge(sph_ctx* c, uint32_t n, bool table_live, uint32_t*& ks, uint32_t*& vs) {
    const uint32_t* A = c->keyS + c->own_off;
    const uint32_t* B = c->k0;
    const uint32_t nchunks = ceil_div(n, 64u), nt = ceil_div(nchunks, MM_TILE_CHUNKS);
    uint32_t* mk = c->mm_k0; uint32_t* mi = c->v0; uint32_t* mk2 = c->mm_k1; uint32_t* mi2 = c->mm_v1;
    hipLaunchKernelGGL(k_mm_compact, dim3(nt), dim3(256), 0, c->stream, c->mm_mask, nchunks, c->mm_tile_off, A, B,
                       c->mm_M64, mk, mi, table_live ? c->cells : (uint2*)nullptr);
    const uint32_t hint = *c->mm_count_host;                 // whatever step last reported: sizes the grid only
    radix_sort_pairs(c, n, c->mm_count, merge_grid_for(hint, n), false, mk, mi, mk2, mi2);
    ks = c->k1; vs = c->v1;
    hipLaunchKernelGGL(k_mm_place, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, A, n, c->mm_mask, c->mm_M64, mk, mi,
                       c->mm_count, ks, vs);
    hipLaunchKernelGGL(k_mm_place_movers, dim3(min(ceil_div(2u * hint + 1u, 256u) + 15u, 65535u)), dim3(256), 0, c->stream,
                       A, n, c->mm_mask, c->mm_M64, mk, mi, c->mm_count, ks, vs);
}

int launch_sort(sph_ctx* c) {
    const uint32_t n = c->n;
    if (n == 0) return SPH_OK;
    const uint32_t nblocks = ceil_div(n, SORT_TILE);
    SPH_REQUIRE(nblocks <= c->sort_blocks_cap, SPH_E_CAPACITY, "sort: %u blocks > capacity %u", nblocks,
                c->sort_blocks_cap);
    uint32_t* kin = c->k0; uint32_t* vin = c->v0;
    uint32_t* kout = c->k1; uint32_t* vout = c->v1;
    // The merge needs the order of the last sort to be intact; it is correct for any number of movers but
    // only cheaper than the full sort while they are few (last known count: a hint, never a condition).
    // The mover count the host reads below is whatever the device last reported.  A caller that queues many
    // steps without synchronising would decide all of them on one stale value, so the host never runs more
    // than four sorts ahead of the device (the queue stays several steps deep: the device never waits).
    const uint32_t ring = (uint32_t)(c->sort_calls & 3u);
    if (c->sort_merge && c->sort_calls >= 4) SPH_HIP(hipEventSynchronize(c->mm_done[ring]));
    c->sort_calls++;
    c->last_sort_skipped = false;
    const bool can_merge = c->sort_merge && c->order_valid;
    // whole-domain contexts: sph_hash left the old cell table in place (cells_clear_deferred) when this
    // sort could take the merge path, which clears only the cells the movers left
    const bool table_live = c->cells_clear_deferred && c->cells_valid;
    c->cells_clear_deferred = false;
    bool table_kept = false;               // the merge path keeps the live table and clears it sparsely
    if (can_merge && *c->mm_count_host <= n / 8u) {
        const bool was_still = *c->mm_count_host == 0u;
        launch_merge_count(c, n);
        if (was_still && table_live && c->own_off == c->gcap) {
            // Nothing moved last time (a fluid at rest: no particle crosses a cell face for many steps).  If that
            // is still so, the order, the keys and the cell table are already those of this step and the whole
            // sort -- 0.3 ms of copying at C3 -- can be left out.  Only the device knows, and the host does not
            // wait for it: the count is looked at only if the device has ALREADY produced it (a caller in
            // lockstep with the device, e.g. one update() per frame); a host that runs ahead of the device
            // queues the merge, which does the same job for 0 movers.  sph_step stays asynchronous.
            SPH_HIP(hipEventRecord(c->mm_counted, c->stream));
            if (hipEventQuery(c->mm_counted) == hipSuccess && *c->mm_count_host == 0u) {
                c->sort_merges++;
                c->sort_skips++;
                c->last_sort_skipped = true;
                SPH_HIP(hipEventRecord(c->mm_done[ring], c->stream));
                c->last_perm = nullptr;            // identity
                c->order_valid = true;             // cells_valid / cells_lo / cells_hi: unchanged and still true
                return SPH_OK;
            }
        }
        launch_sort_merge(c, n, table_live, kin, vin);
        c->sort_merges++;
        table_kept = table_live;
    } else {
        if (table_live) {
            int rc = launch_cells_clear(c);
            if (rc) return rc;
        }
        // keep the hint alive, or it would stay high for ever: for free when the integrate epilogue marked
        // the movers (the scan also re-zeroes the tile counts those marks added to), else every 8th sort
        if (c->mm_marked) {
            mm_tilescan(c, c->mm_marked_n, true);
            c->mm_marked = false;
        } else if (can_merge && (c->sort_calls & 7u) == 0) {
            hipLaunchKernelGGL(k_mm_mark, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, c->keyS + c->own_off, c->k0, n,
                               c->mm_mask, c->mm_tile_cnt);
            mm_tilescan(c, n, true);
        }
        radix_sort_pairs(c, n, nullptr, nblocks, true, kin, vin, kout, vout);
    }
    // (kin, vin) now hold the sorted pairs; gather the payload to the canonical offset gcap and write the cell
    // table of the owned slots in the same pass.  A slab context adds the cells of its ghost layers later
    // (launch_cells_build), and drops those of the particles that leave (sph_migrants_pack).
    if (c->cells_valid && !table_kept) {   // a table nobody cleared (e.g. sph_sort without sph_hash): start clean
        int rc = launch_cells_clear(c);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_reorder<true>, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, kin, vin, n,
                       c->posi + c->own_off, c->velr + c->own_off, c->posi2 + c->gcap, c->velr2 + c->gcap,
                       c->keyS + c->gcap, c->cells, c->gcap);
    SPH_HIP(hipGetLastError());
    if (c->sort_merge) SPH_HIP(hipEventRecord(c->mm_done[ring], c->stream));
    float4* t4;
    t4 = c->posi; c->posi = c->posi2; c->posi2 = t4;
    t4 = c->velr; c->velr = c->velr2; c->velr2 = t4;
    c->own_off = c->gcap;
    c->last_perm = vin;
    c->order_valid = true;
    c->cells_lo = c->gcap; c->cells_hi = c->gcap + n; c->cells_valid = true;
    return SPH_OK;
}

// sorted slot -> slot before the sort (valid until the next sph_hash); used by the compat seam to
// move the caller's AoS structs the way thrust::sort would
const uint32_t* last_sort_permutation(sph_ctx* c) { return c->last_perm; }

}  // namespace sph
