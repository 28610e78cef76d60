// sph_pairs.hip -- cell table and the neighbour passes (density, force, collision, integrate).
//
// Replaces cudaConstructBGrid / cudaConstructGridArray / cudaComputeDensities / cudaComputeForces /
// cudaParticleCollisions / cudaIntegrate (particleSystem.cu:479-534, kernels :132-420).
//
// Work mapping (MI355X, wave64).  The reference launches one 32-thread block per <=32-particle
// chunk of one cell (B'), i.e. ~8 live lanes of 32 at rest density, and re-stages whole 88-byte
// structs of the 27 neighbour cells through shared memory with three block barriers per batch.
// Here a wave takes 64 CONSECUTIVE particles of the sorted SoA arrays (always 64 live lanes).
// Cells are numbered x-fastest, so for every particle the 27 neighbour cells are 9 ROWS
// (dz, dy) of up to 3 consecutive cells = 9 contiguous slot ranges [lo, hi) of the sorted arrays.
// Per row the wave stages the hull of its lanes' ranges (about 10 cells, ~80 particles at rest
// density) into a wave-private LDS slice with coalesced 16-byte loads, then every lane walks ITS
// OWN range, so each lane tests exactly its 27-cell candidates (~216) -- the reference semantics
// (the 3x3x3 stencil truncates the support ball: cell edge 0.0625 < h = 0.1).
// No block barriers: LDS traffic of one wave is processed in order.
#include "sph_device.hpp"

namespace sph {

#ifndef SPH_PAIR_THREADS
#define SPH_PAIR_THREADS 256
#endif
constexpr int PAIR_THREADS = SPH_PAIR_THREADS;   // 4 independent waves
constexpr int PAIR_WAVES = PAIR_THREADS / WAVE;
// block of the fp32 density pass (A/B knob).  Round 4 tried 128 for it alone: +1.0 % SQ cycles, wall clock in the noise
// (profiles/r04_block_size_and_xcd_pmc_ab.txt: with 128 threads for BOTH kernels the density pass reads -2.2 % and k_force
// +5.4 %, with 512 both are worse; an XCD-contiguous block order gives k_force -1.2 % and the density pass +1.7 %: all
// within what the clock ratio of a power-capped chip moves these counters)
#ifndef SPH_DENS_THREADS
#define SPH_DENS_THREADS SPH_PAIR_THREADS
#endif
constexpr int DENS_THREADS = SPH_DENS_THREADS;
// A SMALL context (fewer than sph_ctx::pair_small_slots owned particles) launches blocks of 128 threads.  Waves are independent
// of each other (a wave-private LDS slice, no block barrier in the walk), so the block size changes nothing in the results --
// only how the dispatcher spreads the waves and how long a block's slowest wave holds its CU's resources: config 2's flowing
// dam (4096 waves in all, some of them sparse and slow) runs `k_force` 14 % and `k_density` 7 % faster with 2-wave blocks, the
// step 184 -> 162 us; a lattice at rest of the same size does not care; from ~10^6 particles on 128-thread blocks LOSE (config
// 3: `k_force` +7 %: twice the blocks re-stage the same hulls, and BlockOrder's z strips are cut for 256), and the boundary
// launches of a 2.1 M-particle slab lose ~1 % of its step.  Measured: profiles/r06_pair_block_size.txt.
constexpr int SMALL_THREADS_PAIR = 128;
static inline bool small_blocks(const sph_ctx* c) { return c->n < c->pair_small_slots; }
constexpr int PIECE = 128;          // staged candidates per piece (2 coalesced loads per lane)

// The candidate walks read LDS through volatile LDS-address-space pointers: each read then stays ONE
// ds_read_b64 / ds_read_b32 (2 LDS cycles per wave) with its own immediate offset.  Left alone, hipcc fuses
// neighbouring reads into ds_read2_b64 / ds_read2_b32, which take 8 / 4 cycles -- half the rate per byte
// (MI355X_MICROARCH.md, LDS table) -- and that kept the LDS pipe 69 % busy (40 % with single reads).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) v2f* lds_v2f_ptr;
typedef const volatile __attribute__((address_space(3))) float* lds_f32_ptr;

// ---- cell table -----------------------------------------------------------------------------------
// {start, end} per occupied cell; empty cells stay {0, 0}.  Only the cells touched by the previous
// build are cleared (the reference memsets the whole table every step, particleSystem.cu:506, which
// is 1 GiB at 512^3).  No atomics (the reference: one atomicAdd per particle, :318,327).
__global__ __launch_bounds__(256) void k_cells_clear(const uint32_t* __restrict__ key, uint32_t lo, uint32_t hi,
                                                     uint2* __restrict__ cells) {
    uint32_t s = lo + blockIdx.x * 256u + threadIdx.x;
    if (s >= hi) return;
    uint32_t k = key[s];
    if (s == lo || key[s - 1] != k) cells[k] = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(256) void k_cells_build(const uint32_t* __restrict__ key, uint32_t lo, uint32_t hi,
                                                     uint2* __restrict__ cells, volatile uint32_t* __restrict__ ends_host, uint32_t seq) {
    cells_build_thread(key, lo, hi, cells, ends_host, seq, blockIdx.x * 256u + threadIdx.x);      // sph_device.hpp
}

// the same two kernels over TWO slot ranges in one launch (a slab's two ghost ranges, the leavers at both ends)
__global__ __launch_bounds__(256) void k_cells_clear2(const uint32_t* __restrict__ key, uint32_t lo0, uint32_t hi0,
                                                      uint32_t lo1, uint32_t hi1, uint2* __restrict__ cells) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, n0 = hi0 - lo0;
    const uint32_t lo = t < n0 ? lo0 : lo1, hi = t < n0 ? hi0 : hi1;
    const uint32_t s = t < n0 ? lo0 + t : lo1 + (t - n0);
    if (s >= hi) return;
    const uint32_t k = key[s];
    if (s == lo || key[s - 1] != k) cells[k] = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(256) void k_cells_build2(const uint32_t* __restrict__ key, uint32_t lo0, uint32_t hi0,
                                                      uint32_t lo1, uint32_t hi1, uint2* __restrict__ cells) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, n0 = hi0 - lo0;
    const uint32_t lo = t < n0 ? lo0 : lo1, hi = t < n0 ? hi0 : hi1;
    const uint32_t s = t < n0 ? lo0 + t : lo1 + (t - n0);
    if (s >= hi) return;
    const uint32_t k = key[s];
    if (s == lo || key[s - 1] != k) cells[k].x = s;
    if (s + 1 == hi || key[s + 1] != k) cells[k].y = s + 1;
}

int launch_cells_clear_2ranges(sph_ctx* c, uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1) {
    if (hi0 < lo0) hi0 = lo0;
    if (hi1 < lo1) hi1 = lo1;
    const uint32_t tot = (hi0 - lo0) + (hi1 - lo1);
    if (!tot) return SPH_OK;
    hipLaunchKernelGGL(k_cells_clear2, dim3(ceil_div(tot, 256)), dim3(256), 0, c->stream, c->keyS, lo0, hi0, lo1, hi1, c->cells);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

int launch_cells_build_2ranges(sph_ctx* c, uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1) {
    if (hi0 < lo0) hi0 = lo0;
    if (hi1 < lo1) hi1 = lo1;
    const uint32_t tot = (hi0 - lo0) + (hi1 - lo1);
    if (!tot) return SPH_OK;
    hipLaunchKernelGGL(k_cells_build2, dim3(ceil_div(tot, 256)), dim3(256), 0, c->stream, c->keyS, lo0, hi0, lo1, hi1, c->cells);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

int launch_cells_clear(sph_ctx* c) {
    if (!c->cells_valid || c->cells_hi <= c->cells_lo) { c->cells_valid = false; return SPH_OK; }
    hipLaunchKernelGGL(k_cells_clear, dim3(ceil_div(c->cells_hi - c->cells_lo, 256)), dim3(256), 0, c->stream, c->keyS,
                       c->cells_lo, c->cells_hi, c->cells);
    SPH_HIP(hipGetLastError());
    c->cells_valid = false;
    return SPH_OK;
}

// clear / build the table entries of the cells that occur in the slot range [lo, hi) (no bookkeeping)
int launch_cells_clear_range(sph_ctx* c, uint32_t lo, uint32_t hi) {
    if (hi <= lo) return SPH_OK;
    hipLaunchKernelGGL(k_cells_clear, dim3(ceil_div(hi - lo, 256)), dim3(256), 0, c->stream, c->keyS, lo, hi, c->cells);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

int launch_cells_build_range(sph_ctx* c, uint32_t lo, uint32_t hi) {
    if (hi <= lo) return SPH_OK;
    // (the whole owned range: the sort's table build -- also tells the host the first and the last key)
    uint32_t* ends = (lo == c->own_off && hi == c->own_off + c->n) ? c->mm_count_host_dev + 1 : (uint32_t*)nullptr;
    const uint32_t seq = ends ? c->cells_seq_next : 0u;
    if (ends) c->cells_seq_next = 0u;
    hipLaunchKernelGGL(k_cells_build, dim3(cells_build_blocks(hi - lo)), dim3(256), 0, c->stream, c->keyS, lo, hi, c->cells, ends, seq);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

int launch_cells_build(sph_ctx* c) {
    const uint32_t lo = c->own_off - c->n_glo, hi = c->own_off + c->n + c->n_ghi;
    int rc;
    if (c->cells_valid && c->cells_lo == c->own_off && c->cells_hi == c->own_off + c->n) {
        // the sort's reorder pass built the owned cells; ghost layers hold no owned particle, so their
        // cells are disjoint from those: add them from the two ghost ranges only
        rc = launch_cells_build_2ranges(c, lo, c->own_off, c->own_off + c->n, hi);
    } else {
        rc = launch_cells_clear(c);      // a table over another slot set, if any
        if (!rc) rc = launch_cells_build_range(c, lo, hi);
    }
    if (rc) return rc;
    c->cells_lo = lo; c->cells_hi = hi; c->cells_valid = true;
    return SPH_OK;
}

// ---- per-lane candidate ranges -----------------------------------------------------------------------
struct Rows {
    uint32_t lo[9], hi[9];
};

// rows in (dz, dy) order, cells of a row in dx order, particles of a cell in slot order: the
// candidate order of one particle is fixed, so its fp32 sums are reproducible run to run.
//
// Branch-free: all 27 table entries are requested before any is used.  The three cells of a row are
// cells[k-1], cells[k], cells[k+1] with k = key + (dz*gy + dy)*gx -- one address, three immediate offsets;
// the table has a zero guard entry on either side, so the reads are in bounds for every key, and whatever
// lies outside the grid (cx-1 < 0, cx+1 >= gx, a row beyond the y/z faces, an idle lane) is masked after
// the loads.  A row beyond a face reads the particle's own row instead (any valid address will do).
__device__ __forceinline__ void lane_rows(const uint2* __restrict__ cells, const GridDesc& g, uint32_t key, bool active,
                                          Rows& R) {
    const uint32_t gx = g.g[0], gy = g.g[1];
    uint32_t cx, cy, lz;
    if (((gx & (gx - 1u)) | (gy & (gy - 1u))) == 0u) {   // the reference's grids are powers of two (nextPow2)
        const uint32_t sx = 31u - (uint32_t)__clz((int)gx), sy = 31u - (uint32_t)__clz((int)gy);
        cx = key & (gx - 1u);
        cy = (key >> sx) & (gy - 1u);
        lz = key >> (sx + sy);
    } else {
        cx = key % gx;
        const uint32_t t = key / gx;
        cy = t % gy;
        lz = t / gy;
    }
    const bool has_l = cx > 0u, has_r = cx + 1u < gx;
    const bool y_ok[3] = {cy > 0u, true, cy + 1u < gy};
    const bool z_ok[3] = {lz > 0u, true, lz + 1u < g.zl};
    const int sy_ = (int)gx, sz_ = (int)(gx * gy);       // wave-uniform strides
    uint2 a[9], b[9], d[9];
    bool ok[9];
#pragma unroll
    for (int dz = -1; dz <= 1; dz++) {
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) {
            const int r = (dz + 1) * 3 + (dy + 1);
            ok[r] = active && y_ok[dy + 1] && z_ok[dz + 1];
            const uint2* p = cells + (ok[r] ? (int)key + dz * sz_ + dy * sy_ : (int)key);
            a[r] = p[-1]; b[r] = p[0]; d[r] = p[1];
        }
    }
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const bool na = ok[r] && has_l && a[r].y > a[r].x, nb = ok[r] && b[r].y > b[r].x,
                   nd = ok[r] && has_r && d[r].y > d[r].x;
        // first non-empty start / last non-empty end; all empty -> lo = hi = 0
        R.lo[r] = na ? a[r].x : (nb ? b[r].x : (nd ? d[r].x : 0u));
        R.hi[r] = nd ? d[r].y : (nb ? b[r].y : (na ? a[r].y : 0u));
    }
}

// the same for ONE row with a run-time (dz, dy): the direct walk below looks its rows up again instead of keeping the
// nine ranges of lane_rows alive in a second branch (that cost the staged walk of k_force a register spill)
__device__ __forceinline__ void lane_row_one(const uint2* __restrict__ cells, const GridDesc& g, uint32_t key, bool active,
                                             int dz, int dy, uint32_t& lo, uint32_t& hi) {
    const uint32_t gx = g.g[0], gy = g.g[1];
    uint32_t cx, cy, lz;
    if (((gx & (gx - 1u)) | (gy & (gy - 1u))) == 0u) {   // powers of two, as in lane_rows
        const uint32_t sx = 31u - (uint32_t)__clz((int)gx), sy = 31u - (uint32_t)__clz((int)gy);
        cx = key & (gx - 1u); cy = (key >> sx) & (gy - 1u); lz = key >> (sx + sy);
    } else {
        cx = key % gx;
        const uint32_t t = key / gx;
        cy = t % gy; lz = t / gy;
    }
    const bool ok = active && (int)cy + dy >= 0 && (int)cy + dy < (int)gy && (int)lz + dz >= 0 && (int)lz + dz < (int)g.zl;
    const uint2* p = cells + (ok ? (int)key + dz * (int)(gx * gy) + dy * (int)gx : (int)key);
    const uint2 a = p[-1], b = p[0], d = p[1];
    const bool na = ok && cx > 0u && a.y > a.x, nb = ok && b.y > b.x, nd = ok && cx + 1u < gx && d.y > d.x;
    lo = na ? a.x : (nb ? b.x : (nd ? d.x : 0u));
    hi = nd ? d.y : (nb ? b.y : (na ? a.y : 0u));
}

#ifndef SPH_DENS_OCC
#define SPH_DENS_OCC 6      // waves per SIMD asked of the register allocator (<= 80 VGPRs)
#endif
#ifndef SPH_FORCE_OCC
#define SPH_FORCE_OCC 5     // <= 96 VGPRs (95 used, no spill); 4, 5 and 6 waves time within 2 % of each other
#endif

// A wave-uniform constant lives in an SGPR, and a VALU instruction with an SGPR source issues at ~0.6 of the
// rate of an all-VGPR one in a back-to-back stream on gfx950 (profiles/valu_rate.hip: 1.75 vs 1.05-1.25 ns per
// wave-instruction per SIMD; the same for v_cmp + v_addc through an SGPR pair).  Inside the real loops the
// effect is small (k_force -1 %), but the constants of the candidate loops are parked in VGPRs all the same;
// the empty asm hides their uniformity from the compiler.
__device__ __forceinline__ float in_vgpr(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ float inv_sqrt(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32, 1 ulp

__device__ __forceinline__ void wave_lds_sync() {
    // one wave's LDS operations are processed in issue order; only the compiler must not move them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS (LDS_ENT entries): one PIECE-entry slice per wave plus one spare slice.  The unrolled candidate loop may read up
// to PIECE+3 entries past the end of a lane's valid range; the spare slice keeps those reads inside
// the block's allocation.  Such lanes are masked by selecting a zero WEIGHT, and a zero weight times
// a NaN is a NaN, so everything a lane can over-read must be finite: the arrays are zero-filled once
// per block, and whatever is staged later comes from the (zero-padded) particle arrays.
constexpr int lds_ent(int threads) { return (threads / WAVE + 1) * PIECE + 8; }      // "LDS_ENT" of a block of `threads`
#ifndef SPH_DENS_UNROLL
#define SPH_DENS_UNROLL 4
#endif
constexpr int UNROLL = SPH_DENS_UNROLL;        // density: candidates per unrolled group
#ifndef SPH_FORCE_UNROLL
#define SPH_FORCE_UNROLL 2
#endif
// 1: IEEE divisions in the integrate / collision epilogue as the reference writes them (kernelIntegrate f / rho,
// particleSystem.cu:375-420; kernelComputeCollisions :299); 0 (product): one v_rcp_f32 each.  A/B knob, see integrate_one.
#ifndef SPH_EXACT_DIV
#define SPH_EXACT_DIV 0
#endif

// Per-row hulls of the wave's candidate ranges (wave-uniform).
struct Hulls {
    uint32_t A[9], B[9];
};

// Lanes are in key order and the row offset is the same for every lane, so lo and hi are
// non-decreasing across the lanes that have a range: the hull is [lo of the first such lane, hi of
// the last one) -- one ballot and two v_readlane instead of two cross-lane reductions.
__device__ __forceinline__ void wave_hulls(const Rows& R, Hulls& H) {
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const uint64_t m = __ballot(R.hi[r] > R.lo[r]);
        if (m == 0ull) { H.A[r] = 0u; H.B[r] = 0u; continue; }
        const int first = __builtin_ctzll(m), last = 63 - __builtin_clzll(m);
        H.A[r] = (uint32_t)__builtin_amdgcn_readlane((int)R.lo[r], first);
        H.B[r] = (uint32_t)__builtin_amdgcn_readlane((int)R.hi[r], last);
    }
}

// Walk the 9 rows: stage every piece [a, b) of a row's hull through registers into the wave's LDS
// slice (`load(a)` issues the global loads, `store()` writes them to LDS) and call `pieces(r, a, b)`
// for the per-lane work.  The first piece of the NEXT row is requested before the current piece
// is processed, so its global-memory latency hides behind the pair arithmetic.
//
template <class Load, class Store, class Work>
__device__ __forceinline__ void traverse(const Hulls& H, Load&& load, Store&& store, Work&& work) {
    bool ready = false;
#pragma unroll
    for (int r = 0; r < 9; r++) {
        if (H.A[r] >= H.B[r]) continue;          // wave-uniform
        if (!ready) load(H.A[r]);
        ready = false;
        for (uint32_t a = H.A[r]; a < H.B[r]; a += PIECE) {
            const uint32_t b = min(a + PIECE, H.B[r]);
            if (a != H.A[r]) load(a);
            store();
            if (b >= H.B[r] && r + 1 < 9 && H.A[r + 1] < H.B[r + 1]) {
                load(H.A[r + 1]);
                ready = true;
            }
            wave_lds_sync();
            work(r, a, b);
            wave_lds_sync();
        }
    }
}

// A hull is the image of the wave's key interval under the row's offset, i.e. about as many slots as the wave's own
// 64 -- UNLESS the wave's particles are sparse and the row is dense: 64 particles of a nearly empty cell layer (the
// first few of a lattice plane that crosses a z face, spray) span 5-8 y-rows, and their dz = -1 rows then cover 5-8
// whole y-rows of the dense layer next door: hulls of 12,000-19,000 entries where 80 are usual, a hundred pieces
// staged and walked for a handful of candidates each.  28 such waves out of 32,768 made k_density 2.1x and k_force
// 1.7x slower as the kernel's TAIL (profiles/r04_stretch_cells.txt; it is what round 3 had taken for a clock effect).
// A wave with such a row (hull > direct_hull slots) does not stage anything: `direct_rows` lets every lane read ITS
// OWN candidates from global memory (gathers), row by row -- the same candidates in the same order through the same
// arithmetic, so a particle's sums have the same bits whichever way its wave goes.  One copy of that loop per kernel,
// in a branch of its own (a copy per unrolled row cost k_force a spill).
// `key` is the lane's own cell key.  Pre-filter: a hull covers the cells [first key - 1, last key + 1] shifted by the
// row's offset, so a wave whose keys span at most 64 cells (any ordinary wave: 64 particles at ~8 per cell span ~8) has
// hulls of at most 66 cells -- long only at > 30 particles per cell, which is dense, not pathological.  Two v_readlane,
// a subtraction and a compare for those; the nine exact tests only for waves that straddle a row end or are sparse.
__device__ __forceinline__ bool wave_has_long_hull(const Hulls& H, uint32_t key, uint32_t direct_hull) {
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, 0), k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, 63);
    if (k1 - k0 <= 64u && direct_hull >= 128u) return false;        // (thresholds below 128 are test settings: always the exact test)
    bool any = false;
#pragma unroll
    for (int r = 0; r < 9; r++) any = any || (H.B[r] > H.A[r] && H.B[r] - H.A[r] > direct_hull);     // wave-uniform (scalar)
    return any;
}

template <class Row>
__device__ __forceinline__ void direct_rows(const uint2* __restrict__ cells, const GridDesc& g, uint32_t key, bool active,
                                            Row&& row) {
#pragma clang loop unroll(disable)
    for (int r = 0; r < 9; r++) {                                  // (dz, dy) in the order of lane_rows
        uint32_t l0, l1;
        lane_row_one(cells, g, key, active, r / 3 - 1, r % 3 - 1, l0, l1);
        const uint32_t len = l1 > l0 ? l1 - l0 : 0u;
        const uint32_t T = wave_max_u32(len);
        if (T) row(l0, len, T);
    }
}

// Which slots a launch of a pair kernel works on: [lo, hi) minus the hole [gap_lo, gap_lo + gap_len) -- thread t takes
// slot lo + t, shifted up by gap_len from gap_lo on, so ONE launch covers a slab's two boundary layers (the hole is
// its interior) or its interior around the part that was computed earlier.  gap_lo - lo is a multiple of 64: a wave
// never straddles the hole (its lanes stay 64 consecutive slots).  dev != null: {lo, hi} come from device memory
// (the slab step queues the density of its deep interior BEFORE the host knows the layer bounds, csrc/sph_slab.hip);
// the grid is then an upper bound and waves beyond hi leave at once.
struct Targets {
    uint32_t lo, hi, gap_lo, gap_len;
    const uint32_t* dev;
    uint32_t direct_hull;       // rows whose hull is longer are read straight from global memory (see traverse)
    BlockOrder order;           // which block takes which slots (sph_device.hpp)
};

// Which 256 slots a block takes.  The waves that take the direct walk (sparse particles next to a dense layer, see
// traverse) run 2-3x as long as the others and sit where the fluid's free z faces are: at the END of the slot range
// (the last blocks, which start last and then ARE the kernel's tail) or at its beginning.  So the last 1/64 of the
// blocks are dispatched first, then the others in order: on the thin slab whose outer layer is sparse `k_force` 0.302
// -> 0.276 ms and `k_density` 0.135 -> 0.130 (the times of the state without sparse layers), C3 unchanged
// (profiles/r04_rotate_dispatch_experiment.txt).  Dispatching everything backwards does the same for the tail but costs
// C3 3.5 % of wall clock (profiles/r04_reverse_dispatch_experiment.txt).
__device__ __forceinline__ uint32_t pair_block() {
    const uint32_t rot = max(gridDim.x >> 6, 1u);
    return blockIdx.x < rot ? gridDim.x - rot + blockIdx.x : blockIdx.x - rot;
}

__device__ __forceinline__ bool wave_targets(const Targets& T, uint32_t wave, uint32_t lane, uint32_t& i, uint32_t& hi,
                                             uint32_t& first) {
    uint32_t lo = T.lo;
    hi = T.hi;
    if (T.dev) { lo = T.dev[0]; hi = T.dev[1]; }          // wave-uniform (scalar loads)
    first = lo + (ordered_block(pair_block(), gridDim.x, T.order) * (blockDim.x >> 6) + wave) * WAVE;
    if (first >= T.gap_lo) first += T.gap_len;
    i = first + lane;
    return first < hi;
}

// ---- density + pressure (kernelComputeDensities, particleSystem.cu:132-187) ---------------------------
// rho_i = sum_{j in 27 cells, r2 < h2} m * POLY6 * (h2 - r2)^3   (self included)   (.cu:28-37)
// p_i   = max(0, k * (rho_i - rho0))                                              (.cu:15-17)
template <int THREADS>
__global__ __launch_bounds__(THREADS, SPH_DENS_OCC) void k_density(const float4* __restrict__ posi,
                                                                   const uint32_t* __restrict__ keyS,
                                                                   const uint2* __restrict__ cells,
                                                                   float2* __restrict__ dp, float2* __restrict__ cw,
                                                                   Targets tg, GridDesc g, Phys ph) {
    // Round 5: the z coordinates of TWO neighbouring candidates come out of ONE ds_read_b64.  The LDS pipe
    // charges an instruction by its bytes with 8 as the minimum -- a ds_read_b32 costs what a ds_read_b64 costs, 1.1 ns per
    // wave-instruction and CU (profiles/r02_lds_read_rates.txt) -- so {x, y} as a b64 plus z as a b32 paid for 16 bytes per
    // candidate and used 12.  Now a group of four candidates is four b64 {x, y} + two b64 {z, z}: 6 reads instead of 8, on
    // the pipe that is 87 % busy in this kernel.  A lane's range may start at an odd entry: the z array is kept twice,
    // s_z[j] = z_j and s_zo[j] = z_(j+1), and a lane reads pairs from the copy in which ITS first candidate sits at an
    // even index (two more ds_write_b32 per staged piece and lane).  Same candidates, same order, same arithmetic: the
    // sums keep their bits.  (Tried and dropped, records in profiles/: one pad entry per 32 against the bank conflicts of entries
    // 32 apart -- two more instructions per candidate, 1.37 ms against 0.96, r03; {x, y} b64 + z b32 per candidate, the form
    // before this one, r05_density_zz_pairs_ab.txt.)
    static_assert(UNROLL % 2 == 0, "the z-pair walk reads candidates two at a time");
    constexpr int DENS_ENT = lds_ent(THREADS);                      // as LDS_ENT, for this kernel's block
    __shared__ float2 s_xy[DENS_ENT];
    __shared__ __attribute__((aligned(8))) float s_z[DENS_ENT];
    __shared__ __attribute__((aligned(8))) float s_zo_[DENS_ENT + 2];
    float* const s_zo = s_zo_ + 2;                                   // s_zo[-1] exists (the entry in front of the first slice)
    // a range read from device memory: the grid is an upper bound, whole blocks beyond the range leave before the zero-fill
    if (tg.dev && tg.dev[0] + ordered_block(pair_block(), gridDim.x, tg.order) * (uint32_t)THREADS >= tg.dev[1]) return;
    for (uint32_t k = threadIdx.x; k < DENS_ENT; k += THREADS) {   // see LDS_ENT: keep over-reads finite
        s_xy[k] = make_float2(0.f, 0.f);
        s_z[k] = 0.f;
        s_zo_[k] = 0.f;
        if (k < 2u) s_zo_[DENS_ENT + k] = 0.f;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slice = wave * PIECE;
    uint32_t i, tgt_hi, wave_first;
    if (!wave_targets(tg, wave, lane, i, tgt_hi, wave_first)) return;      // wave-uniform; no barrier below
    const bool active = i < tgt_hi;
    const uint32_t ii = active ? i : tgt_hi - 1;
    const float4 pi = posi[ii];
    Rows R;
    const uint32_t my_key = keyS[ii];
    lane_rows(cells, g, my_key, active, R);
    Hulls H;
    wave_hulls(R, H);
    float4 q0, q1;
    float acc = 0.f;
    const float h2_v = in_vgpr(ph.h2);
    // one candidate; the staged walk and the direct walk share it, so a row gives the same bits either way
    auto pair_math = [&](float x, float y, float z, bool valid) {
        const float dx = pi.x - x, dy = pi.y - y, dz = pi.z - z;
        // h^2 - r^2 in three fmas (the subtraction rides along); max(., 0) is the r < h test
        float d = fmaxf(fmaf(-dz, dz, fmaf(-dy, dy, fmaf(-dx, dx, h2_v))), 0.f);
        d = valid ? d : 0.f;
        acc = fmaf(d * d, d, acc);
    };
    auto finish = [&]() {
        if (active) {
            float rho = acc * ph.poly6_mass;
            float p = fmaxf(0.f, ph.gas_constant * (rho - ph.rest_density));
            dp[i] = make_float2(rho, p);
            cw[i] = neighbour_terms(ph, rho, p);
        }
    };
    if (wave_has_long_hull(H, my_key, tg.direct_hull)) {          // a branch of its own, to its own end: the staged walk keeps its registers
        const uint32_t me = active ? i : tgt_hi - 1u;        // (= ii, derived again: nothing of this branch stays alive in the staged walk)
        direct_rows(cells, g, keyS[me], active, [&](uint32_t l0, uint32_t len, uint32_t T) {      // a lane out of range reads itself (masked)
            for (uint32_t t = 0; t < T; t += 2u) {
                float4 q[2];
#pragma unroll
                for (uint32_t u = 0; u < 2u; u++) q[u] = posi[t + u < len ? l0 + t + u : me];
#pragma unroll
                for (uint32_t u = 0; u < 2u; u++) pair_math(q[u].x, q[u].y, q[u].z, t + u < len);
            }
        });
        finish();
        return;
    }
    traverse(
        H,
        [&](uint32_t a) {   // the arrays are padded by 2*PIECE entries: no bounds predicate needed
            q0 = posi[a + lane];
            q1 = posi[a + WAVE + lane];
        },
        [&]() {
            s_xy[slice + lane] = make_float2(q0.x, q0.y);
            s_z[slice + lane] = q0.z;
            s_xy[slice + WAVE + lane] = make_float2(q1.x, q1.y);
            s_z[slice + WAVE + lane] = q1.z;
            // the shifted copy; entry slice - 1 belongs to the slice in front (or to the pad): only a lane's masked
            // over-read ever looks at a slice's last shifted entry, and whatever stands there is a finite z
            s_zo[(int)(slice + lane) - 1] = q0.z;
            s_zo[slice + WAVE + lane - 1u] = q1.z;
        },
        [&](int r, uint32_t a, uint32_t b) {
            const uint32_t l0 = max(R.lo[r], a), l1 = min(R.hi[r], b);
            const uint32_t len = l1 > l0 ? l1 - l0 : 0u;
            const uint32_t rel = len ? l0 - a : 0u;
            uint32_t idx = slice + rel;
            // every lane has at least tmin candidates: that part of the walk needs no per-lane range test
            const uint32_t tmin = wave_min_u32_uniform_first(len) & ~(uint32_t)(UNROLL - 1);
            // z pairs: candidates (rel, rel + 1), (rel + 2, rel + 3), ... -- from s_z when rel is even, else from the copy
            // shifted by one, where candidate rel sits at the even index rel - 1
            lds_v2f_ptr zp = (lds_v2f_ptr)((rel & 1u) ? (const float*)s_zo : (const float*)s_z) + ((slice + rel) >> 1);
            auto group = [&](uint32_t t, bool check) {
#pragma unroll
                for (int u = 0; u < UNROLL; u += 2) {
                    const v2f xy0 = ((lds_v2f_ptr)s_xy)[idx + u], xy1 = ((lds_v2f_ptr)s_xy)[idx + u + 1];
                    const v2f zz = zp[u >> 1];
                    pair_math(xy0.x, xy0.y, zz.x, !check || t + u < len);
                    pair_math(xy1.x, xy1.y, zz.y, !check || t + u + 1 < len);
                }
                idx += UNROLL;
                zp += UNROLL / 2;
            };
            uint32_t t = 0;
            for (; t < tmin; t += UNROLL) group(t, false);
            for (; __ballot(t < len) != 0ull; t += UNROLL) group(t, true);    // until every lane is through its range
        });
    finish();
}

// ---- mixed precision (BASELINE config 5): fp32 positions, fp16 neighbour accumulators -----------------------
// Same traversal, same candidates, same order; what changes is the arithmetic of a PAIR.  Candidates are staged
// in LDS as fp16 coordinates relative to a wave-uniform reference point and in units of h (so |x'| <~ 4 and
// r'^2 = r^2 / h^2 comes out of the packed arithmetic directly), two neighbouring candidates per 32-bit word:
// every v_pk_* instruction works on TWO candidates of the lane's range and every LDS read fetches two.  The sum of
// the NORMALISED kernel W' = (1 - r'^2)^3 in [0, 1] is accumulated in packed fp16 over one row (<= ~30 terms of at
// most 1: the raw densities, ~2e6, would overflow fp16) and added to an fp32 total after every
// row; the physical scale m * POLY6 * h^6 is applied once, in fp32.  Tolerance: DESIGN.md section 4 (mixed).
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) h2* lds_h2_ptr;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) h4* lds_h4_ptr;

__device__ __forceinline__ h2 h2_splat(float x) { const _Float16 v = (_Float16)x; return h2{v, v}; }
// 1.0 in the halves that are valid, 0.0 in the others
__device__ __forceinline__ h2 h2_mask(bool lo, bool hi) {
    const uint32_t bits = (lo ? 0x3C00u : 0u) | (hi ? 0x3C000000u : 0u);
    return __builtin_bit_cast(h2, bits);
}

constexpr int HPAIRS = PIECE / 2;                       // packed candidate pairs per staged piece
constexpr int LDS_PAIRS = (PAIR_WAVES + 1) * HPAIRS + 8;
constexpr int HUNROLL = 4;                              // pairs per unrolled group (8 candidates)

// A lane's range [l0, l1) may start at an odd candidate.  So that its pairs are whole all the same, every piece is
// staged TWICE: pairs (2p, 2p+1) in the lower half of the LDS arrays and pairs (2p+1, 2p+2) in the upper half; a
// lane reads the copy that matches the parity of its l0.  Only the LAST pair of a range can then be half outside it
// (odd length) -- one masked iteration per row instead of a mask on every pair.
struct PairWalk {
    uint32_t idx;        // LDS pair index of the first pair
    uint32_t whole;      // pairs entirely inside the range
    bool odd;            // one more candidate in the low half of pair `whole`
    uint32_t kmin, T;    // wave-uniform: pairs every lane has / pairs of the longest lane
};

__device__ __forceinline__ PairWalk pair_walk(uint32_t slice_pairs, uint32_t l0, uint32_t l1) {
    PairWalk w;
    const uint32_t len = l1 > l0 ? l1 - l0 : 0u;
    w.idx = (len && (l0 & 1u) ? (uint32_t)LDS_PAIRS : 0u) + slice_pairs + (len ? l0 >> 1 : 0u);
    w.whole = len >> 1;
    w.odd = (len & 1u) != 0u;
    uint32_t tmin;
    wave_min_max_u32(w.whole + (w.odd ? 1u : 0u), tmin, w.T);
    w.kmin = wave_min_u32_uniform_first(w.whole) & ~(uint32_t)(HUNROLL - 1);
    return w;
}

#ifndef SPH_DENSH_OCC
#define SPH_DENSH_OCC 5          // <= 96 VGPRs: at 6 waves (80) the pass loop spills 28 registers, 1.31 ms against 1.16
#endif
#ifndef SPH_MIXED_SPAN
#define SPH_MIXED_SPAN 6.0f      // h: how far from the reference a target may lie for the plain packed walk
#endif
#ifndef SPH_MIXED_PASSES
#define SPH_MIXED_PASSES 3       // reference points tried on a far-apart wave before the rest takes the fp32 walk
#endif

// The packed arithmetic works on coordinates RELATIVE to a reference point, and fp16 carries 11 bits: that is 2^-9 h while
// everything lies within ~4 h of the reference, 2^-8 h up to 8 h -- and nothing when the particles are far apart.  An
// ordinary wave is 64 sorted particles = ~8 cells of one x-row = 5 h, and rounds 3 and 4 took the wave's first particle as the
// reference for everybody.  That is wrong for a wave that straddles the end of an x-row (its second half starts again at the
// other side of the fluid, tens of h away), a wave in thin fluid (fewer particles per cell, more cells per wave: wide in x;
// together 7.5 % of the waves of the flowing C3 dam are wider than 8 h in x), one across the end of a cell layer (the other
// side in y; 130 waves), or one that holds the scattered particles of a nearly empty region.  Round 5 found densities 33 %
// off on such waves (config 5 cut into slabs against one context: the cuts change which particles share a wave).  Now:
//   x is carried as a coarse part (a multiple of h/2: exact in fp16 up to 1024 h) plus a fine part (|.| <= h/4):
//     dx = (txh - xh) + (txl - xl), two more packed instructions per pair, right at any width;
//   y and z by PASSES: the first lane not yet served gives the reference, the lanes within MIXED_SPAN h of it in y and z
//     are served by this pass -- their row ranges staged and walked as ever, everybody else's ranges empty -- and the rest wait
//     for the next (one pass for an ordinary wave, two for a wave across a layer's end); after MIXED_PASSES passes the lanes
//     still waiting are scattered particles with short candidate lists: they gather their own candidates and sum in fp32
//     (k_density's arithmetic), all in one last walk.
// The price, and what else was tried (profiles/r05_mixed_wide_waves_ab.txt; flowing C3, round 4's kernel 1.08 ms): this form
// 1.31 ms.  Every far-apart wave down the fp32 gather walk: a wave of DENSE fluid takes ~0.2 ms in it and the 130 waves at
// the layer ends became the kernel's tail, 1.66 ms.  Passes in x too instead of the split: a thin-fluid wave needs 4-7 of
// them, 1.47 ms.  Round 4's walk kept for the waves within 6 h of their first particle and this one for the rest, in one
// kernel: 1.39 ms -- slower than this one for everybody; the two inlined walks no longer share the instruction cache (the
// same kernel with the second walk never taken: 1.17 ms).
__global__ __launch_bounds__(PAIR_THREADS, SPH_DENSH_OCC) void k_density_h(const float4* __restrict__ posi,
                                                                           const uint32_t* __restrict__ keyS,
                                                                           const uint2* __restrict__ cells,
                                                                           float2* __restrict__ dp, float2* __restrict__ cw,
                                                                           Targets tg, GridDesc g, Phys ph) {
    struct XY { h2 x, y; };                                   // coarse x, y: one ds_read_b64 per pair
    struct ZL { h2 z, xl; };                                  // z, fine x: another
    __shared__ XY s_xy[2 * LDS_PAIRS];
    __shared__ ZL s_zl[2 * LDS_PAIRS];
    const h2 zero = h2{(_Float16)0, (_Float16)0}, one = h2{(_Float16)1, (_Float16)1};
    for (uint32_t k = threadIdx.x; k < 2 * LDS_PAIRS; k += PAIR_THREADS) {      // over-reads stay finite
        s_xy[k].x = zero; s_xy[k].y = zero; s_zl[k].z = zero; s_zl[k].xl = zero;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slice = wave * HPAIRS;
    uint32_t i, tgt_hi, wave_first;
    if (!wave_targets(tg, wave, lane, i, tgt_hi, wave_first)) return;      // wave-uniform; no barrier below
    const bool active = i < tgt_hi;
    const uint32_t ii = active ? i : tgt_hi - 1;
    const float4 pi = posi[ii];
    const uint32_t my_key = keyS[ii];
    const float inv_h = 1.0f / ph.h;
    float4 q0, q1, q2;
    float acc = 0.f;          // sum of the NORMALISED kernel (1 - r'^2)^3 over the rows walked in packed fp16
    float acc32 = 0.f;        // sum of (h^2 - r^2)^3 over the candidates of the lanes left to the fp32 walk

    // the lanes `in`, relative to (rx, ry, rz): staged hull walk, or the per-lane gather when a hull is long
    auto serve = [&](bool in, float rx, float ry, float rz) {
        const float X = (pi.x - rx) * inv_h;
        const float Xc = rintf(X + X) * 0.5f;
        const h2 tx = h2_splat(Xc), txl = h2_splat(X - Xc);
        const h2 ty = h2_splat((pi.y - ry) * inv_h), tz = h2_splat((pi.z - rz) * inv_h);
        // one PAIR of candidates (the two halves of every operand)
        auto pair_math = [&](h2& row, h2 x, h2 xl, h2 y, h2 z, h2 m, bool masked) {
            const h2 dx = (tx - x) + (txl - xl), dy = ty - y, dz = tz - z;
            h2 t = one - dx * dx;                                  // 1 - r'^2: three v_pk_fma_f16
            t = t - dy * dy;
            t = t - dz * dz;
            t = __builtin_elementwise_max(t, zero);                // r' < 1  <=>  r < h
            if (masked) t = t * m;
            row = row + (t * t) * t;
        };
        auto x_parts = [&](float qx, _Float16& hi, _Float16& lo) {
            const float v = (qx - rx) * inv_h;
            const float c = rintf(v + v) * 0.5f;
            hi = (_Float16)c;
            lo = (_Float16)(v - c);
        };
        Rows R;
        lane_rows(cells, g, my_key, in, R);
        Hulls H;
        wave_hulls(R, H);
        if (wave_has_long_hull(H, my_key, tg.direct_hull)) {
            direct_rows(cells, g, keyS[ii], in, [&](uint32_t l0, uint32_t len, uint32_t) {        // every lane gathers its own candidates, two at a time
                const uint32_t T = wave_max_u32((len + 1u) >> 1);
                h2 row = zero;
                for (uint32_t k = 0; k < T; k++) {
                    const bool v0 = 2u * k < len, v1 = 2u * k + 1u < len;
                    const float4 a = posi[v0 ? l0 + 2u * k : ii], b = posi[v1 ? l0 + 2u * k + 1u : ii];
                    _Float16 xa, xb, la, lb;
                    x_parts(a.x, xa, la);
                    x_parts(b.x, xb, lb);
                    const h2 x = h2{xa, xb}, xl = h2{la, lb};
                    const h2 y = h2{(_Float16)((a.y - ry) * inv_h), (_Float16)((b.y - ry) * inv_h)};
                    const h2 z = h2{(_Float16)((a.z - rz) * inv_h), (_Float16)((b.z - rz) * inv_h)};
                    pair_math(row, x, xl, y, z, h2_mask(v0, v1), true);
                }
                acc += (float)row.x + (float)row.y;
            });
            return;
        }
        traverse(
            H,
            [&](uint32_t a) {   // lane L stages candidates 2L, 2L+1 and 2L+2; the arrays are padded by 2*PIECE
                q0 = posi[a + 2u * lane];
                q1 = posi[a + 2u * lane + 1u];
                q2 = posi[a + 2u * lane + 2u];
            },
            [&]() {
                _Float16 x0, x1, x2, l0, l1, l2;
                x_parts(q0.x, x0, l0);
                x_parts(q1.x, x1, l1);
                x_parts(q2.x, x2, l2);
                const _Float16 y0 = (_Float16)((q0.y - ry) * inv_h), y1 = (_Float16)((q1.y - ry) * inv_h),
                               y2 = (_Float16)((q2.y - ry) * inv_h);
                const _Float16 z0 = (_Float16)((q0.z - rz) * inv_h), z1 = (_Float16)((q1.z - rz) * inv_h),
                               z2 = (_Float16)((q2.z - rz) * inv_h);
                s_xy[slice + lane].x = h2{x0, x1}; s_xy[slice + lane].y = h2{y0, y1};
                s_zl[slice + lane].z = h2{z0, z1}; s_zl[slice + lane].xl = h2{l0, l1};
                s_xy[LDS_PAIRS + slice + lane].x = h2{x1, x2}; s_xy[LDS_PAIRS + slice + lane].y = h2{y1, y2};
                s_zl[LDS_PAIRS + slice + lane].z = h2{z1, z2}; s_zl[LDS_PAIRS + slice + lane].xl = h2{l1, l2};
            },
            [&](int r, uint32_t a, uint32_t b) {
                const uint32_t l0 = max(R.lo[r], a), l1 = min(R.hi[r], b);
                const PairWalk w = pair_walk(slice, l1 > l0 ? l0 - a : 0u, l1 > l0 ? l1 - a : 0u);
                uint32_t idx = w.idx;
                h2 row = zero;                                             // this row's sum of W', both halves
                auto pair = [&](int u, h2 m, bool masked) {
                    const h4 xy = ((lds_h4_ptr)s_xy)[idx + u], zl = ((lds_h4_ptr)s_zl)[idx + u];
                    pair_math(row, __builtin_shufflevector(xy, xy, 0, 1), __builtin_shufflevector(zl, zl, 2, 3),
                              __builtin_shufflevector(xy, xy, 2, 3), __builtin_shufflevector(zl, zl, 0, 1), m, masked);
                };
                uint32_t k = 0;
                for (; k < w.kmin; k += HUNROLL) {
#pragma unroll
                    for (int u = 0; u < HUNROLL; u++) pair(u, one, false);
                    idx += HUNROLL;
                }
                for (; k < w.T; k++) {                                     // until every lane is through its range
                    pair(0, h2_mask(k < w.whole + (w.odd ? 1u : 0u), k < w.whole), true);
                    idx++;
                }
                acc += (float)row.x + (float)row.y;
            });
    };

    constexpr float MIXED_SPAN = SPH_MIXED_SPAN;
    uint64_t todo = __ballot(active);
    for (int pass = 0; todo != 0ull; pass++) {                // wave-uniform
        if (pass == SPH_MIXED_PASSES) {
            const bool in = ((todo >> lane) & 1ull) != 0ull;
            const float h2_v = in_vgpr(ph.h2);
            direct_rows(cells, g, keyS[ii], in, [&](uint32_t l0, uint32_t len, uint32_t T) {      // a lane out of range reads itself (masked)
                for (uint32_t t = 0; t < T; t += 2u) {
                    float4 q[2];
#pragma unroll
                    for (uint32_t u = 0; u < 2u; u++) q[u] = posi[t + u < len ? l0 + t + u : ii];
#pragma unroll
                    for (uint32_t u = 0; u < 2u; u++) {
                        const float dx = pi.x - q[u].x, dy = pi.y - q[u].y, dz = pi.z - q[u].z;
                        float d = fmaxf(fmaf(-dz, dz, fmaf(-dy, dy, fmaf(-dx, dx, h2_v))), 0.f);
                        d = t + u < len ? d : 0.f;
                        acc32 = fmaf(d * d, d, acc32);
                    }
                }
            });
            break;
        }
        const int lead = __builtin_ctzll(todo);
        const float rx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pi.x), lead)),
                    ry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pi.y), lead)),
                    rz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pi.z), lead));
        const float far = fmaxf(fabsf(pi.y - ry), fabsf(pi.z - rz)) * inv_h;
        const bool in = ((todo >> lane) & 1ull) != 0ull && far <= MIXED_SPAN;      // (the lead lane: far = 0)
        todo &= ~__ballot(in);
        serve(in, rx, ry, rz);
    }
    if (active) {
        const float h2f = ph.h2;
        // m POLY6 h^6 sum (1 - r'^2)^3 for the packed walks + m POLY6 sum (h^2 - r^2)^3 for the fp32 walk
        const float rho = fmaf(acc, ph.poly6_mass * (h2f * h2f * h2f), acc32 * ph.poly6_mass);
        const float p = fmaxf(0.f, ph.gas_constant * (rho - ph.rest_density));
        dp[i] = make_float2(rho, p);
        cw[i] = neighbour_terms(ph, rho, p);
    }
}

int launch_density_range(sph_ctx* c, uint32_t lo, uint32_t hi);
int launch_density(sph_ctx* c) {
    if (c->n == 0) return SPH_OK;
    return launch_density_range(c, c->own_off, c->own_off + c->n);
}

// ---- integrate one particle (kernelIntegrate, particleSystem.cu:375-420) ------------------------------
// a = (f_press + f_visc + (0, g*G*rho, 0)) / rho ; v += dt*a + dv ; x += dt*v ; walls per axis X,Y,Z
// (lower wall first): x = wall +- eps, v *= -0.75 ; out[index] = (x, y, z, 1).
__device__ __forceinline__ void wall(float& x, float& v, float lo, float hi, float eps, float damp) {
    if (x - eps < lo) { x = lo + eps; v *= damp; }
    if (x + eps > hi) { x = hi - eps; v *= damp; }
}

__device__ __forceinline__ void integrate_one(const Phys& ph, float dt, float4& pi, float4& vi, float rho, float fx,
                                              float fy, float fz, float dvx, float dvy, float dvz) {
    fy += ph.gravity_y * rho;
    // one v_rcp_f32 (1 ulp) and three multiplications instead of three IEEE divisions (~10 instructions each): the
    // acceleration moves by <= 1.5 ulp, five orders below the stated tolerance.  -DSPH_EXACT_DIV=1 builds the reference's
    // divisions (f / rho here, -dv / (m (1 + count)) in the collision epilogue) for A/B runs of long free trajectories
    // (profiles/r05_long_run_parity_rcp_vs_exact_div.txt: the two builds leave the oracle at the same step, by the same amount)
#if SPH_EXACT_DIV
    float ax = fx / rho, ay = fy / rho, az = fz / rho;
#else
    const float ir = __builtin_amdgcn_rcpf(rho);
    float ax = fx * ir, ay = fy * ir, az = fz * ir;
#endif
    vi.x += dt * ax + dvx;
    vi.y += dt * ay + dvy;
    vi.z += dt * az + dvz;
    pi.x += dt * vi.x;
    pi.y += dt * vi.y;
    pi.z += dt * vi.z;
    // (a wave whose 64 particles are all clear of the walls -- nearly every wave -- skips the six tests' selects)
    const float e = ph.wall_eps;
    const bool near_wall = pi.x - e < ph.box_min[0] || pi.x + e > ph.box_max[0] || pi.y - e < ph.box_min[1] || pi.y + e > ph.box_max[1] ||
                           pi.z - e < ph.box_min[2] || pi.z + e > ph.box_max[2];
    if (__ballot(near_wall) != 0ull) {
        wall(pi.x, vi.x, ph.box_min[0], ph.box_max[0], ph.wall_eps, ph.wall_damping);
        wall(pi.y, vi.y, ph.box_min[1], ph.box_max[1], ph.wall_eps, ph.wall_damping);
        wall(pi.z, vi.z, ph.box_min[2], ph.box_max[2], ph.wall_eps, ph.wall_damping);
    }
}

// ---- force / collision / integrate in ONE neighbour traversal ------------------------------------------
// kernelComputeForces (.cu:189-243, pair .cu:39-50):
//   f_press_i += -r^_ij * m * (p_i + p_j) / (2 rho_j) * SPIKY_GRAD * (h - r)^2,  r < h
//   f_visc_i  += VISC * m * (v_j - v_i) / rho_j * VISC_LAP * (h - r)
//   a zero r_ij normalises to zero (Eigen Dot.h:124-134): self / coincident pairs add no pressure.
// kernelComputeCollisions (.cu:245-300, pair .cu:52-65): j != i, d <= 2R and r.v < 0:
//   dv += m (1 + e) (r.v / d^2) r, count++ ; finally dv = -dv / (m (1 + count)).
// The reference runs three separate traversals plus an integrate pass; FORCE, COLL and INTEG select
// what this instantiation does so that the phase API can still run them one at a time.
#ifdef SPH_PAIR_STATS
__device__ unsigned long long g_pair_stats[12];  // debug build only (-DSPH_PAIR_STATS): waves, pieces, walk length, chunks, collision rounds,
                                                 // waves with a hull > 256 / 512 / 2048 slots, [8] sum over waves of the busiest LANE's collision
                                                 // candidates (what a per-particle queue would have to work off), [9] sum over all lanes of them
__device__ int g_pair_stats_on;
#define PAIR_STAT(k, v) do { if (lane == 0 && g_pair_stats_on) atomicAdd(&g_pair_stats[k], (unsigned long long)(v)); } while (0)
extern "C" void sph_debug_pair_stats(unsigned long long* out, int on) {     // read + clear the counters, then count or not
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stats), sizeof(g_pair_stats));
    unsigned long long z[12] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stats), z, sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stats_on), &on, sizeof(on));
}
#else
#define PAIR_STAT(k, v) do { } while (0)
#endif
template <bool FORCE, bool COLL, bool INTEG, int THREADS>
__global__ __launch_bounds__(THREADS, SPH_FORCE_OCC) void k_force(
    const float4* __restrict__ posi, const float4* __restrict__ velr, const float2* __restrict__ dp,
    const float2* __restrict__ cw, const uint32_t* __restrict__ keyS, const uint2* __restrict__ cells, float4* __restrict__ fpress,
    float4* __restrict__ fvisc, float4* __restrict__ dvel, float4* __restrict__ posi_out,
    float4* __restrict__ velr_out, float4* __restrict__ pos_by_index, uint32_t* __restrict__ keys_out,
    uint64_t* __restrict__ mm_mask, uint32_t* __restrict__ mm_tile_cnt, Targets tg,
    uint32_t slot0, float dt, GridDesc g, Phys ph) {
    // One candidate = 4 float2 {x,y} {z,vx} {vy,vz} {cp,w} at a 40-byte stride (5 float2, the fifth is
    // padding): one address register serves all four ds_read_b64 through immediate offsets, and the
    // 8-entry (one cell) distance between the lane groups of a wave is 80 dwords = 16 banks, so the four
    // distinct addresses of a half-wave never share a bank.
    // (Round 5 tried the same 32 bytes as TWO aligned ds_read_b128 at a 48-byte stride: LDS instructions halve, SQ_BUSY_CYCLES +2.6 %,
    // the same 2.56 ms -- at 8 particles per cell every 16-byte-aligned stride is a two-way bank conflict.  Not kept:
    // profiles/r05_force_b128_experiment.txt.)
    constexpr int E2 = 5;
    constexpr int ENT = lds_ent(THREADS);
    __shared__ float2 s_e[ENT * 5];
    // a range read from device memory: the grid is an upper bound, whole blocks beyond the range leave before the zero-fill
    if (tg.dev && tg.dev[0] + ordered_block(pair_block(), gridDim.x, tg.order) * (uint32_t)THREADS >= tg.dev[1]) return;
    for (uint32_t k = threadIdx.x; k < ENT * E2; k += THREADS)   // see LDS_ENT: keep over-reads finite
        s_e[k] = make_float2(0.f, 0.f);
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slice = wave * PIECE;
    uint32_t i, tgt_hi, wave_first;
    if (!wave_targets(tg, wave, lane, i, tgt_hi, wave_first)) return;      // wave-uniform; no barrier below
    const bool active = i < tgt_hi;
    const uint32_t ii = active ? i : tgt_hi - 1;
    float4 pi = posi[ii];
    float4 vi = velr[ii];
    const float2 dpi = dp[ii];
    Rows R;
    const uint32_t my_key = keyS[ii];
    lane_rows(cells, g, my_key, active, R);
    Hulls H;
    wave_hulls(R, H);
    PAIR_STAT(0, 1);
#ifdef SPH_PAIR_STATS
    {   // longest hull of the wave: how many waves would take the direct walk at which threshold
        uint32_t hmax = 0;
        for (int r = 0; r < 9; r++) hmax = max(hmax, H.B[r] > H.A[r] ? H.B[r] - H.A[r] : 0u);
        PAIR_STAT(5, hmax > 256u ? 1 : 0); PAIR_STAT(6, hmax > 512u ? 1 : 0); PAIR_STAT(7, hmax > 2048u ? 1 : 0);
    }
#endif
    float4 q0, q1, w0, w1;
    float2 e0, e1;
    const float cpi = ph.cp_scale * dpi.y;                 // (the lane's own cp: as neighbour_terms computes it)
    const float h_v = in_vgpr(ph.h);
    // Collision range, two stages.  The candidate loop only needs a cheap SUPERSET: its r2 comes out of three
    // fused multiply-adds and differs from the reference's x*x + (y*y + z*z) by a few ulps, so the mask bit is
    // the sign of r2 - (coll_dist2 + 8 ulps), shifted in by one v_alignbit (two all-VGPR instructions; a compare
    // + add-with-carry goes through an SGPR pair).  The EXACT predicate of computeCollision -- the reference's r2
    // in the reference's operation order, unfused, against the exact threshold coll_dist2 (see derive()) -- is
    // applied to the few masked candidates in the drain loop below.
    const float coll_next_v = in_vgpr(__uint_as_float(__float_as_uint(ph.coll_dist2) + 8u));
    // viscosity: sum_j w_j (v_j - v_i) is accumulated as sum_j w_j v_j and sum_j w_j; v_i sum_j w_j comes off once per
    // particle in the epilogue -- 3 fma + 1 add per candidate instead of 3 sub + 3 fma.  (The sum's terms are |v| / |dv|
    // times larger than those of the differences; at dt = 5e-7 that moves the new velocity by ~1e-9 |v|max, four orders
    // below the stated tolerance, and the force arrays by ~1e-7 of the largest force.  The pressure term is NOT split
    // this way: its r_ij would cancel ~300-fold.)
    float fpx = 0.f, fpy = 0.f, fpz = 0.f, fvx = 0.f, fvy = 0.f, fvz = 0.f, sw = 0.f;
    float cvx = 0.f, cvy = 0.f, cvz = 0.f;
    uint32_t ccount = 0;
#ifdef SPH_PAIR_STATS
    uint32_t stat_near = 0;
#endif
    // One candidate (position q, velocity u, cp_j, w_j): the pressure / viscosity sums; returns r2 - (collision range + 8
    // ulps), whose sign is the candidate's bit of the collision SUPERSET.  Shared by the staged walk and the direct walk
    // (traverse), so a row gives the same bits either way.
    auto pair_math = [&](float qx, float qy, float qz, float ux, float uy, float uz, float cpj, float wj, bool valid) -> float {
        const float dx = pi.x - qx, dy = pi.y - qy, dz = pi.z - qz;
        // r^2 + 1e-30: the tiny term rides in the first fma for free and is far below one ulp of any
        // r^2 that matters.  r = 0 (the particle itself, coincident particles): 1/r is capped at 1e15,
        // r*1/r = 0, and the huge but finite pressure weight multiplies r_ij = 0 -- no pressure term,
        // as with Eigen's normalized() of a zero vector (Dot.h:124-134); the viscous term is exact.
        float r2 = fmaf(dz, dz, fmaf(dx, dx, fmaf(dy, dy, 1e-30f)));
        // a lane beyond its range (the tail of the walk): ONE select makes the candidate infinitely far away --
        // h - r clamps to 0 (no force, no viscosity weight) and r2 - coll is positive (no collision bit)
        r2 = valid ? r2 : 1e30f;
        if (FORCE) {
            const float rinv = inv_sqrt(r2);
            // both kernels vanish continuously at r = h, so "r < h" is max(h - r, 0): one v_max instead
            // of a compare and two selects
            const float hr = fmaxf(fmaf(-r2, rinv, h_v), 0.f);
            const float w = wj * hr;                                // VISC m VISC_LAP (h-r) / rho_j
            const float s = (cpi + cpj) * w * (hr * rinv);          // m (p_i+p_j)/(2 rho_j) 45/(pi h^6) (h-r)^2 / r
            fpx += s * dx; fpy += s * dy; fpz += s * dz;
            fvx += w * ux; fvy += w * uy; fvz += w * uz;            // sum w_j v_j  (v_i sum w_j: epilogue)
            sw += w;
        }
        return r2 - coll_next_v;                                    // negative <=> within collision range
    };
    // One candidate of the collision superset, decided EXACTLY.
    // computeCollision (particleSystem.cu:52-65) decides on dij = sqrtf(rij.squaredNorm()) <= 2R
    // and rij.dot(vij) < 0, Eigen reducing a 3-vector as a0 + (a1 + a2) with one rounding per
    // operation.  Both predicates are evaluated exactly that way (no contraction), so that on
    // identical inputs the same pairs collide as in the reference: sqrtf is monotone, hence
    // dij <= 2R  <=>  r2 <= coll_dist2 (the largest float whose root is <= 2R, derive()).
    // j == i needs no test: r_ij = 0 gives r.v = -0, which is not < 0 (the reference skips
    // the pair by index, :54; a coincident pair fails r.v < 0 there too).
    auto collide_math = [&](float qx, float qy, float qz, float wx, float wy, float wz) {
        const float dx = pi.x - qx, dy = pi.y - qy, dz = pi.z - qz;
        const float ux = wx - vi.x, uy = wy - vi.y, uz = wz - vi.z;
        float r2c, dot;
        {
#pragma clang fp contract(off)
            r2c = dx * dx + (dy * dy + dz * dz);
            dot = -(dx * ux + (dy * uy + dz * uz));                 // r_ij . (v_i - v_j)
        }
        const bool hit = r2c <= ph.coll_dist2 && dot < 0.f;
        // (one v_rcp_f32 of r2, 1 ulp, for the reference's sqrtf and division dot / (dij * dij): the PREDICATES above are exact
        // either way; with the exact form the free-run numbers were identical to 9 digits)
        const float cfac = hit ? ph.coll_mass * dot * __builtin_amdgcn_rcpf(r2c) : 0.f;
        cvx += cfac * dx; cvy += cfac * dy; cvz += cfac * dz;
        ccount += hit ? 1u : 0u;
    };
    auto finish = [&]() {
#ifdef SPH_PAIR_STATS
        if (COLL) {
            const uint32_t busiest = wave_max_u32(stat_near);          // (all lanes: the macro's body runs in lane 0 only)
            PAIR_STAT(8, busiest);
            uint32_t sum = stat_near;
            for (int off = 32; off; off >>= 1) sum += (uint32_t)__shfl_xor((int)sum, off);
            PAIR_STAT(9, sum);
        }
#endif
        if (FORCE) { fvx = fmaf(-vi.x, sw, fvx); fvy = fmaf(-vi.y, sw, fvy); fvz = fmaf(-vi.z, sw, fvz); }
        bool moved = false;
        if (active) {
            float dvx = 0.f, dvy = 0.f, dvz = 0.f;
            if (COLL) {
#if SPH_EXACT_DIV
                const float den = ph.mass * (float)(1u + ccount);
                dvx = -cvx / den; dvy = -cvy / den; dvz = -cvz / den;
#else
                const float nid = -__builtin_amdgcn_rcpf(ph.mass * (float)(1u + ccount));       // one v_rcp_f32 for the three components
                dvx = cvx * nid; dvy = cvy * nid; dvz = cvz * nid;
#endif
            }
            if (INTEG) {
                integrate_one(ph, dt, pi, vi, dpi.x, fpx + fvx, fpy + fvy, fpz + fvz, dvx, dvy, dvz);
                posi_out[i] = pi;
                velr_out[i] = vi;
                if (pos_by_index) pos_by_index[__float_as_uint(pi.w)] = make_float4(pi.x, pi.y, pi.z, 1.0f);
                // the next step's cell hash (kernelGetZIndex) while the new position is still in registers
                const uint32_t key = cell_key(g, pi.x, pi.y, pi.z);
                keys_out[i - slot0] = key;                      // slot0: first owned slot (the launch may cover a sub-range)
                if (mm_mask) moved = key != keyS[i];
            } else {
                if (FORCE) {
                    fpress[i] = make_float4(fpx, fpy, fpz, 0.f);
                    fvisc[i] = make_float4(fvx, fvy, fvz, 0.f);
                }
                if (COLL) dvel[i] = make_float4(dvx, dvy, dvz, __uint_as_float(ccount));
            }
        }
        if (INTEG && mm_mask) {
            // movers of the next sort (sph_sort.hip: the merge path), one bit per slot: this wave IS one 64-slot chunk
            // (a sub-range launch starts on a chunk boundary: lo - slot0, gap_lo - lo and gap_len are multiples of 64)
            const uint64_t m = __ballot(moved);
            const uint32_t chunk = (wave_first - slot0) >> 6;
            if (lane == 0) {
                mm_mask[chunk] = m;
                if (m) atomicAdd(&mm_tile_cnt[chunk / MM_TILE_CHUNKS], (uint32_t)__popcll(m));
            }
        }
    };
    if (wave_has_long_hull(H, my_key, tg.direct_hull)) {          // a branch of its own, to its own end: the staged walk keeps its registers
        const uint32_t me = active ? i : tgt_hi - 1u;        // (= ii, derived again: nothing of this branch stays alive in the staged walk)
        direct_rows(cells, g, keyS[me], active, [&](uint32_t l0, uint32_t len, uint32_t T) {
            for (uint32_t t = 0; t < T; t++) {
                const bool valid = t < len;
                const uint32_t j = valid ? l0 + t : me;              // a lane out of range reads itself (finite values, masked)
                const float4 q = posi[j], w = velr[j];
                float2 e = make_float2(0.f, 0.f);
                if (FORCE) e = cw[j];                                // cp_j, w_j
                const float tt = pair_math(q.x, q.y, q.z, w.x, w.y, w.z, e.x, e.y, valid);
                if (COLL && __ballot(tt < 0.f) != 0ull) {            // candidate order, as the staged walk's work-off
                    if (tt < 0.f) collide_math(q.x, q.y, q.z, w.x, w.y, w.z);
                }
            }
        });
        finish();
        return;
    }
    traverse(
        H,
        [&](uint32_t a) {
            q0 = posi[a + lane]; q1 = posi[a + WAVE + lane];
            w0 = velr[a + lane]; w1 = velr[a + WAVE + lane];
            if (FORCE) { e0 = cw[a + lane]; e1 = cw[a + WAVE + lane]; }
        },
        [&]() {
            float2* e0p = &s_e[(slice + lane) * 5];
            float2* e1p = &s_e[(slice + WAVE + lane) * 5];
            e0p[0] = make_float2(q0.x, q0.y);
            e0p[1] = make_float2(q0.z, w0.x);
            e0p[2] = make_float2(w0.y, w0.z);
            e1p[0] = make_float2(q1.x, q1.y);
            e1p[1] = make_float2(q1.z, w1.x);
            e1p[2] = make_float2(w1.y, w1.z);
            if (FORCE) {            // {cp_j, w_j} as the density pass left them (neighbour_terms); padding entries hold 0: weight 0
                e0p[3] = e0;
                e1p[3] = e1;
            }
        },
        [&](int r, uint32_t a, uint32_t b) {
            const uint32_t l0 = max(R.lo[r], a), l1 = min(R.hi[r], b);
            const uint32_t len = l1 > l0 ? l1 - l0 : 0u;
            uint32_t T, tmin_raw;                       // a known trip count lets hipcc interleave two candidates
            wave_min_max_u32(len, tmin_raw, T);
            uint32_t idx = slice + (len ? l0 - a : 0u);
            // Candidates are walked in chunks of 32.  The pressure/viscosity arithmetic runs for every
            // candidate; the collision test only records "d <= 2R" in a per-lane bit mask (2 VALU per
            // candidate) and the few close pairs (~4 of ~216 per particle) are worked off after the chunk.
            // every lane has at least tmin candidates: that part of the walk needs no per-lane range test
            const uint32_t tmin = tmin_raw & ~(uint32_t)(SPH_FORCE_UNROLL - 1);
            uint32_t near = 0u;
            PAIR_STAT(1, 1); PAIR_STAT(2, T);
            auto pair = [&](int u, bool valid) {
                const lds_v2f_ptr e = (lds_v2f_ptr)s_e + (idx + u) * 5;
                const v2f qa = e[0], qb = e[1];
                v2f qc = {0.f, 0.f}, qd = {0.f, 0.f};
                if (FORCE) { qc = e[2]; qd = e[3]; }
                const float t = pair_math(qa.x, qa.y, qb.x, qb.y, qc.x, qc.y, qd.x, qd.y, valid);
                if (COLL) near = __builtin_amdgcn_alignbit(near, __float_as_uint(t), 31);   // (near << 1) | sign(t)
            };
            for (uint32_t t0 = 0; t0 < T; t0 += 32u) {
                const uint32_t tend = min(T, t0 + 32u);
                const uint32_t tsafe = min(tend, max(tmin, t0));      // [t0, tsafe): no range test needed
                const uint32_t idx0 = idx;
                uint32_t done = 0u;
                near = 0u;
                PAIR_STAT(3, 1);
                uint32_t t = t0;
                for (; t < tsafe; t += SPH_FORCE_UNROLL) {
#pragma unroll
                    for (int u = 0; u < SPH_FORCE_UNROLL; u++) pair(u, true);
                    idx += SPH_FORCE_UNROLL;
                    done += SPH_FORCE_UNROLL;
                }
                for (; t < tend; t += SPH_FORCE_UNROLL) {
#pragma unroll
                    for (int u = 0; u < SPH_FORCE_UNROLL; u++) pair(u, t + u < len);
                    idx += SPH_FORCE_UNROLL;
                    done += SPH_FORCE_UNROLL;
                }
#ifdef SPH_PAIR_STATS
                if (COLL) stat_near += (uint32_t)__popc(near);
#endif
                if (COLL) {
                    // bit (done-1-k) of `near` belongs to the k-th candidate of this chunk; highest bit first
                    // keeps the candidate order of the sums
                    while (__ballot(near != 0u) != 0ull) {
                        PAIR_STAT(4, 1);
                        if (near != 0u) {
                            const uint32_t hb = 31u - (uint32_t)__clz((int)near);
                            near &= ~(1u << hb);
                            const uint32_t ci = idx0 + (done - 1u - hb);
                            const float2* e = &s_e[(ci << 2) + ci];          // ci * 5 without v_mul_lo_u32
                            const float2 qa = e[0], qb = e[1], qc = e[2];
                            collide_math(qa.x, qa.y, qb.x, qb.y, qc.x, qc.y);
                        }
                    }
                }
            }
        });
    finish();
}

// The block order of a launch over `nblocks` blocks (BlockOrder, sph_device.hpp).  The z tiling needs the blocks per cell
// layer: estimated from the keys of the first and the last owned slot, which k_cells_build leaves in mapped host memory
// after every sort (a few steps old when the host runs ahead: an estimate is all that is needed).
static BlockOrder block_order(const sph_ctx* c, uint32_t nblocks, bool ztile) {
    BlockOrder o{c->order_xcd ? 1u : 0u, 0u, 0u, 0u, c->order_xrot ? 1u : 0u};
    if (!o.xcd || !ztile || !c->order_ztile || nblocks < 1024u) return o;
    const uint32_t k0 = c->mm_count_host[1], k1 = c->mm_count_host[2], layer = c->grid.g[0] * c->grid.g[1];
    if (k1 < k0 || layer == 0u) return o;
    const uint32_t nz = k1 / layer - k0 / layer + 1u;                    // occupied cell layers
    const uint32_t per_layer = nblocks / nz;                             // blocks per layer, about
    if (per_layer < 32u) return o;
    uint32_t lb_sh = 31u - (uint32_t)__builtin_clz(per_layer);
    if ((per_layer >> lb_sh) && (per_layer - (1u << lb_sh)) > (1u << lb_sh) / 2u) lb_sh++;      // nearest power of two
    const uint32_t q = nblocks >> 3, nl = q >> lb_sh;
    if (nl < 2u) return o;
    o.lb_sh = lb_sh; o.s_sh = c->order_strip_sh; o.nl_sh = 31u - (uint32_t)__builtin_clz(nl);
    if (o.s_sh >= o.lb_sh) o.nl_sh = 0u;
    return o;
}

// [lo, hi) minus the hole [hole_lo, hole_hi): the hole's start is rounded UP to a whole wave from lo (see Targets),
// what is cut off the hole that way is simply computed by this launch as well.
static Targets targets_with_hole(uint32_t lo, uint32_t hi, uint32_t hole_lo, uint32_t hole_hi, uint32_t& threads) {
    Targets t{lo, hi, hi, 0u, nullptr, 0u, BlockOrder{0u, 0u, 0u, 0u, 0u}};
    threads = hi - lo;
    if (hole_lo < lo) hole_lo = lo;
    if (hole_hi > hi) hole_hi = hi;
    if (hole_lo < hole_hi) {
        const uint32_t g0 = lo + ((hole_lo - lo + 63u) & ~63u);
        if (g0 < hole_hi) { t.gap_lo = g0; t.gap_len = hole_hi - g0; threads -= t.gap_len; }
    }
    return t;
}

// One launch of the pair kernel over the owned slots [lo, hi) minus the hole [hole_lo, hole_hi) (lo - own_off,
// hole_lo - lo and the hole's length multiples of 64: a wave is one chunk of the mover marks).  The fused form
// (integrate) writes the ping-pong arrays; force_finish() swaps them once every sub-range has been launched.
int launch_force_hole(sph_ctx* c, uint32_t lo, uint32_t hi, uint32_t hole_lo, uint32_t hole_hi, bool force, bool collide,
                      bool integrate, float dt, bool mark) {
    if (hi <= lo) return SPH_OK;
    SPH_REQUIRE(((lo - c->own_off) & 63u) == 0u, SPH_E_INVALID, "force sub-range does not start on a 64-slot chunk");
    uint32_t threads;
    Targets tg = targets_with_hole(lo, hi, hole_lo, hole_hi, threads);
    tg.direct_hull = c->direct_hull;
    const bool small = small_blocks(c);                                                            // (see SMALL_THREADS_PAIR)
    const uint32_t bt = small ? (uint32_t)SMALL_THREADS_PAIR : (uint32_t)PAIR_THREADS;
    tg.order = block_order(c, ceil_div(threads, bt), integrate && force);      // (the fused launch: see BlockOrder)
    SPH_REQUIRE(tg.gap_len == 0u || (((tg.gap_lo - lo) | tg.gap_len) & 63u) == 0u || tg.gap_lo + tg.gap_len == hi, SPH_E_INVALID,
                "force hole is not made of whole 64-slot chunks");
    if (threads == 0) return SPH_OK;
    dim3 grid(ceil_div(threads, bt)), block(bt);
#define SPH_LAUNCH_FORCE_T(F, C, I, T)                                                                         \
    hipLaunchKernelGGL((k_force<F, C, I, T>), grid, block, 0, c->stream, c->posi, c->velr, c->dp, c->cw, c->keyS, c->cells, \
                       c->fpress, c->fvisc, c->dvel, c->posi2, c->velr2, c->slab ? nullptr : c->pos_out, c->k0,              \
                       mark ? c->mm_mask : nullptr, c->mm_tile_cnt, tg, c->own_off, dt, c->grid, c->phys)
#define SPH_LAUNCH_FORCE(F, C, I) do { if (small) SPH_LAUNCH_FORCE_T(F, C, I, SMALL_THREADS_PAIR); else SPH_LAUNCH_FORCE_T(F, C, I, PAIR_THREADS); } while (0)
    if (force && collide && integrate) SPH_LAUNCH_FORCE(true, true, true);
    else if (force && !collide && !integrate) SPH_LAUNCH_FORCE(true, false, false);
    else if (!force && collide && !integrate) SPH_LAUNCH_FORCE(false, true, false);
    else {
        set_error("launch_force: unsupported combination");
        return SPH_E_INVALID;
    }
#undef SPH_LAUNCH_FORCE
#undef SPH_LAUNCH_FORCE_T
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

// The fused force pass over the slots [range_dev[0], range_dev[1]) -- two words of DEVICE memory written by an earlier kernel
// of the stream -- for the slab step's early launch (csrc/sph_slab.hip): at most max_count slots size the grid, waves beyond
// the range leave at once.  The next step's cell keys go to keyS2 by ABSOLUTE slot (the owned range does not have its
// final start yet) and no movers are marked: k_slab_early_finish does both later.
int launch_force_dev_range(sph_ctx* c, const uint32_t* range_dev, uint32_t max_count, float dt) {
    if (max_count == 0) return SPH_OK;
    const Targets tg{0u, 0u, 0xFFFFFFFFu, 0u, range_dev, c->direct_hull, BlockOrder{0u, 0u, 0u, 0u, 0u}};
#define SPH_LAUNCH_EARLY(T)                                                                                                         \
    hipLaunchKernelGGL((k_force<true, true, true, T>), dim3(ceil_div(max_count, (uint32_t)T)), dim3(T), 0, c->stream, c->posi, c->velr,   \
                       c->dp, c->cw, c->keyS, c->cells, c->fpress, c->fvisc, c->dvel, c->posi2, c->velr2, c->slab ? nullptr : c->pos_out, \
                       c->keyS2, (uint64_t*)nullptr, c->mm_tile_cnt, tg, 0u, dt, c->grid, c->phys)
    if (small_blocks(c)) SPH_LAUNCH_EARLY(SMALL_THREADS_PAIR); else SPH_LAUNCH_EARLY(PAIR_THREADS);
#undef SPH_LAUNCH_EARLY
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

int launch_force_range(sph_ctx* c, uint32_t lo, uint32_t hi, bool force, bool collide, bool integrate, float dt, bool mark) {
    return launch_force_hole(c, lo, hi, hi, hi, force, collide, integrate, dt, mark);
}

// the integrate epilogue marks the movers of the next sort; marks of an earlier launch that no sort
// consumed are dropped first (they would be counted twice)
bool force_begin(sph_ctx* c, bool integrate) {
    if (integrate) mm_drop_marks(c);
    return integrate && c->sort_merge && c->order_valid;
}

void force_finish(sph_ctx* c, bool integrate, bool mark) {
    if (!integrate) return;
    float4* t;
    t = c->posi; c->posi = c->posi2; c->posi2 = t;
    t = c->velr; c->velr = c->velr2; c->velr2 = t;
    c->keys_fresh = true;
    if (mark) {
        c->mm_marked = true; c->mm_scanned = false; c->mm_marked_off = c->own_off; c->mm_marked_n = c->n;
        mm_scan_marks(c);          // count them now: the next sort finds the number ready
    }
}

int launch_force(sph_ctx* c, bool force, bool collide, bool integrate, float dt) {
    if (c->n == 0) return SPH_OK;
    const bool mark = force_begin(c, integrate);
    int rc = launch_force_range(c, c->own_off, c->own_off + c->n, force, collide, integrate, dt, mark);
    if (rc) return rc;
    force_finish(c, integrate, mark);
    return SPH_OK;
}

static int launch_density_targets(sph_ctx* c, const Targets& tg, uint32_t threads) {
    if (threads == 0) return SPH_OK;
    if (c->precision == SPH_PRECISION_MIXED_F16)
        hipLaunchKernelGGL(k_density_h, dim3(ceil_div(threads, PAIR_THREADS)), dim3(PAIR_THREADS), 0, c->stream, c->posi,
                           c->keyS, c->cells, c->dp, c->cw, tg, c->grid, c->phys);
    else if (small_blocks(c))
        hipLaunchKernelGGL(k_density<SMALL_THREADS_PAIR>, dim3(ceil_div(threads, (uint32_t)SMALL_THREADS_PAIR)), dim3(SMALL_THREADS_PAIR), 0, c->stream,
                           c->posi, c->keyS, c->cells, c->dp, c->cw, tg, c->grid, c->phys);
    else
        hipLaunchKernelGGL(k_density<DENS_THREADS>, dim3(ceil_div(threads, (uint32_t)DENS_THREADS)), dim3(DENS_THREADS), 0, c->stream, c->posi,
                           c->keyS, c->cells, c->dp, c->cw, tg, c->grid, c->phys);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

// density over the owned slots [lo, hi) minus the hole [hole_lo, hole_hi)
int launch_density_hole(sph_ctx* c, uint32_t lo, uint32_t hi, uint32_t hole_lo, uint32_t hole_hi) {
    if (hi <= lo) return SPH_OK;
    uint32_t threads;
    Targets tg = targets_with_hole(lo, hi, hole_lo, hole_hi, threads);
    tg.direct_hull = c->direct_hull;
    const uint32_t bt = c->precision == SPH_PRECISION_MIXED_F16 ? (uint32_t)PAIR_THREADS : (small_blocks(c) ? (uint32_t)SMALL_THREADS_PAIR : (uint32_t)DENS_THREADS);
    tg.order = block_order(c, ceil_div(threads, bt), c->order_ztile_dens);
    return launch_density_targets(c, tg, threads);
}

int launch_density_range(sph_ctx* c, uint32_t lo, uint32_t hi) { return launch_density_hole(c, lo, hi, hi, hi); }

// density over the slots [range_dev[0], range_dev[1]) -- two words of DEVICE memory written by an earlier kernel of
// the stream; at most max_count slots (sizes the grid; waves beyond the range leave at once)
int launch_density_dev_range(sph_ctx* c, const uint32_t* range_dev, uint32_t max_count) {
    // plain block order: the grid is only an upper bound, and with a contiguous eighth per XCD the blocks beyond the range
    // would all belong to the last XCDs -- the first ones would do all the work
    const Targets tg{0u, 0u, 0xFFFFFFFFu, 0u, range_dev, c->direct_hull, BlockOrder{0u, 0u, 0u, 0u, 0u}};
    return launch_density_targets(c, tg, max_count);
}

// ---- stand-alone integrate for the phase API --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_integrate(float4* __restrict__ posi, float4* __restrict__ velr,
                                                   const float2* __restrict__ dp, const float4* __restrict__ fpress,
                                                   const float4* __restrict__ fvisc, const float4* __restrict__ dvel,
                                                   float4* __restrict__ pos_by_index, uint32_t lo, uint32_t hi, float dt,
                                                   Phys ph) {
    uint32_t i = lo + blockIdx.x * 256u + threadIdx.x;
    if (i >= hi) return;
    float4 pi = posi[i], vi = velr[i];
    const float4 fp = fpress[i], fv = fvisc[i], dv = dvel[i];
    integrate_one(ph, dt, pi, vi, dp[i].x, fp.x + fv.x, fp.y + fv.y, fp.z + fv.z, dv.x, dv.y, dv.z);
    posi[i] = pi;
    velr[i] = vi;
    if (pos_by_index) pos_by_index[__float_as_uint(pi.w)] = make_float4(pi.x, pi.y, pi.z, 1.0f);
}

int launch_integrate(sph_ctx* c, float dt) {
    if (c->n == 0) return SPH_OK;
    hipLaunchKernelGGL(k_integrate, dim3(ceil_div(c->n, 256)), dim3(256), 0, c->stream, c->posi, c->velr, c->dp,
                       c->fpress, c->fvisc, c->dvel, c->slab ? nullptr : c->pos_out, c->own_off, c->own_off + c->n, dt, c->phys);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

}  // namespace sph
