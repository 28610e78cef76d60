// sph_sort.hip -- cell hash, on-device LSD radix sort of (cell key, slot) pairs, reorder.
//
// Replaces cudaMapZIndex + cudaSortParticles (particleSystem.cu:491-501): the reference
// thrust::sort()s the whole 88-byte AoS array with a comparator (a merge sort moving 88 B per
// element per level, 90 % of its step at 131k particles).  Here only 8-byte (key, slot) pairs are
// sorted, 8 bits per pass over the significant key bits, and the SoA payload is gathered once.
// The sort is stable, so the order of particles inside a cell is deterministic.
#include "sph_device.hpp"

#include <atomic>
#include <chrono>

namespace sph {

constexpr int SORT_THREADS = 256;            // 4 waves
#ifndef SPH_SORT_KPT
#define SPH_SORT_KPT 16
#endif
// The ranked pairs of a tile are parked in LDS in digit order and written out with neighbouring threads on neighbouring
// addresses.  (Tried and dropped: every lane storing its pair straight from registers -- uncoalesced, 8 KB of LDS per block
// instead of 40; keys and values taking turns in ONE 16 KB buffer, 26 KB of LDS per block -- the registers that hold the
// tile positions meanwhile cost more waves than the LDS frees: 137-177 us per pass against 106.)
constexpr int SORT_KPT = SPH_SORT_KPT;       // keys per thread
constexpr int SORT_TILE = SORT_THREADS * SORT_KPT;       // 4096 keys per block

// ---- hash: key per owned particle ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash(const float4* __restrict__ posi, uint32_t n, GridDesc g,
                                              uint32_t* __restrict__ keys) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = posi[i];
    keys[i] = cell_key(g, p.x, p.y, p.z);
}

// ---- onesweep LSD radix sort of (key, value) pairs ----------------------------------------------------
// A histogram kernel counts the digits per GROUP of tiles, a small kernel turns the counts into the first output
// position of every (group, digit); then ONE kernel per pass:
// a block takes a tile of 4096 consecutive keys (by ticket, so that tiles are started in order), counts its
// digits, publishes the counts, obtains the number of equal digits in the EARLIER tiles OF ITS GROUP by decoupled
// look-back over the predecessors' published words, ranks its keys (stable: waves, rows and lanes in key order), parks the
// pairs in LDS in digit order and writes them out -- neighbouring threads store neighbouring addresses of one
// digit's run.  Per pass 16 B per key of HBM traffic (the three-kernel hist / scan / scatter it replaces: 20 B and
// an uncoalesced scatter) and P + 1 launches per sort instead of 3 P.
//
// Inter-block words (MI355X: 8 XCDs, private L2s, no coherent L1): one 8-byte word per (tile, digit),
// {epoch << 1 | is_prefix, count}, stored and polled with relaxed agent-scope atomics (sc1: write-through /
// L1-bypassing) -- the value IS the flag, so no fence is needed (cdna_hip_programming.md, Guideline 16, R2).  The
// epoch changes with every pass of every sort, so the table is never cleared.  A block only ever waits for blocks
// with a smaller ticket, and a ticket is taken by a block that is already running: the wait always ends; it is
// bounded all the same (sph_sync reports a timeout).
//
// All kernels take the element count by value or from device memory (n_dev, clamped to n): the movers' sort of
// the merge path is sized from a stale estimate and is correct for any count.
constexpr int OS_TILE = SORT_TILE;                       // 4096 keys per tile
// Tiles per look-back group.  One chain over all 4096 tiles of C3 made a pass take 240 us (~58 ns per tile), so a
// big sort uses groups of 16 tiles whose chains run in parallel -- at the price of one histogram kernel PER PASS (the
// digit counts of a group are those of the keys as that pass finds them).  Up to OS_ONE_GROUP_TILES tiles (the movers'
// sort of the merge path) there is ONE group: its counts do not depend on the order of the keys, so a single
// histogram kernel up front serves every pass (P + 2 launches per sort).
#ifndef SPH_OS_GROUP
#define SPH_OS_GROUP 16
#endif
constexpr uint32_t OS_GROUP = SPH_OS_GROUP;
constexpr uint32_t OS_ONE_GROUP_TILES = 64;
constexpr uint32_t OS_ALL_TILES = 0xFFFFFFFFu;           // group_tiles value for "one group"
constexpr uint32_t OS_SPIN_LIMIT = 1u << 22;
// A big pass runs with twice the blocks that fit the chip at once (3 per CU: LDS); every block takes tiles until none is
// left, its next ticket always drawn while it works on the tile in hand.
#ifndef SPH_OS_PASS_GRID_MAX
#define SPH_OS_PASS_GRID_MAX 1536
#endif
constexpr uint32_t OS_PASS_GRID_MAX = SPH_OS_PASS_GRID_MAX;
// Block of a pass kernel.  A tile is 4096 keys whatever the block: with 512 threads a thread holds 8 pairs instead of
// 16 and a CU keeps twice the waves in flight at the same LDS per tile -- measured 119 us per pass against 107 with
// 256 threads (1024: 153): what a tile waits for is its own chain (ticket, loads, the predecessors' counts:
// profiles/r02_os_pass_stages.txt), not a lack of waves.
#ifndef SPH_OS_PASS_THREADS
#define SPH_OS_PASS_THREADS 256
#endif
constexpr int PASS_THREADS = SPH_OS_PASS_THREADS;
constexpr int PASS_WAVES = PASS_THREADS / WAVE;
constexpr int PASS_KPT = OS_TILE / PASS_THREADS;         // pairs per thread
constexpr int PASS_WAVE_TILE = WAVE * PASS_KPT;          // consecutive keys per wave

__device__ __forceinline__ uint32_t sort_count(uint32_t n, const uint32_t* __restrict__ n_dev) {
    return n_dev ? min(*n_dev, n) : n;
}

// exclusive scan of one value per thread over a block of NW waves (wave shuffles + NW wave totals in LDS)
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wtot /*[NW]*/, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= (uint32_t)off) inc += t;
    }
    if (lane == 63u) s_wtot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < (uint32_t)NW; w++) { const uint32_t t = s_wtot[w]; if (w < wave) base += t; tot += t; }
    if (total) *total = tot;
    __syncthreads();                                     // s_wtot may be reused by the caller
    return base + inc - v;
}

// count[d] += 1 for every valid lane.  Keys arrive nearly sorted (particle order of the previous step, or lattice
// order), so the high digits are the same in all 64 lanes more often than not: then one lane adds the popcount
// instead of 64 lanes queueing up on one LDS word.
__device__ __forceinline__ void wave_count_digit(uint32_t* counts, uint32_t d, bool valid) {
    const uint64_t act = __ballot(valid);
    if (act == 0ull) return;                                              // wave-uniform
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, __builtin_ctzll(act));
    if (__ballot(valid && d != d0) == 0ull) {                             // wave-uniform
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(act)) atomicAdd(&counts[d0], (uint32_t)__popcll(act));
    } else if (valid) {
        atomicAdd(&counts[d], 1u);
    }
}

// digit counts of the passes pass0 .. pass0+passes-1 per group of tiles: hist[(group * 4 + pass) * 512 + digit],
// ACCUMULATED (k_os_scan zeroes what it has read).  A block counts OS_HIST_TILES consecutive tiles (all of one
// group) in LDS and adds what it found; 16-byte loads.
constexpr uint32_t OS_HIST_TILES = 2;
template <int BITS>
__global__ __launch_bounds__(SORT_THREADS) void k_os_hist(const uint32_t* __restrict__ keys, uint32_t n_arg,
                                                          const uint32_t* __restrict__ n_dev, uint32_t pass0,
                                                          uint32_t passes, uint32_t group_tiles,
                                                          uint32_t* __restrict__ hist, uint32_t small_max) {
    constexpr int RADIX = 1 << BITS;
    __shared__ uint32_t h[4 * RADIX];
    const uint32_t n = sort_count(n_arg, n_dev);
    if (n <= small_max) return;                              // k_os_small sorts it (small_max = 0: never)
    const uint32_t ntiles = (n + OS_TILE - 1) / OS_TILE;
    const uint32_t nchunks = (ntiles + OS_HIST_TILES - 1) / OS_HIST_TILES;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        for (uint32_t d = threadIdx.x; d < passes * RADIX; d += SORT_THREADS) h[d] = 0;
        __syncthreads();
        const uint32_t t0 = chunk * OS_HIST_TILES, t1 = min(t0 + OS_HIST_TILES, ntiles);
        for (uint32_t tile = t0; tile < t1; tile++) {
#pragma unroll
            for (int q = 0; q < SORT_KPT / 4; q++) {
                const uint32_t i = tile * OS_TILE + (q * SORT_THREADS + threadIdx.x) * 4u;
                uint4 k4 = make_uint4(0u, 0u, 0u, 0u);
                if (i + 3u < n) k4 = *reinterpret_cast<const uint4*>(keys + i);
                else {
                    if (i < n) k4.x = keys[i];
                    if (i + 1u < n) k4.y = keys[i + 1u];
                    if (i + 2u < n) k4.z = keys[i + 2u];
                }
                const uint32_t kk[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                    for (uint32_t p = 0; p < passes; p++)
                        wave_count_digit(h + p * RADIX, (kk[e] >> ((pass0 + p) * BITS)) & (RADIX - 1), i + e < n);
            }
        }
        __syncthreads();
        const uint32_t grp = group_tiles == OS_ALL_TILES ? 0u : t0 / group_tiles;
        for (uint32_t d = threadIdx.x; d < passes * RADIX; d += SORT_THREADS)
            if (h[d]) atomicAdd(&hist[(grp * 4u + pass0 + d / RADIX) * 512u + (d % RADIX)], h[d]);
        __syncthreads();
    }
}

// One wave per digit (blockIdx.y = pass - pass0): base[(group * 4 + pass) * 512 + digit] = keys with this digit in
// EARLIER groups, tot[pass * 512 + digit] = keys with this digit; the counts are zeroed for the next sort and the
// pass's ticket is re-armed.
template <int BITS>
__global__ __launch_bounds__(256) void k_os_scan(uint32_t* __restrict__ hist, uint32_t* __restrict__ base,
                                                 uint32_t* __restrict__ tot, uint32_t n_arg,
                                                 const uint32_t* __restrict__ n_dev, uint32_t* __restrict__ tickets,
                                                 uint32_t tickets_stride, uint32_t pass0, uint32_t group_tiles,
                                                 uint32_t small_max) {
    constexpr uint32_t RADIX = 1u << BITS;
    const uint32_t n = sort_count(n_arg, n_dev);
    if (n <= small_max) return;                              // nothing was counted, no ticket was drawn
    const uint32_t ntiles = (n + OS_TILE - 1) / OS_TILE;
    const uint32_t ngroups = group_tiles == OS_ALL_TILES ? 1u : (ntiles + group_tiles - 1) / group_tiles;
    const uint32_t p = pass0 + blockIdx.y;
    const uint32_t d = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (d >= RADIX) return;                                  // wave-uniform
    if (d == 0 && lane == 0) tickets[(size_t)p * tickets_stride] = 0u;      // one-group form: one ticket per pass
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < ngroups; g0 += 64u) {
        const uint32_t g = g0 + lane;
        const uint32_t idx = (g * 4u + p) * 512u + d;
        uint32_t v = 0;
        if (g < ngroups) { v = hist[idx]; hist[idx] = 0u; }
        if (d == 0 && g < ngroups) tickets[(size_t)p * tickets_stride + g] = 0u;   // grouped form: one ticket per group
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= (uint32_t)off) inc += t;
        }
        if (g < ngroups) base[idx] = carry + inc - v;
        carry += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) tot[p * 512u + d] = carry;
}

typedef unsigned long long os_word;
typedef volatile __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
// one wave's LDS operations are processed in issue order; only the compiler must not move them
__device__ __forceinline__ void os_wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void os_publish(os_word* w, uint32_t epoch, bool prefix, uint32_t value) {
    __hip_atomic_store(w, ((os_word)((epoch << 1) | (prefix ? 1u : 0u)) << 32) | value, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// GROUPED (big sorts, groups of OS_GROUP tiles): a tile publishes ONE word per digit, {epoch:19, count:13} (two digits
// per 8-byte store), and sums the counts of the <= 15 earlier tiles of its group -- independent loads, all in flight
// at once, no chain at all; tickets are per group (no hot word).  !GROUPED (one group, the movers' sort: few tiles,
// but correct for any number): the classic {count} / {inclusive prefix} words with a chained look-back.
constexpr uint32_t OS_CNT_BITS = 13;                     // a tile holds at most 4096 = 2^12 keys of one digit
__device__ __forceinline__ uint32_t os_pack32(uint32_t epoch, uint32_t count) { return (epoch << OS_CNT_BITS) | count; }

#ifdef SPH_OS_STATS
__device__ unsigned long long g_os_stats[10];   // debug build only: ticks per stage of k_os_pass, summed over tiles (thread 0)
#define OS_STAT(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
                        atomicAdd(&g_os_stats[k], now_ - t_prev_); t_prev_ = now_; } } while (0)
extern "C" void sph_debug_os_stats(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_os_stats), sizeof(g_os_stats));
    if (reset) { unsigned long long z[10] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_os_stats), z, sizeof(z)); }
}
#else
#define OS_STAT(k) do { } while (0)
#endif
#ifndef SPH_OS_PASS_OCC
#define SPH_OS_PASS_OCC 3                        // waves per SIMD asked of the register allocator
#endif
template <int BITS, bool FIRST, bool GROUPED>
__global__ __launch_bounds__(PASS_THREADS, SPH_OS_PASS_OCC) void k_os_pass(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                          uint32_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                          uint32_t n_arg, const uint32_t* __restrict__ n_dev,
                                                          uint32_t shift, const uint32_t* __restrict__ group_base,
                                                          const uint32_t* __restrict__ digit_tot,
                                                          uint32_t group_tiles, os_word* __restrict__ status,
                                                          uint32_t* __restrict__ status32,
                                                          uint32_t* __restrict__ ticket, uint32_t epoch,
                                                          uint32_t* __restrict__ err, uint32_t small_max) {
    constexpr int RADIX = 1 << BITS;
    constexpr int DPT = RADIX > PASS_THREADS ? RADIX / PASS_THREADS : 1;   // digits per thread: 1 or 2 (consecutive digits)
    if (sort_count(n_arg, n_dev) <= small_max) return;      // k_os_small sorts it (block-uniform, before any barrier)
    static_assert(DPT <= 2, "a thread publishes at most two digits in one store");
    __shared__ uint32_t wh[PASS_WAVES][RADIX];      // per-wave digit counts -> running positions inside the tile
    __shared__ uint32_t s_delta[RADIX];             // global position minus LDS position of a digit's keys of this tile
    __shared__ uint32_t s_key[OS_TILE], s_val[OS_TILE];
    __shared__ uint32_t s_wtot[PASS_WAVES];
    __shared__ uint32_t s_tile;
    const uint32_t n = sort_count(n_arg, n_dev);
    const uint32_t ntiles = (n + OS_TILE - 1) / OS_TILE;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const bool has_d = threadIdx.x * DPT < RADIX;   // more threads than digits: the rest only move pairs
    // first output position of every digit: exclusive scan of the pass's totals (thread t owns digits t*DPT ..)
    uint32_t dbase[DPT];
    {
        uint32_t tot[DPT], sum = 0;
#pragma unroll
        for (int k = 0; k < DPT; k++) { tot[k] = has_d ? digit_tot[threadIdx.x * DPT + k] : 0u; sum += tot[k]; }
        uint32_t run = block_excl_scan<PASS_WAVES>(sum, s_wtot, nullptr);
#pragma unroll
        for (int k = 0; k < DPT; k++) { dbase[k] = run; run += tot[k]; }
    }
#ifdef SPH_OS_STATS
    unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif
    // Tickets are drawn one tile AHEAD (thread 0): the atomic's trip to memory and back hides behind the tile in
    // hand.  The blocks b, b + 8, b + 16 .. share an XCD (blocks are dealt round-robin over the 8 XCDs; speed only).
    // GROUPED: a group's tiles are taken by OS_GROUP blocks of ONE XCD -- neighbouring tiles write neighbouring
    // pieces of every digit's run (a piece is 4096 / RADIX keys: 32 B), and one L2 can put them together before they
    // go to memory; the grid is a multiple of 8; a block takes the next tile of ITS group.  No deadlock: a tile only
    // waits for tiles of its group with a smaller ticket; a block's next tile has a larger ticket than (or a later
    // group than) the tile it has in hand, so the smallest unfinished tile of the earliest unfinished group is
    // always in some running block's hands, not in its pocket.
    uint32_t round = 0, next_tile = 0;
    auto draw = [&]() -> uint32_t {
        if (GROUPED) {
            const uint32_t idx = (blockIdx.x >> 3) + round * (gridDim.x >> 3);
            round++;
            const uint32_t g = (idx / OS_GROUP) * 8u + (blockIdx.x & 7u);
            return g * OS_GROUP < ntiles ? g * OS_GROUP + atomicAdd(ticket + g, 1u) : 0xFFFFFFFFu;
        }
        return atomicAdd(ticket, 1u);
    };
    if (threadIdx.x == 0) next_tile = draw();
    for (;;) {
        OS_STAT(0);                                 // between tiles (first: the prologue)
        if (threadIdx.x == 0) s_tile = next_tile;
        for (int d = threadIdx.x; d < PASS_WAVES * RADIX; d += PASS_THREADS) (&wh[0][0])[d] = 0;
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= ntiles) return;                 // block-uniform
        if (threadIdx.x == 0) next_tile = draw();
        OS_STAT(1);                                 // ticket

        // this wave's 1024 consecutive keys (registers), counted per wave
        const uint32_t wbase = tile * OS_TILE + wave * PASS_WAVE_TILE;
        uint32_t key[PASS_KPT], val[PASS_KPT];
        // every load is issued before the first key is looked at (one loop would wait for each pair in turn: the LDS
        // atomics of the count keep the compiler from moving the later loads up)
#pragma unroll
        for (int t = 0; t < PASS_KPT; t++) {
            const uint32_t i = wbase + t * WAVE + lane;
            key[t] = i < n ? kin[i] : 0xFFFFFFFFu;
            val[t] = FIRST ? i : (i < n ? vin[i] : 0u);
        }
#pragma unroll
        for (int t = 0; t < PASS_KPT; t++)
            wave_count_digit(wh[wave], (key[t] >> shift) & (RADIX - 1), wbase + t * WAVE + lane < n);
        __syncthreads();
        OS_STAT(2);                                 // load + count

        // per digit: count of the tile and exclusive offsets of the four waves; publish the count at once
        const uint32_t grp = group_tiles == OS_ALL_TILES ? 0u : tile / group_tiles;
        const uint32_t gstart = group_tiles == OS_ALL_TILES ? 0u : grp * group_tiles;
        uint32_t cnt[DPT], tstart[DPT];
#pragma unroll
        for (int k = 0; k < DPT; k++) {
            const uint32_t d = threadIdx.x * DPT + k;
            uint32_t o = 0;
            if (has_d) {
#pragma unroll
                for (int w = 0; w < PASS_WAVES; w++) { const uint32_t c = wh[w][d]; wh[w][d] = o; o += c; }
            }
            cnt[k] = o;
            if (!GROUPED && has_d) os_publish(status + (size_t)tile * RADIX + d, epoch, tile == gstart, o);
        }
        if (GROUPED && has_d) {
            uint32_t* st32 = status32 + (size_t)tile * RADIX + threadIdx.x * DPT;
            if (DPT == 2)
                __hip_atomic_store(reinterpret_cast<os_word*>(st32),
                                   ((os_word)os_pack32(epoch, cnt[DPT - 1]) << 32) | os_pack32(epoch, cnt[0]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_store(st32, os_pack32(epoch, cnt[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // positions inside the tile: digits in ascending order
        {
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < DPT; k++) sum += cnt[k];
            uint32_t run = block_excl_scan<PASS_WAVES>(sum, s_wtot, nullptr);
#pragma unroll
            for (int k = 0; k < DPT; k++) {
                const uint32_t d = threadIdx.x * DPT + k;
                tstart[k] = run;
                if (has_d) {
#pragma unroll
                    for (int w = 0; w < PASS_WAVES; w++) wh[w][d] += run;      // wave-exclusive offset + start of the digit
                }
                run += cnt[k];
            }
        }
        __syncthreads();
        OS_STAT(3);                                 // publish + scan

        // GROUPED: the counts of the earlier tiles of the group are requested HALFWAY through the ranking, every load in
        // flight at once, and looked at after it: their trip to memory and back (the words are written through, the
        // loads pass the L2) hides behind the second half, and by then the tiles that started together with this one
        // have published (asked for right after the own publication, most words came back stale and cost a second
        // trip).  A word that is still not there is asked for again below.
        const uint32_t npred = tile - gstart;
        const uint32_t* st32 = status32 + (size_t)gstart * RADIX + threadIdx.x * DPT;
        os_word w[OS_GROUP - 1];

        // rank: rows of 64 keys in order; equal digits of a row by ballot match-any + popcount of the lower lanes
        // an LDS-typed pointer: a generic `volatile uint32_t*` made every access a flat_load/flat_store with sc0 sc1
        // and a full s_waitcnt behind it
        const lds_u32_ptr pos = (lds_u32_ptr)wh[wave];
#pragma unroll
        for (int t = 0; t < PASS_KPT; t++) {
            if (GROUPED && t == PASS_KPT / 2 && has_d) {
#pragma unroll
                for (uint32_t q = 0; q < OS_GROUP - 1; q++)
                    if (q < npred) {
                        if (DPT == 2) w[q] = __hip_atomic_load(reinterpret_cast<const os_word*>(st32 + (size_t)q * RADIX),
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else w[q] = __hip_atomic_load(st32 + (size_t)q * RADIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            }
            const uint32_t i = wbase + t * WAVE + lane;
            const bool valid = i < n;
            const uint32_t d = (key[t] >> shift) & (RADIX - 1);
            // the lanes whose digit differs from mine, bit by bit, in 32-bit halves: bm = my bit b spread over a word
            // (one v_bfe_i32), m = the wave's ballot of bit b, and `dif |= m ^ bm` is one three-input bit-op per half
            // (written with 64-bit per-lane selects the same loop cost twice the instructions, and the pass is bound
            // by the instructions it issues)
            uint32_t dif_lo = 0u, dif_hi = 0u;
#pragma unroll
            for (int b = 0; b < BITS; b++) {
                const uint32_t bm = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);
                const uint64_t m = __ballot(bm != 0u);
                dif_lo |= (uint32_t)m ^ bm;
                dif_hi |= (uint32_t)(m >> 32) ^ bm;
            }
            const uint64_t vmask = __ballot(valid);
            const uint32_t peers_lo = ~dif_lo & (uint32_t)vmask, peers_hi = ~dif_hi & (uint32_t)(vmask >> 32);
            const uint32_t rank = (uint32_t)__popc(peers_lo & (uint32_t)lt_mask) + (uint32_t)__popc(peers_hi & (uint32_t)(lt_mask >> 32));
            uint32_t base = 0;
            if (valid) base = pos[d];                       // every peer reads the same word
            os_wave_lds_order();
            if (valid && rank == 0) pos[d] = base + (uint32_t)__popc(peers_lo) + (uint32_t)__popc(peers_hi);   // LDS is in order per wave
            os_wave_lds_order();
            if (valid) { s_key[base + rank] = key[t]; s_val[base + rank] = val[t]; }
        }

        OS_STAT(4);                                 // rank (thread 0's wave)
        // keys with the same digit in the EARLIER tiles of the group: decoupled look-back, AFTER the ranking -- by
        // now the predecessors have long published, so the walk rarely meets an unfinished tile
        {
            uint32_t excl[DPT];
#pragma unroll
            for (int k = 0; k < DPT; k++) excl[k] = 0;
            if (GROUPED && has_d) {
                // the counts of the earlier tiles of the group (requested before the ranking, see above).  Words that
                // were not there yet are asked for again ALL AT ONCE, round after round: one trip to memory per round,
                // not one per stale word
                auto fresh = [&](os_word v) {
                    return (uint32_t)v >> OS_CNT_BITS == epoch && (DPT == 1 || (uint32_t)(v >> 32) >> OS_CNT_BITS == epoch);
                };
                uint32_t pending = 0u;
#pragma unroll
                for (uint32_t q = 0; q < OS_GROUP - 1; q++)
                    if (q < npred && !fresh(w[q])) pending |= 1u << q;
                uint32_t spins = 0;
                while (pending) {
                    if (++spins > OS_SPIN_LIMIT) {                            // give up: counts of 0
                        *err = 1u;
#pragma unroll
                        for (uint32_t q = 0; q < OS_GROUP - 1; q++) if (pending >> q & 1u) w[q] = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (uint32_t q = 0; q < OS_GROUP - 1; q++)
                        if (pending >> q & 1u) {
                            if (DPT == 2) w[q] = __hip_atomic_load(reinterpret_cast<const os_word*>(st32 + (size_t)q * RADIX),
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else w[q] = __hip_atomic_load(st32 + (size_t)q * RADIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                    for (uint32_t q = 0; q < OS_GROUP - 1; q++)
                        if ((pending >> q & 1u) && fresh(w[q])) pending &= ~(1u << q);
                }
#pragma unroll
                for (uint32_t q = 0; q < OS_GROUP - 1; q++)
                    if (q < npred) {
                        excl[0] += (uint32_t)w[q] & ((1u << OS_CNT_BITS) - 1u);
                        if (DPT == 2) excl[DPT - 1] += (uint32_t)(w[q] >> 32) & ((1u << OS_CNT_BITS) - 1u);
                    }
            } else if (!GROUPED && has_d && tile != gstart) {
                bool open[DPT];
#pragma unroll
                for (int k = 0; k < DPT; k++) open[k] = true;
                uint32_t spins = 0;
                for (uint32_t j = tile; j-- > gstart;) {           // the digits of a thread walk back together
                    bool any_open = false;
#pragma unroll
                    for (int k = 0; k < DPT; k++) {
                        if (!open[k]) continue;
                        const os_word* src = status + (size_t)j * RADIX + threadIdx.x * DPT + k;
                        os_word w;
                        for (;;) {
                            w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)(w >> 33) == epoch) break;
                            if (++spins > OS_SPIN_LIMIT) { *err = 1u; w = (os_word)1 << 32; break; }   // give up: prefix 0
                            __builtin_amdgcn_s_sleep(1);
                        }
                        excl[k] += (uint32_t)w;
                        if ((w >> 32) & 1ull) open[k] = false; else any_open = true;
                    }
                    if (!any_open) break;
                }
#pragma unroll
                for (int k = 0; k < DPT; k++)
                    os_publish(status + (size_t)tile * RADIX + threadIdx.x * DPT + k, epoch, true, excl[k] + cnt[k]);
            }
#pragma unroll
            for (int k = 0; k < DPT; k++) {
                const uint32_t d = threadIdx.x * DPT + k;
                if (has_d) s_delta[d] = dbase[k] + group_base[(grp * 4u) * 512u + d] + excl[k] - tstart[k];
            }
        }
        OS_STAT(5);                                 // look-back (thread 0)
        __syncthreads();
        OS_STAT(6);                                 // waiting for the slowest wave

        // write out in LDS (= digit) order: a digit's keys go to consecutive addresses
        const uint32_t tile_n = min((uint32_t)OS_TILE, n - tile * OS_TILE);
#pragma unroll 4
        for (int t = 0; t < PASS_KPT; t++) {
            const uint32_t q = t * PASS_THREADS + threadIdx.x;
            if (q < tile_n) {
                const uint32_t k = s_key[q];
                const uint32_t d = (k >> shift) & (RADIX - 1);
                const uint32_t dst = s_delta[d] + q;
                kout[dst] = k;
                vout[dst] = s_val[q];
            }
        }
        OS_STAT(7);                                 // write-out issued
        __syncthreads();                                    // LDS is re-used by the next tile
        OS_STAT(8);
    }
}

// ---- the whole sort of up to one tile (4096 pairs) in ONE block ----------------------------------------------------------
// The movers of a step are usually a few thousand (flowing C3 on 8 GPUs: ~5 000 per rank and step).  Sorting them with
// the kernels above is P + 2 launches that each handle a single tile: ~10 us apiece of launch, prologue and latency
// chain, 56 us per step -- 8 % of a 0.7 ms step.  Here one block of 1024 threads keeps up to 8192 pairs in LDS and runs every
// pass itself (the same stable ranking: rows of 64 keys in order, equal digits of a row by ballot match-any), then
// -- the sorted movers still in LDS -- also ranks the coarse tile boundaries of the merge (k_mm_tile_rank's job).
// The count lives on the device, so the generic kernels are still launched behind it; they, and this one, look at the
// count first and leave at once when it is not theirs (count <= OS_SMALL_MAX: this kernel; else: the others).
constexpr int SMALL_THREADS = 1024;
constexpr int SMALL_WAVES = SMALL_THREADS / WAVE;                 // 16
#ifndef SPH_OS_SMALL_TILES
#define SPH_OS_SMALL_TILES 2                                      // pairs the one-block sort takes, in tiles of 4096
#endif
constexpr uint32_t OS_SMALL_MAX = SPH_OS_SMALL_TILES * OS_TILE;   // 8192 pairs: 64 KB of LDS for the pairs + 32 KB of counters
constexpr int SMALL_KPT = OS_SMALL_MAX / SMALL_THREADS;           // rows of 64 keys per wave
// Movers that go IN FRONT of the non-movers of their cell whatever their slot: the slots [lo, lo + cnt).  Only the merge
// of a slab's arrivals uses it (launch_merge_arrivals: the particles that came up from the slab below are appended
// behind the owned range, but in the whole-domain order -- a stable sort by (new key, old slot) -- they precede every
// resident of their new cell); cnt = 0 everywhere else.
struct Front {
    uint32_t lo, cnt;
};
// (key, slot) order of a MOVER against a non-mover's slot `slot` of the same key
__device__ __forceinline__ bool mover_slot_less(uint32_t mover_slot, uint32_t slot, Front f) {
    return mover_slot < slot || mover_slot - f.lo < f.cnt;         // (unsigned: slots below f.lo wrap to huge values)
}

// first r in [0, m) with (sk[r], sv[r]) >= (key, slot), the arrays in LDS
__device__ __forceinline__ uint32_t small_lower_bound(const uint32_t* sk, const uint32_t* sv, uint32_t m, uint32_t key,
                                                      uint32_t slot, Front f) {
    uint32_t lo = 0, hi = m;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t k = sk[mid];
        if (k < key || (k == key && mover_slot_less(sv[mid], slot, f))) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t mm_lower_bound(const uint32_t* __restrict__ mk, const uint32_t* __restrict__ mi,
                                                   uint32_t lo, uint32_t hi, uint32_t key, uint32_t slot, Front f);

// `alone`: no other sort kernel is launched beside this one (radix_sort_bits, when the host's last known count says the
// movers fit): then a count beyond OS_SMALL_MAX -- the first step of a burst, one step ahead of the host's knowledge --
// is sorted here all the same, by this ONE block, tile by tile through global memory (correct for any count, slow for
// large ones: ~10 us per tile and pass; the next sort sees the count and takes the multi-block form).  The pairs
// alternate between the buffers (ak, av) -- the input -- and (bk, bv); the result ends in (ak, av) after an even number
// of passes, else in (bk, bv): where the multi-block passes would have left it.
template <int BITS>
__global__ __launch_bounds__(SMALL_THREADS) void k_os_small(uint32_t* ak, uint32_t* av, uint32_t* bk, uint32_t* bv,
                                                            const uint32_t* __restrict__ n_dev, uint32_t n_cap, uint32_t passes,
                                                            bool alone, const uint32_t* __restrict__ A, uint32_t n_slots, Front front,
                                                            uint32_t* __restrict__ tileL, uint32_t* __restrict__ tileA) {
    constexpr int RADIX = 1 << BITS;
    // ONE buffer for the pairs: a pass ranks from registers into LDS, everybody reads its rows back, the next pass
    // overwrites (the barriers in between are the block scan's and the one behind the ranking)
    __shared__ uint32_t s_key[OS_SMALL_MAX], s_val[OS_SMALL_MAX];
    __shared__ uint32_t wh[SMALL_WAVES][RADIX];
    __shared__ uint32_t s_wtot[SMALL_WAVES];
    const uint32_t m = min(*n_dev, n_cap);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t* kin = ak; const uint32_t* vin = av;
    uint32_t* kout = (passes & 1u) ? bk : ak; uint32_t* vout = (passes & 1u) ? bv : av;
    if (m > OS_SMALL_MAX) {                                     // block-uniform, before any barrier
        if (!alone) return;                                     // the multi-block kernels launched beside this one sort it
        uint32_t* const gbase = s_key;                          // (the pair buffers are free on this path)
        uint32_t* const ghist = s_val;
        const uint32_t wbase_t = wave * (WAVE * SMALL_KPT);
        for (uint32_t p = 0; p < passes; p++) {
            const uint32_t shift = p * BITS;
            const uint32_t* sk = (p & 1u) ? bk : ak; const uint32_t* sv = (p & 1u) ? bv : av;
            uint32_t* dk = (p & 1u) ? ak : bk; uint32_t* dv = (p & 1u) ? av : bv;
            for (int d = threadIdx.x; d < RADIX; d += SMALL_THREADS) ghist[d] = 0;
            __syncthreads();
            for (uint32_t i0 = wave * WAVE; i0 < m; i0 += SMALL_THREADS) {          // wave-uniform bounds
                const uint32_t i = i0 + lane;
                const uint32_t k = i < m ? sk[i] : 0u;
                wave_count_digit(ghist, (k >> shift) & (RADIX - 1), i < m);
            }
            __syncthreads();
            {
                const uint32_t cnt = threadIdx.x < (uint32_t)RADIX ? ghist[threadIdx.x] : 0u;
                const uint32_t start = block_excl_scan<SMALL_WAVES>(cnt, s_wtot, nullptr);
                if (threadIdx.x < (uint32_t)RADIX) gbase[threadIdx.x] = start;
            }
            __syncthreads();
            for (uint32_t tile0 = 0; tile0 < m; tile0 += OS_SMALL_MAX) {
                const uint32_t mt = min(OS_SMALL_MAX, m - tile0);
                uint32_t key[SMALL_KPT], val[SMALL_KPT];
#pragma unroll
                for (int t = 0; t < SMALL_KPT; t++) {
                    const uint32_t j = wbase_t + t * WAVE + lane;
                    key[t] = j < mt ? sk[tile0 + j] : 0xFFFFFFFFu;
                    val[t] = j < mt ? sv[tile0 + j] : 0u;
                }
                for (int d = threadIdx.x; d < SMALL_WAVES * RADIX; d += SMALL_THREADS) (&wh[0][0])[d] = 0;
                __syncthreads();
#pragma unroll
                for (int t = 0; t < SMALL_KPT; t++) {
                    if (wbase_t + t * WAVE >= mt) break;                            // wave-uniform
                    wave_count_digit(wh[wave], (key[t] >> shift) & (RADIX - 1), wbase_t + t * WAVE + lane < mt);
                }
                __syncthreads();
                if (threadIdx.x < (uint32_t)RADIX) {            // first output position of (wave, digit); the digit's run moves on
                    uint32_t run = gbase[threadIdx.x];
#pragma unroll
                    for (int w = 0; w < SMALL_WAVES; w++) { const uint32_t c = wh[w][threadIdx.x]; wh[w][threadIdx.x] = run; run += c; }
                    gbase[threadIdx.x] = run;
                }
                __syncthreads();
                const lds_u32_ptr pos = (lds_u32_ptr)wh[wave];
#pragma unroll
                for (int t = 0; t < SMALL_KPT; t++) {
                    if (wbase_t + t * WAVE >= mt) break;                            // wave-uniform
                    const bool valid = wbase_t + t * WAVE + lane < mt;
                    const uint32_t d = (key[t] >> shift) & (RADIX - 1);
                    uint32_t dif_lo = 0u, dif_hi = 0u;
#pragma unroll
                    for (int b = 0; b < BITS; b++) {
                        const uint32_t bm = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);
                        const uint64_t mm = __ballot(bm != 0u);
                        dif_lo |= (uint32_t)mm ^ bm;
                        dif_hi |= (uint32_t)(mm >> 32) ^ bm;
                    }
                    const uint64_t vmask = __ballot(valid);
                    const uint32_t peers_lo = ~dif_lo & (uint32_t)vmask, peers_hi = ~dif_hi & (uint32_t)(vmask >> 32);
                    const uint32_t rank = (uint32_t)__popc(peers_lo & (uint32_t)lt_mask) + (uint32_t)__popc(peers_hi & (uint32_t)(lt_mask >> 32));
                    uint32_t base = 0;
                    if (valid) base = pos[d];
                    os_wave_lds_order();
                    if (valid && rank == 0) pos[d] = base + (uint32_t)__popc(peers_lo) + (uint32_t)__popc(peers_hi);
                    os_wave_lds_order();
                    if (valid) { dk[base + rank] = key[t]; dv[base + rank] = val[t]; }
                }
                __syncthreads();                                // wh is zeroed for the next tile
            }
            __threadfence();                                    // the next pass (other waves of this block) reads what this one wrote
            __syncthreads();
        }
        if (A) {                                                // the merge's coarse ranks (k_mm_tile_rank's job)
            const uint32_t ntiles = (n_slots + OS_TILE - 1) / OS_TILE;
            for (uint32_t t = threadIdx.x; t <= ntiles; t += SMALL_THREADS) {
                const uint32_t slot = t * OS_TILE;
                const uint32_t a = t == ntiles ? 0xFFFFFFFFu : A[slot];
                tileA[t] = a;
                tileL[t] = t == ntiles ? m : mm_lower_bound(kout, vout, 0u, m, a, slot, front);
            }
        }
        return;
    }
    // rows of 64 pairs per wave: as few as the count needs, so that all 16 waves share a small sort (a wave ranks its rows
    // one after the other: 2000 movers are 2 rows for each of 16 waves, not 8 rows for 4 of them)
    const uint32_t R = max((((m + 63u) >> 6) + SMALL_WAVES - 1u) / SMALL_WAVES, 1u);        // <= SMALL_KPT
    const uint32_t wbase = wave * (WAVE * R);
    uint32_t key[SMALL_KPT], val[SMALL_KPT];
#pragma unroll
    for (int t = 0; t < SMALL_KPT; t++) {
        const uint32_t i = wbase + t * WAVE + lane;
        const bool mine = (uint32_t)t < R && i < m;
        key[t] = mine ? kin[i] : 0xFFFFFFFFu;
        val[t] = mine ? vin[i] : 0u;
    }
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = p * BITS;
        for (int d = threadIdx.x; d < SMALL_WAVES * RADIX; d += SMALL_THREADS) (&wh[0][0])[d] = 0;
        __syncthreads();                                        // (also: every row of the last pass has been read back)
#pragma unroll
        for (int t = 0; t < SMALL_KPT; t++) {
            if ((uint32_t)t >= R || wbase + t * WAVE >= m) break;                   // wave-uniform
            wave_count_digit(wh[wave], (key[t] >> shift) & (RADIX - 1), wbase + t * WAVE + lane < m);
        }
        __syncthreads();
        // per digit: exclusive offsets of the waves, then the digit's start in the tile
        uint32_t cnt = 0;
        const bool has_d = threadIdx.x < (uint32_t)RADIX;
        if (has_d) {
#pragma unroll
            for (int w = 0; w < SMALL_WAVES; w++) { const uint32_t c = wh[w][threadIdx.x]; wh[w][threadIdx.x] = cnt; cnt += c; }
        }
        const uint32_t start = block_excl_scan<SMALL_WAVES>(cnt, s_wtot, nullptr);
        if (has_d) {
#pragma unroll
            for (int w = 0; w < SMALL_WAVES; w++) wh[w][threadIdx.x] += start;
        }
        __syncthreads();
        const lds_u32_ptr pos = (lds_u32_ptr)wh[wave];
#pragma unroll
        for (int t = 0; t < SMALL_KPT; t++) {
            const uint32_t i = wbase + t * WAVE + lane;
            if ((uint32_t)t >= R || wbase + t * WAVE >= m) break;   // wave-uniform: the rows behind the last pair
            const bool valid = i < m;
            const uint32_t d = (key[t] >> shift) & (RADIX - 1);
            uint32_t dif_lo = 0u, dif_hi = 0u;
#pragma unroll
            for (int b = 0; b < BITS; b++) {
                const uint32_t bm = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1);
                const uint64_t mm = __ballot(bm != 0u);
                dif_lo |= (uint32_t)mm ^ bm;
                dif_hi |= (uint32_t)(mm >> 32) ^ bm;
            }
            const uint64_t vmask = __ballot(valid);
            const uint32_t peers_lo = ~dif_lo & (uint32_t)vmask, peers_hi = ~dif_hi & (uint32_t)(vmask >> 32);
            const uint32_t rank = (uint32_t)__popc(peers_lo & (uint32_t)lt_mask) + (uint32_t)__popc(peers_hi & (uint32_t)(lt_mask >> 32));
            uint32_t base = 0;
            if (valid) base = pos[d];
            os_wave_lds_order();
            if (valid && rank == 0) pos[d] = base + (uint32_t)__popc(peers_lo) + (uint32_t)__popc(peers_hi);
            os_wave_lds_order();
            if (valid) { s_key[base + rank] = key[t]; s_val[base + rank] = val[t]; }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < SMALL_KPT; t++) {
            const uint32_t i = wbase + t * WAVE + lane;
            const bool mine = (uint32_t)t < R && i < m;
            key[t] = mine ? s_key[i] : 0xFFFFFFFFu;
            val[t] = mine ? s_val[i] : 0u;
        }
    }
#pragma unroll
    for (int t = 0; t < SMALL_KPT; t++) {
        const uint32_t i = wbase + t * WAVE + lane;
        if ((uint32_t)t < R && i < m) { kout[i] = key[t]; vout[i] = val[t]; }
    }
    // the merge's coarse ranks, from the sorted movers in LDS (k_mm_tile_rank leaves at once for these counts)
    if (A) {
        const uint32_t ntiles = (n_slots + OS_TILE - 1) / OS_TILE;
        for (uint32_t t = threadIdx.x; t <= ntiles; t += SMALL_THREADS) {
            const uint32_t slot = t * OS_TILE;
            const uint32_t a = t == ntiles ? 0xFFFFFFFFu : A[slot];
            tileA[t] = a;
            tileL[t] = t == ntiles ? m : small_lower_bound(s_key, s_val, m, a, slot, front);
        }
    }
}

// ---- reorder: gather the SoA payload into sorted order (full-sort path) -----------------------------
__global__ __launch_bounds__(256) void k_reorder(const uint32_t* __restrict__ ks, const uint32_t* __restrict__ vs,
                                                 uint32_t n, const float4* __restrict__ posi,
                                                 const float4* __restrict__ velr, float4* __restrict__ posi_out,
                                                 float4* __restrict__ velr_out, uint32_t* __restrict__ key_out) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t src = vs[i];
    posi_out[i] = posi[src];
    velr_out[i] = velr[src];
    key_out[i] = ks[i];
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

// LSD radix sort of (key, value) pairs over the context's significant key bits.  `first`: the values of
// the first pass are the element indices (vin unused).  n_dev != null: the count lives on the device
// (<= n) and `grid` blocks share the tiles by ticket.  Returns through kin/vin the buffers that hold the result.
// what k_os_small also does for the merge: the coarse ranks of the tile boundaries (A = null: nothing)
struct SmallTail {
    const uint32_t* A = nullptr;
    uint32_t n_slots = 0;
    Front front{0u, 0u};
    uint32_t hint = 0xFFFFFFFFu;     // the count the device last reported (one step old in a host-paced context)
    int* form = nullptr;             // out: which kernels were launched (SORT_FORM_*)
};
enum { SORT_FORM_BOTH = 0, SORT_FORM_SMALL = 1, SORT_FORM_BIG = 2 };

template <int BITS>
static int radix_sort_bits(sph_ctx* c, uint32_t n, const uint32_t* n_dev, uint32_t grid, bool first, uint32_t passes,
                           uint32_t*& kin, uint32_t*& vin, uint32_t*& kout, uint32_t*& vout, const SmallTail& tail) {
    constexpr uint32_t RADIX = 1u << BITS;
    // counts the device may hand to the one-block sort (only when the count lives on the device: the movers' sort)
    uint32_t small_max = (n_dev && !first) ? OS_SMALL_MAX : 0u;
    // The count lives on the device, so in general BOTH forms are launched and each looks at the count first -- six
    // dispatches that do nothing when the one-block sort takes it (37 us of a 0.63 ms slab step).  A host-paced context
    // (a slab: its host waits for the device once per step) knows the count of the PREVIOUS sort exactly, and the movers
    // of a flow change slowly except at the first step of a burst: there only the form that count asks for is launched.
    // Each form is correct for any count on its own (k_os_small: `alone`); a wrong guess costs time once per burst.
    // A whole-domain context is not host-paced, but its host never runs more than four sorts ahead of the device
    // (launch_sort: the mm_done ring), so the count it reads is at most four steps old: it picks one form as well, with a
    // quarter of the one-block sort's capacity as the margin for what four steps can change, and both forms in between.
    int form = SORT_FORM_BOTH;
    if (small_max && c->host_paced) form = tail.hint <= OS_SMALL_MAX ? SORT_FORM_SMALL : SORT_FORM_BIG;
    else if (small_max && tail.hint != 0xFFFFFFFFu && c->sort_calls > c->sort_form_both_until)
        form = tail.hint <= OS_SMALL_MAX - OS_SMALL_MAX / 4u ? SORT_FORM_SMALL
             : (tail.hint > OS_SMALL_MAX + OS_SMALL_MAX / 4u ? SORT_FORM_BIG : SORT_FORM_BOTH);
    if (tail.form) *tail.form = form;
    if (small_max) c->sort_forms[form]++;
    if (form == SORT_FORM_BIG) small_max = 0u;               // the multi-block kernels do not look for a small count
    uint32_t* const kin0 = kin; uint32_t* const vin0 = vin;
    const bool one_group = grid <= OS_ONE_GROUP_TILES;          // `grid` = tiles expected (exact, or from the hint)
    const uint32_t group_tiles = one_group ? OS_ALL_TILES : OS_GROUP;
    const uint32_t hist_grid = min(ceil_div(grid, OS_HIST_TILES), 4096u);
    const uint32_t gcap = c->os_groups_cap;                     // tickets: [pass][group]
    if (form == SORT_FORM_SMALL) {                              // only the buffers take their turns
        for (uint32_t p = 0; p < passes; p++) {
            uint32_t* t;
            t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
    }
    if (one_group && form != SORT_FORM_SMALL) {                 // every pass from one histogram of the input
        hipLaunchKernelGGL(k_os_hist<BITS>, dim3(hist_grid), dim3(SORT_THREADS), 0, c->stream, kin, n, n_dev, 0u, passes,
                           group_tiles, c->os_hist, small_max);
        hipLaunchKernelGGL(k_os_scan<BITS>, dim3(RADIX / 4, passes), dim3(256), 0, c->stream, c->os_hist, c->os_base, c->os_tot,
                           n, n_dev, c->os_tickets, gcap, 0u, group_tiles, small_max);
        SPH_HIP(hipGetLastError());
    }
    for (uint32_t p = 0; p < passes && form != SORT_FORM_SMALL; p++) {
        if (!one_group) {                                       // the groups' counts of the keys as this pass finds them
            hipLaunchKernelGGL(k_os_hist<BITS>, dim3(hist_grid), dim3(SORT_THREADS), 0, c->stream, kin, n, n_dev, p, 1u,
                               group_tiles, c->os_hist, small_max);
            hipLaunchKernelGGL(k_os_scan<BITS>, dim3(RADIX / 4, 1), dim3(256), 0, c->stream, c->os_hist, c->os_base, c->os_tot,
                               n, n_dev, c->os_tickets, gcap, p, group_tiles, small_max);
            SPH_HIP(hipGetLastError());
        }
        // the epoch tags the look-back words of this pass: 19 bits in the grouped form (never 0: that is what a cleared
        // table holds); when they wrap the table is cleared so that no word of 2^19 passes ago can be taken for new
        c->os_epoch++;
        if ((c->os_epoch & 0x7FFFFu) == 0u) {
            c->os_epoch++;
            SPH_HIP(hipMemsetAsync(c->os_status32, 0, (size_t)512 * c->sort_blocks_cap * sizeof(uint32_t), c->stream));
        }
        const uint32_t epoch = one_group ? (c->os_epoch & 0x7FFFFFFFu) : (c->os_epoch & 0x7FFFFu);
        uint32_t* tk = c->os_tickets + (size_t)p * gcap;
#define SPH_OS_LAUNCH(F, G)                                                                                             \
        hipLaunchKernelGGL((k_os_pass<BITS, F, G>), dim3(G ? min((grid + 7u) & ~7u, OS_PASS_GRID_MAX) : grid), dim3(PASS_THREADS), 0, c->stream, kin, vin, kout, vout, n, \
                           n_dev, p * BITS, c->os_base + p * 512u, c->os_tot + p * 512u, group_tiles, c->os_status,         \
                           c->os_status32, tk, epoch,                                                                    \
                           c->os_err_dev, small_max)
        if (first && p == 0) { if (one_group) SPH_OS_LAUNCH(true, false); else SPH_OS_LAUNCH(true, true); }
        else { if (one_group) SPH_OS_LAUNCH(false, false); else SPH_OS_LAUNCH(false, true); }
#undef SPH_OS_LAUNCH
        SPH_HIP(hipGetLastError());
        uint32_t* t;
        t = kin; kin = kout; kout = t;
        t = vin; vin = vout; vout = t;
    }
    if (small_max) {
        // the one-block sort: from the ORIGINAL input into the buffers the passes above would have ended in (they left at
        // once if this kernel takes the count, and this kernel leaves at once if it does not -- unless it is alone)
        uint32_t* const bk = (passes & 1u) ? kin : kout; uint32_t* const bv = (passes & 1u) ? vin : vout;    // the buffers that are not the input
        hipLaunchKernelGGL(k_os_small<BITS>, dim3(1), dim3(SMALL_THREADS), 0, c->stream, kin0, vin0, bk, bv, n_dev, n, passes,
                           form == SORT_FORM_SMALL, tail.A, tail.n_slots, tail.front, c->mm_tileL, c->mm_tileA);
        SPH_HIP(hipGetLastError());
    }
    return SPH_OK;
}

static int radix_sort_pairs(sph_ctx* c, uint32_t n, const uint32_t* n_dev, uint32_t grid, bool first,
                            uint32_t*& kin, uint32_t*& vin, uint32_t*& kout, uint32_t*& vout, const SmallTail& tail = SmallTail()) {
    // 9-bit digits when they save a pass over 8-bit ones (27 bits: 3 x 9), else 8-bit digits
    const uint32_t p8 = (c->key_bits + 7) / 8, p9 = (c->key_bits + 8) / 9;
    if (p9 < p8) return radix_sort_bits<9>(c, n, n_dev, grid, first, p9, kin, vin, kout, vout, tail);
    return radix_sort_bits<8>(c, n, n_dev, grid, first, p8, kin, vin, kout, vout, tail);
}

// ---- the sort as a merge: only the particles whose cell changed are sorted -----------------------------------
// Between two steps a particle moves a small fraction of a cell, so after the integrate almost every key
// equals the key its slot was sorted under.  With A = the keys of the current (sorted) order and B = the
// new keys, the non-movers (A[i] == B[i]) are already in order; the movers are compacted (stable), radix
// sorted on their own, and both sequences get their merged positions by rank:
//   non-mover i :  (i - #movers before i) + #movers with (key, slot) < (A[i], i)
//   mover r     :  r + #non-movers with (key, slot) < (B, slot)
// The result is the particle order of the full stable sort, element for element, for ANY number of movers;
// launch_sort only prefers the full sort when the last known mover count makes it cheaper.  Every particle
// is moved ONCE, straight to its final slot: the non-movers stream through k_mm_scatter (coalesced reads,
// writes to slots that rise with the source slot), the movers are placed one by one.

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
                                                      unsigned long long* __restrict__ m_total, uint32_t seq) {
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
        if (seq) { __threadfence_system(); m_host[4] = seq; }     // "the count of scan number seq is in [0]" (launch_sort's skip)
    }
}

// thread per chunk: movers before the chunk (M64) and the stable list of movers (new key, slot).
// Where a block's movers start: tile_off[block] when k_mm_tilescan ran (a fluid at rest: the count is wanted at the END of the
// step, so that a lockstep caller can skip the sort), else -- tile_cnt != null -- every block adds up the per-tile counts in front
// of it ITSELF (nt <= MM_FUSED_SCAN_TILES words out of the L2: cheaper than the one-block scan kernel it replaces, ~4.5 us per
// step), block 0 publishes the total, and k_mm_move, the next kernel to run whatever the sort form, re-zeroes the counts.
constexpr uint32_t MM_FUSED_SCAN_TILES = 2048;       // up to 33.5 M slots; beyond that nt^2 reads would cost more than the scan kernel
__global__ __launch_bounds__(256) void k_mm_compact(const uint64_t* __restrict__ mask, uint32_t nchunks, uint32_t n_old,
                                                    const uint32_t* __restrict__ tile_off, const uint32_t* __restrict__ tile_cnt,
                                                    uint32_t nt, uint32_t* __restrict__ m_dev, volatile uint32_t* __restrict__ m_host,
                                                    unsigned long long* __restrict__ m_total,
                                                    const uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                    uint32_t* __restrict__ M64, uint32_t* __restrict__ mk,
                                                    uint32_t* __restrict__ mi, uint2* __restrict__ cells,
                                                    const uint32_t* __restrict__ keys_abs, uint32_t g0lo, uint32_t g0hi,
                                                    uint32_t g1lo, uint32_t g1hi) {
    if (blockIdx.x >= nt) {
        // spare blocks (a slab step): the cells of the OLD ghost ranges [g0lo, g0hi) and [g1lo, g1hi) die here -- what
        // k_cells_clear2 did in a launch of its own at hash time (sph_ctx::defer_ghost_clear).  Ghost cells hold no owned
        // particle, so nothing the other blocks clear or read is touched.
        const uint32_t t = (blockIdx.x - nt) * 256u + threadIdx.x, n0 = g0hi - g0lo;
        const uint32_t lo = t < n0 ? g0lo : g1lo, hi = t < n0 ? g0hi : g1hi;
        const uint32_t s = t < n0 ? g0lo + t : g1lo + (t - n0);
        if (s >= hi) return;
        const uint32_t k = keys_abs[s];
        if (s == lo || keys_abs[s - 1] != k) cells[k] = make_uint2(0u, 0u);
        return;
    }
    __shared__ uint32_t part[256];
    uint32_t tile_start;
    if (tile_cnt) {
        uint32_t below = 0, all = 0;
        for (uint32_t t = threadIdx.x; t < nt; t += 256u) { const uint32_t v = tile_cnt[t]; all += v; below += t < blockIdx.x ? v : 0u; }
#pragma unroll
        for (int off = 32; off; off >>= 1) { below += (uint32_t)__shfl_xor((int)below, off); all += (uint32_t)__shfl_xor((int)all, off); }
        if ((threadIdx.x & 63u) == 0u) { part[threadIdx.x >> 6] = below; part[4 + (threadIdx.x >> 6)] = all; }
        __syncthreads();
        tile_start = part[0] + part[1] + part[2] + part[3];
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const uint32_t m = part[4] + part[5] + part[6] + part[7];
            *m_dev = m; *m_host = m;
            if (m_total) *m_total += m;               // one writer: no atomic needed (sph_sort_stats: movers_total)
        }
        __syncthreads();                              // part is reused below
    } else {
        tile_start = tile_off[blockIdx.x];
    }
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
    uint32_t at = tile_start + part[threadIdx.x] - cnt;
    M64[chunk] = at;
    while (m) {
        const uint32_t i = chunk * 64u + (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        mk[at] = B[i];
        mi[at] = i;
        at++;
        // only a cell a mover left can have become empty: clear those, the cell pass after the merge rewrites
        // every cell that is still occupied (replaces the walk over all old keys, k_cells_clear)
        // (slots from n_old on hold particles that just arrived from another slab: they have no old cell)
        if (cells && i < n_old) cells[A[i]] = make_uint2(0u, 0u);
    }
}

// first r in [lo, hi) with (mk[r], mi[r]) >= (key, slot)
__device__ __forceinline__ uint32_t mm_lower_bound(const uint32_t* __restrict__ mk, const uint32_t* __restrict__ mi,
                                                   uint32_t lo, uint32_t hi, uint32_t key, uint32_t slot, Front f) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t k = mk[mid];
        const bool less = k < key || (k == key && mover_slot_less(mi[mid], slot, f));
        if (less) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Coarse ranks: tileL[t] = #movers with (key, slot) < (A[t * 4096], t * 4096), tileL[ntiles] = m.  The per-wave
// searches of k_mm_scatter then run inside [tileL[t], tileL[t+1]] -- a handful of movers instead of all of them
// (16 dependent loads per search at 40 K movers made the placement latency-bound: 180 us at C3).
constexpr uint32_t MM_RANK_TILE = 4096;
__global__ __launch_bounds__(256) void k_mm_tile_rank(const uint32_t* __restrict__ A, uint32_t n,
                                                      const uint32_t* __restrict__ mk, const uint32_t* __restrict__ mi,
                                                      const uint32_t* __restrict__ m_dev, uint32_t* __restrict__ tileL,
                                                      uint32_t* __restrict__ tileA, bool small_too, uint32_t small_max,
                                                      Front front) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t ntiles = (n + MM_RANK_TILE - 1) / MM_RANK_TILE;
    if (t > ntiles) return;
    const uint32_t m = *m_dev;
    if (small_too && m <= small_max) return;                 // k_os_small (launched beside the passes) ranked the tiles from its LDS copy
    const uint32_t slot = t * MM_RANK_TILE;
    const uint32_t a = t == ntiles ? 0xFFFFFFFFu : A[slot];
    tileA[t] = a;                      // first key of every tile: the coarse level of k_mm_place_movers' searches
    tileL[t] = t == ntiles ? m : mm_lower_bound(mk, mi, 0u, m, a, slot, front);
}

// non-movers: one thread per slot, the particle goes straight to its final slot.  The bracket [L0, L1] of a
// whole wave is found with wave-uniform (scalar) searches inside the tile's coarse bracket; it is a single
// point unless a mover lands inside the wave's key span.
__device__ __forceinline__ void mm_scatter_body(uint32_t bid, const uint32_t* __restrict__ A, uint32_t n,
                                                const uint64_t* __restrict__ mask, const uint32_t* __restrict__ M64,
                                                const uint32_t* __restrict__ mk, const uint32_t* __restrict__ mi,
                                                const uint32_t* __restrict__ tileL, const float4* __restrict__ posi,
                                                const float4* __restrict__ velr, float4* __restrict__ posi_out,
                                                float4* __restrict__ velr_out, uint32_t* __restrict__ key_out,
                                                uint32_t* __restrict__ perm_out, Front front) {
    const uint32_t i = bid * 256u + threadIdx.x;
    const uint32_t first = __builtin_amdgcn_readfirstlane(i);          // slot of lane 0
    if (first >= n) return;                                            // wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = first >> 6;
    const uint32_t last = min(first + 63u, n - 1u);
    const uint32_t tile = first / MM_RANK_TILE;
    const uint32_t b0 = tileL[tile], b1 = tileL[tile + 1];
    uint32_t L0 = b0, L1 = b0;
    if (b0 != b1) {                                                    // wave-uniform
        L0 = mm_lower_bound(mk, mi, b0, b1, A[first], first, front);
        L1 = mm_lower_bound(mk, mi, L0, b1, A[last], last, front);
    }
    if (i >= n) return;
    const uint64_t bits = mask[chunk];
    if ((bits >> lane) & 1ull) return;                                 // a mover: placed by k_mm_place_movers
    const uint32_t key = A[i];
    const uint32_t before = M64[chunk] + (uint32_t)__popcll(bits & ((1ull << lane) - 1ull));
    const uint32_t L = L0 == L1 ? L0 : mm_lower_bound(mk, mi, L0, L1, key, i, front);
    const uint32_t dst = i - before + L;
    posi_out[dst] = posi[i];
    velr_out[dst] = velr[i];
    key_out[dst] = key;
    if (perm_out) perm_out[dst] = i;
}

// movers: one thread each (grid-stride: the grid is generous, not exact).  A mover's slot follows from the run
// [s, e) of its NEW cell in the old order.  Where the cell table of the old order is live and still holds that cell
// (it was occupied, and no mover left it -- k_mm_compact cleared those), the run is ONE table read; else it is
// searched: the first key of every 4096-slot tile (tileA, 16 KB at C3, written by k_mm_tile_rank) brackets it, one
// binary search inside the tile finds s, and e is galloped to from s (a cell holds a handful of particles).  Two
// full-array searches per mover -- 2 x 24 dependent loads at C3 -- made this kernel 167 us on burst steps.
__device__ __forceinline__ void mm_place_body(uint32_t bid, uint32_t nblocks, const uint32_t* __restrict__ A, uint32_t n,
                                              uint32_t nchunks, const uint64_t* __restrict__ mask,
                                              const uint32_t* __restrict__ M64, const uint32_t* __restrict__ mk,
                                              const uint32_t* __restrict__ mi, const uint32_t* __restrict__ m_dev,
                                              const uint32_t* __restrict__ tileA, const uint2* __restrict__ cells,
                                              uint32_t slot_base, const float4* __restrict__ posi,
                                              const float4* __restrict__ velr, float4* __restrict__ posi_out,
                                              float4* __restrict__ velr_out, uint32_t* __restrict__ key_out,
                                              uint32_t* __restrict__ perm_out, Front front) {
    const uint32_t m = *m_dev;
    const uint32_t ntiles = (n + MM_RANK_TILE - 1) / MM_RANK_TILE;
    for (uint32_t r = bid * 256u + threadIdx.x; r < m; r += nblocks * 256u) {
        const uint32_t key = mk[r], slot = mi[r];
        uint32_t s, e;
        uint2 ce = make_uint2(0u, 0u);
        if (cells) ce = cells[key];
        if (ce.y > ce.x) {                              // the old order's run of this cell (absolute slots)
            s = ce.x - slot_base; e = ce.y - slot_base;
        } else {
            // coarse: last tile whose first key is < key (s lies in it or at the start of the next one)
            uint32_t lo = 0, hi = ntiles;               // first tile t with tileA[t] >= key
            while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (tileA[mid] < key) lo = mid + 1; else hi = mid; }
            uint32_t a = lo ? (lo - 1u) * MM_RANK_TILE : 0u, b = min(lo * MM_RANK_TILE, n);
            while (a < b) { const uint32_t mid = a + ((b - a) >> 1); if (A[mid] < key) a = mid + 1; else b = mid; }
            s = a;                                      // first slot with A >= key
            uint32_t step = 1u, p = s;                  // gallop to the first slot with A > key
            while (p < n && A[p] == key) { p = min(p + step, n); step <<= 1; }
            uint32_t q = p > s ? max(s, p - (step >> 1)) : s;   // A[q .. ) still may equal key; A[p] != key or p == n
            a = q; b = p;
            while (a < b) { const uint32_t mid = a + ((b - a) >> 1); if (A[mid] <= key) a = mid + 1; else b = mid; }
            e = a;
        }
        // non-movers of cell `key` below `slot` end here (a front mover -- an arrival from the slab below -- precedes them all)
        const uint32_t j = slot - front.lo < front.cnt ? s : min(max(slot, s), e);
        uint32_t before = m;                           // movers among the slots [0, j)  (j <= n: the slots with an old key)
        if ((j >> 6) < nchunks) before = M64[j >> 6] + (uint32_t)__popcll(mask[j >> 6] & ((1ull << (j & 63u)) - 1ull));
        const uint32_t dst = r + (j - before);
        posi_out[dst] = posi[slot];
        velr_out[dst] = velr[slot];
        key_out[dst] = key;
        if (perm_out) perm_out[dst] = slot;
    }
}

// ONE launch moves everybody: the first `place_blocks` blocks place the movers (few, latency-bound: dependent
// searches), the others stream the non-movers -- the two read the same old arrays and write disjoint slots of the new
// ones, so the movers' placement hides behind the stream instead of running in front of it (26 us at C3).
__global__ __launch_bounds__(256) void k_mm_move(uint32_t place_blocks, uint32_t* __restrict__ zero_cnt, uint32_t zero_n,
                                                 const uint32_t* __restrict__ A, uint32_t n,
                                                 uint32_t nchunks, const uint64_t* __restrict__ mask,
                                                 const uint32_t* __restrict__ M64, const uint32_t* __restrict__ mk,
                                                 const uint32_t* __restrict__ mi, const uint32_t* __restrict__ m_dev,
                                                 const uint32_t* __restrict__ tileL, const uint32_t* __restrict__ tileA,
                                                 const uint2* __restrict__ cells, uint32_t slot_base,
                                                 const float4* __restrict__ posi, const float4* __restrict__ velr,
                                                 float4* __restrict__ posi_out, float4* __restrict__ velr_out,
                                                 uint32_t* __restrict__ key_out, uint32_t* __restrict__ perm_out, Front front) {
    // the per-tile mover counts k_mm_compact added up itself (every block of it has read them: it ran in front of this kernel)
    if (zero_cnt && blockIdx.x * 256u + threadIdx.x < zero_n) zero_cnt[blockIdx.x * 256u + threadIdx.x] = 0u;
    if (blockIdx.x < place_blocks)
        mm_place_body(blockIdx.x, place_blocks, A, n, nchunks, mask, M64, mk, mi, m_dev, tileA, cells, slot_base, posi, velr,
                      posi_out, velr_out, key_out, perm_out, front);
    else
        mm_scatter_body(blockIdx.x - place_blocks, A, n, mask, M64, mk, mi, tileL, posi, velr, posi_out, velr_out, key_out,
                        perm_out, front);
}

// Blocks for the movers' radix sort.  The count is only a hint (the previous report; whole lattice layers cross a
// cell face together: x100 from one step to the next), and blocks without a tile leave at once, so the grid is
// generous: never fewer than a full one-group grid (64 blocks: 262,144 movers at one tile each), twice the hint
// beyond that.  A tight grid made ~20 blocks work through ~200 tiles of a burst with a chained look-back.
static uint32_t merge_grid_for(uint32_t movers_hint, uint32_t n) {
    const uint32_t want = max(ceil_div(2u * movers_hint + 1u, SORT_TILE) + 15u, OS_ONE_GROUP_TILES);
    return min(want, max(ceil_div(n, SORT_TILE), 1u));
}

// `counted`: the movers belong to a sort (they add to the running total), not to marks being dropped
static void mm_tilescan(sph_ctx* c, uint32_t n, bool counted) {
    const uint32_t nt = ceil_div(ceil_div(n, 64u), MM_TILE_CHUNKS);
    // `counted`: the kernel echoes the scan's number behind the count, so that launch_sort can tell "the count of the CURRENT
    // marks is there" by looking at two words of mapped host memory (no event: see sph_ctx::scan_seq_issued)
    uint32_t seq = 0u;
    if (counted) { if (++c->scan_seq_issued == 0u) c->scan_seq_issued = 1u; seq = c->scan_seq_issued; }     // (never 0: that is "no echo")
    hipLaunchKernelGGL(k_mm_tilescan, dim3(1), dim3(1024), 0, c->stream, c->mm_tile_cnt, nt, c->mm_tile_off, c->mm_count,
                       c->mm_count_host_dev, counted ? c->mm_total : (unsigned long long*)nullptr, seq);
    c->mm_counted_valid = counted;
}

static uint32_t mm_tiles(uint32_t n) { return ceil_div(ceil_div(n, 64u), MM_TILE_CHUNKS); }

// Called right after the integrate epilogue has marked the movers.  While the fluid is AT REST (the last count the device
// reported is 0) they are counted at the END of the step, so that the next sort finds the number ready: a caller in lockstep
// with the device can then skip a sort that has nothing to do without ever waiting for the device.  Once particles change
// cell no sort is skipped, and the count is left to the sort itself (k_mm_compact adds the per-tile counts up: one dispatch
// less per step, ~4.5 us); a range of more than MM_FUSED_SCAN_TILES tiles keeps the scan kernel.
void mm_scan_marks(sph_ctx* c) {
    if (!c->mm_marked || c->mm_scanned) return;
    if (*c->mm_count_host != 0u && mm_tiles(c->mm_marked_n) <= MM_FUSED_SCAN_TILES) return;     // the sort will count them
    mm_tilescan(c, c->mm_marked_n, true);
    c->mm_scanned = true;
}

// forget the marks the integrate epilogue left (the scan re-zeroes the tile counts they added to)
void mm_drop_marks(sph_ctx* c) {
    if (!c->mm_marked) return;
    if (!c->mm_scanned) mm_tilescan(c, c->mm_marked_n, false);
    c->mm_marked = false;
    c->mm_scanned = false;
}

// step 1 of the merge: the movers are marked (by the integrate epilogue, or here) and counted -- by the scan kernel, or, when
// this returns true, by k_mm_compact itself (launch_sort_merge: `count_in_compact`)
static bool launch_merge_count(sph_ctx* c, uint32_t n) {
    if (c->mm_marked && !(c->mm_marked_off == c->own_off && c->mm_marked_n == n)) mm_drop_marks(c);   // another range
    if (!c->mm_marked) {                         // else: the fused integrate epilogue compared the keys already
        hipLaunchKernelGGL(k_mm_mark, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, c->keyS + c->own_off, c->k0, n,
                           c->mm_mask, c->mm_tile_cnt);
        c->mm_scanned = false;
    }
    bool in_compact = false;
    if (!c->mm_scanned) {
        if (mm_tiles(n) <= MM_FUSED_SCAN_TILES) { in_compact = true; c->mm_counted_valid = false; }
        else mm_tilescan(c, n, true);
    }
    c->mm_marked = false;
    c->mm_scanned = false;
    return in_compact;
}

// every slot from n_old on is a mover, none below (particles appended behind a sorted range)
__global__ __launch_bounds__(256) void k_mm_mark_tail(uint32_t n_old, uint32_t n_tot, uint64_t* __restrict__ mask,
                                                      uint32_t* __restrict__ tile_cnt) {
    const uint32_t chunk = blockIdx.x * 256u + threadIdx.x;
    if (chunk * 64u >= n_tot) return;
    const uint32_t lo = chunk * 64u, hi = min(lo + 64u, n_tot);
    uint64_t m = 0ull;
    if (hi > n_old) {
        const uint32_t first = max(lo, n_old) - lo, cnt = hi - lo;
        m = (cnt == 64u ? ~0ull : ((1ull << cnt) - 1ull)) & ~((1ull << first) - 1ull);
        atomicAdd(&tile_cnt[chunk / MM_TILE_CHUNKS], (uint32_t)__popcll(m));
    }
    mask[chunk] = m;
}

// The merged order, written straight into posi2 / velr2 / keyS2 at the canonical offset gcap.  Slots [0, n) carry
// an old key (A) and a new one (B); slots [n, n_tot) -- particles that arrived from a neighbouring slab -- only a
// new one, and all of them are movers.
static int launch_sort_merge(sph_ctx* c, uint32_t n, uint32_t n_tot, bool table_live, uint32_t hint, Front front = Front{0u, 0u},
                             bool count_in_compact = false, const uint32_t* ghost_clear = nullptr) {
    const uint32_t* A = c->keyS + c->own_off;
    const uint32_t* B = c->k0;
    const uint32_t nchunks = ceil_div(n_tot, 64u), nt = ceil_div(nchunks, MM_TILE_CHUNKS);
    uint32_t* mk = c->mm_k0; uint32_t* mi = c->v0; uint32_t* mk2 = c->mm_k1; uint32_t* mi2 = c->mm_v1;
    // (ghost_clear: the old ghosts' cells, cleared by spare blocks of this launch -- launch_sort)
    const uint32_t gc[4] = {ghost_clear ? ghost_clear[0] : 0u, ghost_clear ? ghost_clear[1] : 0u, ghost_clear ? ghost_clear[2] : 0u,
                            ghost_clear ? ghost_clear[3] : 0u};
    const uint32_t gc_blocks = ceil_div((gc[1] - gc[0]) + (gc[3] - gc[2]), 256u);
    hipLaunchKernelGGL(k_mm_compact, dim3(nt + gc_blocks), dim3(256), 0, c->stream, c->mm_mask, nchunks, n, c->mm_tile_off,
                       count_in_compact ? c->mm_tile_cnt : (const uint32_t*)nullptr, nt, c->mm_count, c->mm_count_host_dev, c->mm_total, A, B,
                       c->mm_M64, mk, mi, table_live ? c->cells : (uint2*)nullptr, c->keyS, gc[0], gc[1], gc[2], gc[3]);
    SPH_HIP(hipGetLastError());
    SmallTail tail;
    int form = SORT_FORM_BOTH;
    tail.A = A; tail.n_slots = n; tail.front = front; tail.hint = hint; tail.form = &form;
    int rc = radix_sort_pairs(c, n_tot, c->mm_count, merge_grid_for(hint, n_tot), false, mk, mi, mk2, mi2, tail);
    if (rc) return rc;
    if (form != SORT_FORM_SMALL) {                              // (the one-block sort ranks the tile boundaries itself)
        const uint32_t rank_tiles = ceil_div(n, MM_RANK_TILE) + 1u;
        hipLaunchKernelGGL(k_mm_tile_rank, dim3(ceil_div(rank_tiles, 256u)), dim3(256), 0, c->stream, A, n, mk, mi, c->mm_count,
                           c->mm_tileL, c->mm_tileA, form == SORT_FORM_BOTH, OS_SMALL_MAX, front);
        SPH_HIP(hipGetLastError());
    }
    uint32_t* perm = c->keep_perm ? c->v1 : (uint32_t*)nullptr;
    const float4* ps = c->posi + c->own_off; const float4* vs = c->velr + c->own_off;
    float4* po = c->posi2 + c->gcap; float4* vo = c->velr2 + c->gcap; uint32_t* ko = c->keyS2 + c->gcap;
    // generous grid for the movers (grid-stride loop over the device-side count): a burst is not left to a handful of blocks
    const uint32_t place_blocks = min(max(ceil_div(2u * hint + 1u, 256u) + 15u, 512u), 65535u);
    hipLaunchKernelGGL(k_mm_move, dim3(place_blocks + ceil_div(n, 256)), dim3(256), 0, c->stream, place_blocks,
                       count_in_compact ? c->mm_tile_cnt : (uint32_t*)nullptr, nt, A, n, nchunks,
                       c->mm_mask, c->mm_M64, mk, mi, c->mm_count, c->mm_tileL, c->mm_tileA,
                       table_live ? c->cells : (const uint2*)nullptr, c->own_off, ps, vs, po, vo, ko, perm, front);
    SPH_HIP(hipGetLastError());
    c->last_perm = perm;
    return SPH_OK;
}

// The mover count the host reads is whatever the device last reported: a caller that queues many steps without
// synchronising would decide all of them on one stale value, so the host never runs more than four sorts ahead of the
// device (the queue stays several steps deep: the device never waits).  The device's progress is the sort number the table
// build echoes into mapped host memory (rounds 1-5: a ring of four events, each ~5.5 us of device idle at the next dispatch).
static int sort_throttle(sph_ctx* c) {
    volatile const uint32_t* hw = c->mm_count_host;
    if ((int32_t)(c->sort_seq_issued - hw[3]) < 4) return SPH_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1; (int32_t)(c->sort_seq_issued - hw[3]) >= 4; spins++) {
        __builtin_ia32_pause();
        if ((spins & 0xFFFu) == 0u) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) break;                            // nothing queued any more (the echo of a build that never ran: upload, re-cut)
            if (q != hipErrorNotReady) SPH_HIP(q);                 // a device fault shows up here, not in the mapped word
            SPH_REQUIRE(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 300.0, SPH_E_DEVICE,
                        "sort: the device has not reached the table build of sort %u after 300 s", c->sort_seq_issued - 3u);
        }
    }
    return SPH_OK;
}

int launch_sort(sph_ctx* c) {
    const uint32_t n = c->n;
    if (n == 0) return SPH_OK;
    const uint32_t nblocks = ceil_div(n, SORT_TILE);
    SPH_REQUIRE(nblocks <= c->sort_blocks_cap, SPH_E_CAPACITY, "sort: %u blocks > capacity %u", nblocks,
                c->sort_blocks_cap);
    // The merge needs the order of the last sort to be intact; it is correct for any number of movers but
    // only cheaper than the full sort while they are few (last known count: a hint, never a condition).
    // The mover count the host reads below is whatever the device last reported.  A caller that queues many
    // steps without synchronising would decide all of them on one stale value, so the host never runs more
    // than four sorts ahead of the device (the queue stays several steps deep: the device never waits).
    if (c->sort_merge && !c->host_paced) { const int rc = sort_throttle(c); if (rc) return rc; }
    c->sort_calls++;
    c->last_sort_skipped = false;
    const bool can_merge = c->sort_merge && c->order_valid;
    // whole-domain contexts: sph_hash left the old cell table in place (cells_clear_deferred) when this
    // sort could take the merge path, which clears only the cells the movers left
    const bool table_live = c->cells_clear_deferred && c->cells_valid;
    c->cells_clear_deferred = false;
    // the old ghosts' cells, left by the hash of a slab step (sph_ctx::defer_ghost_clear): spare blocks of k_mm_compact clear
    // them when this sort merges; a sort that is skipped does it with the kernel; the full sort clears the whole table range,
    // which still includes them
    const bool ghosts_left = c->ghost_clear_pending && table_live;
    c->ghost_clear_pending = false;
    if (can_merge && (*c->mm_count_host <= n / 8u || c->sort_merge_always)) {
        const bool was_still = *c->mm_count_host == 0u;
        const bool count_in_compact = launch_merge_count(c, n);
        if (was_still && table_live && c->own_off == c->gcap && !count_in_compact) {
            // Nothing moved last time (a fluid at rest: no particle crosses a cell face for many steps).  If that
            // is still so, the order, the keys and the cell table are already those of this step and the whole
            // sort -- 0.25 ms of copying at C3 -- can be left out.  Only the device knows, and the host does not
            // wait for it: the count is looked at only if the device has ALREADY produced it (it is queued at the
            // end of the previous step, so a caller in lockstep with the device -- one update() per frame --
            // finds it); a host that runs ahead of the device queues the merge, which does the same job for 0
            // movers.  sph_step stays asynchronous.
            volatile const uint32_t* hw = c->mm_count_host;
            if (c->mm_counted_valid && hw[4] == c->scan_seq_issued && (std::atomic_thread_fence(std::memory_order_acquire), hw[0] == 0u)) {
                c->sort_merges++;
                c->sort_skips++;
                c->last_sort_skipped = true;
                c->last_perm = nullptr;            // identity
                c->order_valid = true;             // cells_valid / cells_lo / cells_hi: unchanged and still true
                if (ghosts_left) {
                    const int rc2 = launch_cells_clear_2ranges(c, c->ghost_clear[0], c->ghost_clear[1], c->ghost_clear[2], c->ghost_clear[3]);
                    if (rc2) return rc2;
                    c->cells_lo = c->own_off; c->cells_hi = c->own_off + n;
                }
                return SPH_OK;
            }
        }
        int rc = launch_sort_merge(c, n, n, table_live, *c->mm_count_host, Front{0u, 0u}, count_in_compact,   // hint: whatever step last reported
                                   ghosts_left ? c->ghost_clear : (const uint32_t*)nullptr);
        if (rc) return rc;
        c->sort_merges++;
        if (c->cells_valid && !table_live) {       // a table nobody cleared (e.g. sph_sort without sph_hash): start clean
            rc = launch_cells_clear(c);
            if (rc) return rc;
        }
    } else {
        if (c->cells_valid) {                      // live or not: the full sort rebuilds the table from nothing
            int rc = launch_cells_clear(c);
            if (rc) return rc;
        }
        // keep the hint alive, or it would stay high for ever: for free when the integrate epilogue marked
        // the movers (the scan also re-zeroes the tile counts those marks added to), else every 8th sort
        if (c->mm_marked) {
            if (!c->mm_scanned) mm_tilescan(c, c->mm_marked_n, true);
            c->mm_marked = false;
            c->mm_scanned = false;
        } else if (can_merge && (c->sort_calls & 7u) == 0) {
            hipLaunchKernelGGL(k_mm_mark, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, c->keyS + c->own_off, c->k0, n,
                               c->mm_mask, c->mm_tile_cnt);
            mm_tilescan(c, n, true);
            SPH_HIP(hipGetLastError());
        }
        uint32_t* kin = c->k0; uint32_t* vin = c->v0;
        uint32_t* kout = c->k1; uint32_t* vout = c->v1;
        int rc = radix_sort_pairs(c, n, nullptr, nblocks, true, kin, vin, kout, vout);
        if (rc) return rc;
        // (kin, vin) now hold the sorted pairs; gather the payload to the canonical offset gcap
        hipLaunchKernelGGL(k_reorder, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, kin, vin, n, c->posi + c->own_off,
                           c->velr + c->own_off, c->posi2 + c->gcap, c->velr2 + c->gcap, c->keyS2 + c->gcap);
        SPH_HIP(hipGetLastError());
        c->last_perm = vin;
    }
    float4* t4;
    t4 = c->posi; c->posi = c->posi2; c->posi2 = t4;
    t4 = c->velr; c->velr = c->velr2; c->velr2 = t4;
    uint32_t* tk = c->keyS; c->keyS = c->keyS2; c->keyS2 = tk;
    c->own_off = c->gcap;
    // the cell table of the owned slots from boundary flags on the new keys (no atomics, no scan, no host sync).
    // A slab context adds the cells of its ghost layers later (launch_cells_build), and drops those of the
    // particles that leave (sph_migrants_pack).
    c->cells_valid = false;
    if (++c->sort_seq_issued == 0u) c->sort_seq_issued = 1u;       // the table build echoes the sort's number (sort_throttle)
    c->cells_seq_next = c->sort_seq_issued;
    if (c->owned_cells_in_bounds) {
        c->owned_cells_pending = true;             // the slab step's bounds kernel, queued next, builds it (sph_slab.hip)
        c->cells_seq_next = 0u;
    } else {
        int rc = launch_cells_build_range(c, c->gcap, c->gcap + n);
        if (rc) return rc;
    }
    c->order_valid = true;
    c->cells_lo = c->gcap; c->cells_hi = c->gcap + n; c->cells_valid = true;
    return SPH_OK;
}

// `n_in` particles were appended behind the SORTED owned range (positions, velocities and their new keys in
// k0[n ...]): merge them in -- the movers are exactly the appended slots, everybody else keeps rank and key.  One
// pass over the particles instead of a full radix sort (the slab step, csrc/sph_slab.hip, every time a neighbour
// sends particles).  The cell table of the owned range must be valid; it is rebuilt for the new order.
// The first n_front of them came from the slab BELOW: among equal keys they go in front of the residents (the order
// of the whole-domain stable sort, see Front), the others behind them.
int launch_merge_arrivals(sph_ctx* c, uint32_t n_in, uint32_t n_front) {
    const uint32_t n = c->n, n_tot = n + n_in;
    SPH_REQUIRE(c->order_valid && c->cells_valid && c->cells_lo == c->own_off && c->cells_hi == c->own_off + n, SPH_E_STATE,
                "launch_merge_arrivals needs the sorted owned range and its cell table");
    SPH_REQUIRE(ceil_div(n_tot, SORT_TILE) <= c->sort_blocks_cap, SPH_E_CAPACITY, "sort: capacity exceeded");
    mm_drop_marks(c);
    const uint32_t nchunks = ceil_div(n_tot, 64u);
    hipLaunchKernelGGL(k_mm_mark_tail, dim3(ceil_div(nchunks, 256u)), dim3(256), 0, c->stream, n, n_tot, c->mm_mask,
                       c->mm_tile_cnt);
    const uint32_t nt = ceil_div(nchunks, MM_TILE_CHUNKS);
    hipLaunchKernelGGL(k_mm_tilescan, dim3(1), dim3(1024), 0, c->stream, c->mm_tile_cnt, nt, c->mm_tile_off, c->mm_count,
                       c->mm_count_host_dev, (unsigned long long*)nullptr, 0u);
    c->mm_counted_valid = false;
    SPH_HIP(hipGetLastError());
    int rc = launch_sort_merge(c, n, n_tot, true, n_in, Front{n, n_front < n_in ? n_front : n_in});
    if (rc) return rc;
    float4* t4;
    t4 = c->posi; c->posi = c->posi2; c->posi2 = t4;
    t4 = c->velr; c->velr = c->velr2; c->velr2 = t4;
    uint32_t* tk = c->keyS; c->keyS = c->keyS2; c->keyS2 = tk;
    c->own_off = c->gcap;
    c->n = n_tot;
    c->cells_valid = false;
    rc = launch_cells_build_range(c, c->gcap, c->gcap + n_tot);
    if (rc) return rc;
    c->cells_lo = c->gcap; c->cells_hi = c->gcap + n_tot; c->cells_valid = true;
    c->order_valid = true;
    c->keys_fresh = false;
    c->n_glo = c->n_ghi = 0;
    c->stage = sph_ctx::ST_SORTED;
    c->have_dens = c->have_force = c->have_coll = false;
    return SPH_OK;
}

// sorted slot -> slot before the sort (valid until the next sph_hash; null = identity); used by the compat
// seam to move the caller's AoS structs the way thrust::sort would (it sets keep_perm)
const uint32_t* last_sort_permutation(sph_ctx* c) { return c->last_perm; }

// Stable LSD radix sort of the indices 0 .. n-1 by `keys_dev[i]` (`bits` significant bits), with the context's radix
// scratch: *perm_out[a] = the index with the a-th smallest key.  The pointer is into the scratch: valid until the next
// sph_hash / sph_sort; the keys the hash left in k0 (and the permutation of the last particle sort) are overwritten.
// Used by the compat seam to number its caller's array the reference's way (Morton) on top of the native order.
int sort_indices_by_key(sph_ctx* c, const uint32_t* keys_dev, uint32_t n, uint32_t bits, const uint32_t** perm_out) {
    *perm_out = nullptr;
    if (n == 0) return SPH_OK;
    const uint32_t nblocks = ceil_div(n, SORT_TILE);
    SPH_REQUIRE(nblocks <= c->sort_blocks_cap && n <= c->cap, SPH_E_CAPACITY, "sort: %u keys > capacity %u", n, c->cap);
    SPH_HIP(hipMemcpyAsync(c->k0, keys_dev, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    uint32_t* kin = c->k0; uint32_t* vin = c->v0;
    uint32_t* kout = c->k1; uint32_t* vout = c->v1;
    const uint32_t saved = c->key_bits;
    c->key_bits = bits;
    const int rc = radix_sort_pairs(c, n, nullptr, nblocks, true, kin, vin, kout, vout);
    c->key_bits = saved;
    if (rc) return rc;
    c->keys_fresh = false;
    c->last_perm = nullptr;
    *perm_out = vin;
    return SPH_OK;
}

}  // namespace sph
