// sph_slab.hip -- one time step of a z-slab with its two neighbours, under the C ABI (sph_slab_step).
//
// No counterpart in the reference (single GPU).  The whole step is queued by C++ on two HIP streams with ONE host
// wait, and that wait is covered by device work:
//
//   main stream                                   comm stream (high priority)
//   -----------                                   -----------
//   hash + sort owned particles
//   k_slab_bounds_pack: layer bounds, leavers
//     + header {#leavers, #boundary, #far}     -> exchange MIGRANTS (header + 255 inline records: 8 KB)
//   density of the DEEP interior (layers >= 4        k_slab_post_headers: own bounds + the neighbours' headers
//     from either cut; its slot range is read        into mapped host memory, then a sequence word
//     from DEVICE memory: the host does not
//     know the bounds yet)
//   [early stream, behind the deep density by event: force+collision+integrate of the INNERMOST layers (>= 6 from either
//    cut; every density they read is the deep launch's) -- keys by absolute slot, no mover marks; k_slab_early_finish]
//   ............ host polls the sequence word (bounded): the only wait of the step, hidden behind the deep density ...
//   [more than 255 leavers on a side: the rest of them in a second, exact-size message]
//   --- a step WITHOUT arrivals (the usual one): the halo work goes to the comm stream at once ---
//                                                 clear the leavers' cells ; pack both boundary layers
//                                                 exchange HALO A (positions, velocities; exact size)
//                                                 unpack ghosts + their cell table (one kernel)
//                                                 density of everything that is not deep (ONE launch, beside the
//                                                   tail of the deep launch) ; pack (rho, p) of the boundary layers
//   <-(event) force+collision+integrate, interior    exchange HALO B ; unpack ghost (rho, p)
//                                                 force+collision+integrate, the boundary chunks (one launch)
//   <-(event) mover count of the next sort
//   --- a step WITH arrivals ---
//   arrivals merged into their boundary layer in place (a "far" one -- it crossed more than one layer -- sends the
//   step through the pass over all particles instead) ; pack boundary layers -> HALO A ; density interior-minus-deep
//   while it travels ; (event) density boundary ; pack (rho, p) -> HALO B ; force interior ; force boundary on comm
//
// Round 6: a second protocol beside this one (sph_slab_set_protocol(s, 1); the context then keeps TWO ghost layers): ONE message per
// neighbour and step -- header, leavers and the residents of the two layers next to the cut as they are right after the sort, its
// size fixed in advance by a rule on the previous step's counts -- in the place of MIGRANTS; no HALO A (the receiver merges its own
// leavers into its copy of the neighbour's layers: k_slab_unpack_ghosts_merge) and no HALO B (the density launch over "everything
// that is not deep" also covers the inner ghost layers: same candidates, same order, the neighbour's bits).  DESIGN.md section 6
// has both side by side and the measured table; `p1` marks that protocol's branches in slab_step_body.
//
// Equal keys at a cut keep the order of the whole-domain stable sort (what came up from below in front of the residents
// of its cell, what came down from above behind them): an N-slab run has the bits of the one-context run.
// A rank that fails still exchanges what the step owes, then sends "abort" in its next migrant header (slab_fail): its
// neighbours return SPH_E_PEER one step later instead of waiting for a timeout; a dead transport is aborted.
//
// Also here: the neighbour ping (sph_slab_ping), per-group and host-wait timing (sph_slab_timing_*), re-balancing on the
// device (sph_slab_recut), and the loop transport (one slab between its periodic images: a middle rank's whole step on one GPU).
//
// The interior layers (all but the first and last owned layer) never look at a ghost, so their passes run while
// the halos travel.  Messages go point to point to the two z-neighbours only: RCCL ncclSend/ncclRecv in one group
// on the comm stream (sph_rccl_transport_create; over the direct xGMI link), or through a caller-supplied
// transport (tests: host-staged -- several processes over gloo -- or device-to-device between the streams of one
// process, sph_local_transport_create).
#include "sph_device.hpp"
#include <vector>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstring>
#include <new>

namespace sph {

// words of the pinned, mapped host block the step's one wait reads (sph_slab::h_lb)
enum {
    HL_LB = 0,        // [0..3] own layer bounds: first slot (relative to the owned range) with a key >= layer, 2*layer,
                      //        (zl-2)*layer, (zl-1)*layer
    HL_DEEP = 4,      // [4..5] the deep interior [first key >= 4*layer, first key >= (zl-4)*layer), ABSOLUTE slots
    HL_FAR = 6,       // [6..7] my leavers (down, up) that are NOT in the neighbour's adjacent layer (crossed > 1 layer)
    HL_HDR_LO = 8,    // [8..11]  header received from the lower neighbour {#arrivals, #its boundary layer, #far, abort}
    HL_HDR_HI = 12,   // [12..15] ... from the upper neighbour
    HL_SEQ = 16,      // written last: the step number
    HL_ERR = 17,      // [17..18] sticky error words set by device-side checks (plain stores of 1): SLAB_ERR_*
    HL_NEAR = 20,     // [20..21] first slot of local layer 3 / of local layer zl-3 (ABSOLUTE): the layers next to the deep interior
    HL_EARLY = 22,    // [22..23] the slots whose force pass may run in front of the wait: local layers [6, zl-6) (ABSOLUTE; empty: equal)
    HL_RECUT = 24,    // [24..25] sph_slab_recut: particles that go down / up ; [26..27] what the neighbours send (from below, from above)
    HL_HDR2_LO = 32,  // [32..35] second header word group from the lower neighbour {#its second layer, #very far leavers, 0, 0}
    HL_HDR2_HI = 36,  // [36..39] ... from the upper neighbour
    HL_VFAR = 40,     // [40..41] my leavers (down, up) that are not even in the neighbour's SECOND layer (crossed > 2 layers)
    HL_WORDS = 48
};
enum { SLAB_ERR_INSERT_LAYER = 0, SLAB_ERR_ARRIVAL_OUTSIDE = 1 };
// device words (sph_slab::d_lb): [0..3] bounds, [4..5] deep range (absolute), [6..7] far counts, [8..10] the
// fused kernel's block counters {far down, far up, blocks done}
enum { DL_DEEP = 4, DL_FAR = 6, DL_CTR = 8, DL_NEAR = 12, DL_PING = 14, DL_EARLY = 16, DL_VFAR = 18, DL_CTR2 = 20, DL_WORDS = 24 };

constexpr uint32_t MIG_INLINE = 255;   // leavers per side that ride in the first (fixed-size, 8 KB) migrant message

// slot of the first key >= target, for 4 targets (one thread each): the layer bounds of the owned range (used
// after the rare pass over all particles that takes in far arrivals)
__global__ void k_slab_bounds(const uint32_t* __restrict__ keys, uint32_t n, uint32_t layer, uint32_t zl, uint32_t G,
                              uint32_t* __restrict__ out, volatile uint32_t* __restrict__ out_host) {
    const uint32_t t = threadIdx.x;
    if (t >= 4) return;
    const uint32_t targets[4] = {G * layer, (G + 1u) * layer, (zl - G - 1u) * layer, (zl - G) * layer};      // G ghost layers per side
    uint32_t v = targets[t], lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    out[t] = lo;
    out_host[t] = lo;
}

// ONE kernel after the sort: every block finds the six layer bounds of the sorted owned keys (six threads, binary
// searches over L2-resident lines: cheaper than a kernel boundary), then the blocks pack the leavers -- they sit at
// the two ends of the sorted owned range, [0, lb0) go down, [lb3, n) go up -- and count the FAR ones: a leaver whose
// true cell layer is not the neighbour's adjacent layer (it crossed more than one layer in a step; its key is clamped
// into my ghost layer, so only its position can tell).  The block that finishes last writes record 0 of both
// messages, the header {#leavers, #particles that stay in my boundary layer on that side, #far leavers, 0}, and the
// device words the deep-interior density launch reads its slot range from.
// first slot in [0, n) with keys[slot] >= v, by ONE WAVE: 64 probes per round at equal spacing, 4 rounds for 16.7 M keys
// (a lane-per-target binary search is 24 DEPENDENT loads, ~0.8 us each from a cold L2: this kernel sits between the
// sort and the migrant exchange, on every rank's critical path)
__device__ __forceinline__ uint32_t wave_lower_bound(const uint32_t* __restrict__ keys, uint32_t n, uint32_t v) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0, hi = n;                              // invariant: keys[lo - 1] < v <= keys[hi] (virtual ends)
    while (hi - lo > 1u) {
        const uint32_t len = hi - lo, step = (len + 63u) / 64u;
        const uint32_t p = lo + min((lane + 1u) * step, len) - 1u;          // probes lo+step-1, lo+2 step-1, ..., hi-1
        const bool less = keys[p] < v;
        const uint64_t m = __ballot(less);                                   // monotone: a prefix of the lanes
        const uint32_t k = (uint32_t)__popcll(m);                            // probes 0 .. k-1 are < v
        const uint32_t new_lo = lo + min(k * step, len);
        const uint32_t new_hi = k >= 64u ? hi : lo + min((k + 1u) * step, len) - 1u;
        lo = new_lo; hi = max(new_hi, new_lo);
        if (k >= 64u) break;
    }
    if (hi - lo == 1u) lo += keys[lo] < v ? 1u : 0u;
    return lo;
}

// Round 6: the blocks from `pack_blocks` on build the cell table of the owned slots [own_off, own_off + n) (the sort leaves that
// to this kernel in a slab step: sph_ctx::owned_cells_in_bounds) -- the build and the bounds / leavers work read the same sorted
// keys and nobody needs either before the other, so they share one dispatch on every rank's critical path (~5 us).
__global__ __launch_bounds__(256) void k_slab_bounds_pack(const uint32_t* __restrict__ keys, const float4* __restrict__ posi,
                                                          const float4* __restrict__ velr, uint32_t n, uint32_t own_off,
                                                          uint32_t layer, uint32_t cap, GridDesc g, uint32_t early_cap,
                                                          uint32_t* __restrict__ dl, float4* __restrict__ out_lo,
                                                          float4* __restrict__ out_hi, uint32_t pack_blocks,
                                                          const uint32_t* __restrict__ keys_abs, uint2* __restrict__ cells,
                                                          volatile uint32_t* __restrict__ ends_host, uint32_t G, uint32_t two_layers,
                                                          uint32_t cap2) {
    if (blockIdx.x >= pack_blocks) {                   // (block-uniform; no barrier on this path)
        cells_build_thread(keys_abs, own_off, own_off + n, cells, ends_host, 0u, (blockIdx.x - pack_blocks) * 256u + threadIdx.x);
        return;
    }
    __shared__ uint32_t s_lb[12];
    __shared__ uint32_t s_last;
    const uint32_t zl = g.zl;
    {
        // deep interior = local layers [4, zl-4): owned layers at least THREE layers away from either cut (empty for
        // slabs of fewer than 7 owned layers).  Two would do for the density itself (arrivals land in the boundary
        // layer, the ghosts beyond it); the third lets the boundary layers' force launch -- which, rounded to whole
        // 64-slot chunks, reaches into the second layer and so reads densities of the third -- run on the comm stream
        // without waiting for the deep launch (the host checks that it really stays inside: targets 6 and 7)
        // (G = ghost layers per side: the owned layers are the local layers [G, zl - G); "layer k" in these comments counts from
        // the first owned layer = 1, as with one ghost layer)
        const uint32_t d0 = min(G + 3u, zl - 1u), d1 = zl >= 2u * G + 6u ? zl - G - 3u : d0;
        const uint32_t e0 = min(G + 2u, zl - 1u), e1 = zl >= 2u * G + 4u ? zl - G - 2u : e0;
        // [8..10]: the EARLY force range, local layers [6, zl-6) -- particles whose 27 cells lie in layers 5 .. zl-6, all of
        // whose densities the deep launch (layers >= 4, from a chunk boundary inside layer 4) has written: their force pass
        // needs nothing that comes over a link or moves before it.  [10] = first slot of layer 5, to check exactly that.
        const uint32_t f0 = min(G + 5u, zl - 1u), f1 = zl >= 2u * G + 11u ? zl - G - 5u : f0;
        const uint32_t targets[11] = {G * layer, (G + 1u) * layer, (zl - G - 1u) * layer, (zl - G) * layer, d0 * layer, max(d1, d0) * layer,
                                      e0 * layer, max(e1, e0) * layer, f0 * layer, max(f1, f0) * layer, min(G + 4u, zl - 1u) * layer};
        const uint32_t wave = threadIdx.x >> 6;
        for (uint32_t t = wave; t < 11u; t += 4u) {                         // wave w: targets w, w + 4, w + 8
            const uint32_t r = wave_lower_bound(keys, n, targets[t]);
            if ((threadIdx.x & 63u) == 0u) s_lb[t] = r;
        }
    }
    __syncthreads();
    const uint32_t lb0 = s_lb[0], lb3 = s_lb[3];
    const uint32_t m_lo = lb0, m_hi = n - lb3;
    uint32_t far_l = 0, far_h = 0;                     // per thread (a thread packs several leavers when there are many)
    uint32_t vfar = 0;                                 // leavers of either side beyond the neighbour's SECOND layer
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < cap && (k < m_lo || k < m_hi); k += pack_blocks * 256u) {
        if (k < m_lo) {
            const float4 p = posi[k];
            out_lo[2 + 2 * k] = p; out_lo[3 + 2 * k] = velr[k];
            const int lz = (int)cell_coord(p.z, g.box_min[2], g.box_dims[2], g.gf[2], g.g[2]);
            far_l += lz != g.z_off + (int)G - 1 ? 1u : 0u;    // not global layer z_lo - 1
            vfar += lz < g.z_off + (int)G - 2 ? 1u : 0u;
        }
        if (k < m_hi) {
            const float4 p = posi[lb3 + k];
            out_hi[2 + 2 * k] = p; out_hi[3 + 2 * k] = velr[lb3 + k];
            const int lz = (int)cell_coord(p.z, g.box_min[2], g.box_dims[2], g.gf[2], g.g[2]);
            far_h += lz != g.z_off + (int)zl - (int)G ? 1u : 0u;   // not global layer z_hi
            vfar += lz > g.z_off + (int)zl - (int)G + 1 ? 1u : 0u;
        }
    }
    if (two_layers) {
        // The ONE-MESSAGE step: behind the leavers of a side go the RESIDENTS of the two owned layers next to that cut, in slot
        // order -- [first layer | second layer] towards the lower neighbour, [second-last | last] towards the upper one: for the
        // receiver both are two of ITS ghost layers in ascending key order.  Slices that have been final since the sort: the
        // receiver merges its own leavers in (k_slab_unpack_ghosts_merge), so nothing waits for this rank's arrivals.
        const uint32_t r_lo0 = lb0, r_lo1 = max(s_lb[6], lb0), r_hi0 = min(s_lb[7], lb3), r_hi1 = lb3;
        const uint32_t n_rl = min(r_lo1 - r_lo0, cap2), n_rh = min(r_hi1 - r_hi0, cap2);
        float4* const dl_ = out_lo + 2u * (1u + min(m_lo, cap));
        float4* const dh_ = out_hi + 2u * (1u + min(m_hi, cap));
        for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n_rl + n_rh; k += pack_blocks * 256u) {
            if (k < n_rl) { dl_[2 * k] = posi[r_lo0 + k]; dl_[2 * k + 1] = velr[r_lo0 + k]; }
            else { const uint32_t j = k - n_rl; dh_[2 * j] = posi[r_hi0 + j]; dh_[2 * j + 1] = velr[r_hi0 + j]; }
        }
    }
    // far counts and the "last block" ticket: RETURNING device-scope atomics (performed at the memory side; the
    // returned value is awaited, so an add has been performed before its wave passes the barrier below) -- no fence
    uint32_t seen = 0;
    if (__ballot(far_l | far_h) != 0ull) {            // rare: somebody crossed more than one layer
        if (far_l) seen += atomicAdd(&dl[DL_CTR + 0], far_l);
        if (far_h) seen += atomicAdd(&dl[DL_CTR + 1], far_h);
        if (vfar) seen += atomicAdd(&dl[DL_CTR2], vfar);
    }
    asm volatile("" :: "v"(seen));                                        // keep the returns (and their waits)
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&dl[DL_CTR + 2], 1u) == pack_blocks - 1u ? 1u : 0u;
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    const uint32_t far_lo = atomicExch(&dl[DL_CTR + 0], 0u), far_hi = atomicExch(&dl[DL_CTR + 1], 0u);
    const uint32_t vfar_all = atomicExch(&dl[DL_CTR2], 0u);                            // (either side: any of them fails the one-message step)
    atomicExch(&dl[DL_CTR + 2], 0u);                                                   // re-armed for the next step
    // header, 8 words: {#leavers, #my boundary layer on that side, #far leavers, abort} {#my SECOND layer on that side, #very far, 0, 0}
    const uint32_t n2_lo = max(s_lb[6], s_lb[1]) - s_lb[1], n2_hi = s_lb[2] - min(s_lb[7], s_lb[2]);
    out_lo[0] = make_float4(__uint_as_float(m_lo), __uint_as_float(s_lb[1] - lb0), __uint_as_float(far_lo), 0.f);
    out_lo[1] = make_float4(__uint_as_float(n2_lo), __uint_as_float(vfar_all), 0.f, 0.f);
    out_hi[0] = make_float4(__uint_as_float(m_hi), __uint_as_float(lb3 - s_lb[2]), __uint_as_float(far_hi), 0.f);
    out_hi[1] = make_float4(__uint_as_float(n2_hi), __uint_as_float(vfar_all), 0.f, 0.f);
    dl[DL_VFAR] = vfar_all; dl[DL_VFAR + 1] = vfar_all;
    for (int t = 0; t < 4; t++) dl[t] = s_lb[t];
    // the deep range starts on a 64-slot chunk of what will be the owned range once the lower leavers are gone (the usual
    // step: no arrivals): the launch over "everything that is not deep" cuts its hole on whole waves from there
    // (targets_with_hole rounds the hole's start UP), so the two launches -- on two streams -- cover disjoint slots
    // instead of both writing the up to 63 slots in front of the first whole chunk (the same values: benign, but unordered)
    const uint32_t deep0 = lb0 + ((s_lb[4] - lb0 + 63u) & ~63u);
    dl[DL_DEEP] = own_off + deep0;
    dl[DL_DEEP + 1] = own_off + max(s_lb[5], deep0);
    dl[DL_FAR] = far_lo; dl[DL_FAR + 1] = far_hi;
    dl[DL_NEAR] = own_off + s_lb[6]; dl[DL_NEAR + 1] = own_off + max(s_lb[7], s_lb[6]);
    {   // empty unless every slot of layer 5 (and so of every layer up to zl-6) lies inside the deep range
        // and no longer than early_cap slots: the launch is there to fill the links' latency, not to run beside the interior
        // launch for its whole length (two big k_force grids side by side evict each other's L2 working sets)
        // (+ 64: on a step whose arrivals are merged in place the owned range's start shifts by a count that is no multiple of
        // 64, and the launch over "everything that is not deep" then re-rounds the hole's start -- it recomputes up to 63
        // slots at the head of the deep range on ANOTHER stream.  Same bits in fp32, but in mixed precision a density depends on
        // which particles share a wave: no slot the early launch reads -- layer 5 onwards -- may lie in that head.)
        const bool ok = zl >= 2u * G + 11u && s_lb[10] >= deep0 + 64u && s_lb[9] <= max(s_lb[5], deep0) && s_lb[9] > s_lb[8];
        dl[DL_EARLY] = own_off + s_lb[8];
        dl[DL_EARLY + 1] = own_off + (ok ? min(s_lb[9], s_lb[8] + early_cap) : s_lb[8]);
    }
}

// The early force launch (slab_step_body) ran before the owned range had its final start, so it left the next step's cell
// keys in a scratch array by ABSOLUTE slot and marked no movers.  This pass, over whole 64-slot chunks of the final owned
// range, puts the keys where the sort reads them and writes the chunks' mover bits (as the fused epilogue of k_force does).
__global__ __launch_bounds__(256) void k_slab_early_finish(const uint32_t* __restrict__ key_abs, const uint32_t* __restrict__ keyS,
                                                           uint32_t lo, uint32_t hi, uint32_t slot0, uint32_t* __restrict__ keys_out,
                                                           uint64_t* __restrict__ mm_mask, uint32_t* __restrict__ mm_tile_cnt) {
    const uint32_t i = lo + blockIdx.x * 256u + threadIdx.x;             // (lo - slot0 and hi - lo are multiples of 64)
    if (i >= hi) return;
    const uint32_t key = key_abs[i];
    keys_out[i - slot0] = key;
    if (mm_mask) {
        const uint64_t m = __ballot(key != keyS[i]);
        if ((threadIdx.x & 63u) == 0u) {
            const uint32_t chunk = (i - slot0) >> 6;
            mm_mask[chunk] = m;
            if (m) atomicAdd(&mm_tile_cnt[chunk / MM_TILE_CHUNKS], (uint32_t)__popcll(m));
        }
    }
}

// A rank that failed tells its neighbours: the header of its NEXT migrant message says "abort" (word 3), nothing else
__global__ void k_slab_abort_headers(float4* __restrict__ out_lo, float4* __restrict__ out_hi) {
    if (threadIdx.x == 0) out_lo[0] = make_float4(0.f, 0.f, 0.f, __uint_as_float(1u));
    if (threadIdx.x == 1) out_hi[0] = make_float4(0.f, 0.f, 0.f, __uint_as_float(1u));
}

// comm stream, right behind the migrant exchange: everything the host's one wait needs, in one mapped block, the
// sequence word last (a system-scope fence in between: the host polls that word and then reads the rest)
__global__ void k_slab_post_headers(const uint32_t* __restrict__ dl, const float4* __restrict__ hdr_lo,
                                    const float4* __restrict__ hdr_hi, volatile uint32_t* __restrict__ host, uint32_t seq) {
    const uint32_t t = threadIdx.x;                       // one wave: a lane per word, all stores in flight together
    if (t < 8u) host[t] = dl[t];
    else if (t == 16u || t == 17u) host[HL_NEAR + (t - 16u)] = dl[DL_NEAR + (t - 16u)];
    else if (t == 18u || t == 19u) host[HL_EARLY + (t - 18u)] = dl[DL_EARLY + (t - 18u)];
    else if (t == 20u || t == 21u) host[HL_VFAR + (t - 20u)] = dl[DL_VFAR + (t - 20u)];
    else if (t >= 32u && t < 40u) {                           // words 4..7 of the neighbours' headers
        const float4* h = t < 36u ? hdr_lo : hdr_hi;
        const uint32_t* hw = reinterpret_cast<const uint32_t*>(h);
        host[t] = h ? hw[4u + ((t - 32u) & 3u)] : 0u;         // {#its second layer, #very far leavers, 0, 0}
    }
    else if (t < 16u) {
        const float4* h = t < 12u ? hdr_lo : hdr_hi;
        const uint32_t w = (t - 8u) & 3u;
        const uint32_t* hw = reinterpret_cast<const uint32_t*>(h);
        host[t] = h ? hw[w] : 0u;                             // {#arrivals, #its boundary layer, #far, abort}
    }
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (t == 0u) host[HL_SEQ] = seq;
}

// both boundary layers -> their halo messages, one launch
__global__ __launch_bounds__(256) void k_slab_pack2(const float4* __restrict__ posi, const float4* __restrict__ velr,
                                                    uint32_t first_lo, uint32_t n_lo, uint32_t first_hi, uint32_t n_hi,
                                                    float4* __restrict__ rec_lo, float4* __restrict__ rec_hi) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_lo) { rec_lo[2 * t] = posi[first_lo + t]; rec_lo[2 * t + 1] = velr[first_lo + t]; }
    else if (t - n_lo < n_hi) {
        const uint32_t i = t - n_lo;
        rec_hi[2 * i] = posi[first_hi + i]; rec_hi[2 * i + 1] = velr[first_hi + i];
    }
}

// (density, pressure) of both boundary layers -> messages, and of both ghost layers <- messages: one launch each
__global__ __launch_bounds__(256) void k_slab_copy_dp2(const float2* __restrict__ src_lo, float2* __restrict__ dst_lo,
                                                       uint32_t n_lo, const float2* __restrict__ src_hi,
                                                       float2* __restrict__ dst_hi, uint32_t n_hi) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_lo) dst_lo[t] = src_lo[t];
    else if (t - n_lo < n_hi) dst_hi[t - n_lo] = src_hi[t - n_lo];
}

// ghost (rho, p) of both sides <- messages, with the neighbour terms the force pass reads of them (cw)
__global__ __launch_bounds__(256) void k_slab_unpack_dp2(const float2* __restrict__ src_lo, float2* __restrict__ dp_lo,
                                                         float2* __restrict__ cw_lo, uint32_t n_lo, const float2* __restrict__ src_hi,
                                                         float2* __restrict__ dp_hi, float2* __restrict__ cw_hi, uint32_t n_hi, Phys ph) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_lo) { const float2 v = src_lo[t]; dp_lo[t] = v; cw_lo[t] = neighbour_terms(ph, v.x, v.y); }
    else if (t - n_lo < n_hi) { const float2 v = src_hi[t - n_lo]; dp_hi[t - n_lo] = v; cw_hi[t - n_lo] = neighbour_terms(ph, v.x, v.y); }
}

// Ghost records of both sides -> the slots in front of / behind the owned range, with their cell keys, AND the cell
// table entries of the two ghost layers, in one launch: a thread recomputes the keys of its two neighbours from
// their records instead of reading them back (boundary flags as in k_cells_build; the records are in key order).
__global__ __launch_bounds__(256) void k_slab_unpack_ghosts(const float4* __restrict__ rec_lo, uint32_t n_lo, uint32_t slot_lo,
                                                            const float4* __restrict__ rec_hi, uint32_t n_hi, uint32_t slot_hi,
                                                            float4* __restrict__ posi, float4* __restrict__ velr,
                                                            uint32_t* __restrict__ key, uint2* __restrict__ cells, GridDesc g) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool low = t < n_lo;
    const uint32_t i = low ? t : t - n_lo, n = low ? n_lo : n_hi;
    if (i >= n) return;
    const float4* rec = low ? rec_lo : rec_hi;
    const uint32_t s = (low ? slot_lo : slot_hi) + i;
    const float4 p = rec[2 * i];
    const uint32_t k = cell_key(g, p.x, p.y, p.z);
    posi[s] = p;
    velr[s] = rec[2 * i + 1];
    key[s] = k;
    bool first = i == 0, last = i + 1 == n;
    if (!first) { const float4 q = rec[2 * (i - 1)]; first = cell_key(g, q.x, q.y, q.z) != k; }
    if (!last) { const float4 q = rec[2 * (i + 1)]; last = cell_key(g, q.x, q.y, q.z) != k; }
    if (first) cells[k].x = s;
    if (last) cells[k].y = s + 1;
}

// The one-message step (two ghost layers): the neighbour sent the RESIDENTS of its two layers next to the cut as they were right
// after its sort; what this rank itself sent INTO those layers this step -- its own leavers towards that side, still in its
// send buffer -- completes them.  Residents and leavers are both in ascending key order (this rank's numbering); equal keys as
// the neighbour will order them when it merges its arrivals (k_slab_insert / launch_merge_arrivals, i.e. the whole-domain
// stable sort): below this slab (side 0) the leavers arrive at the neighbour from ABOVE and go behind the residents of their
// cell, above it (side 1) they arrive from BELOW and go in front.  A thread per record: a resident counts the leavers in front
// of it, a leaver the residents (binary searches on keys recomputed from the records' positions; with no leavers on a side the
// count is 0 and this is a plain unpack).  The cell table of the ghost slots is built by k_cells_build2 afterwards.
__device__ __forceinline__ uint32_t rec_key(const float4* __restrict__ rec, uint32_t i, const GridDesc& g) {
    const float4 p = rec[2 * i];
    return cell_key(g, p.x, p.y, p.z);
}
__device__ __forceinline__ uint32_t rec_count_below(const float4* __restrict__ rec, uint32_t n, uint32_t bound, const GridDesc& g) {
    uint32_t lo = 0, hi = n;                                      // first record with key >= bound
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (rec_key(rec, mid, g) < bound) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_slab_unpack_ghosts_merge(const float4* __restrict__ res_lo, uint32_t nr_lo, const float4* __restrict__ lv_lo,
                                                                  uint32_t k_lo, uint32_t slot_lo, const float4* __restrict__ res_hi,
                                                                  uint32_t nr_hi, const float4* __restrict__ lv_hi, uint32_t k_hi,
                                                                  uint32_t slot_hi, float4* __restrict__ posi, float4* __restrict__ velr,
                                                                  uint32_t* __restrict__ key, GridDesc g) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const bool high = t >= nr_lo + k_lo;
    const uint32_t u = high ? t - (nr_lo + k_lo) : t;
    const uint32_t nr = high ? nr_hi : nr_lo, k = high ? k_hi : k_lo;
    if (u >= nr + k) return;
    const float4* res = high ? res_hi : res_lo;
    const float4* lv = high ? lv_hi : lv_lo;
    const uint32_t first = high ? 1u : 0u;                        // leavers first among equal keys (above this slab)
    const bool resident = u < nr;
    const uint32_t j = resident ? u : u - nr;
    const float4* rec = (resident ? res : lv) + 2 * (size_t)j;
    const float4 p = rec[0];
    const uint32_t kk = cell_key(g, p.x, p.y, p.z);
    const uint32_t ahead = resident ? rec_count_below(lv, k, kk + first, g) : rec_count_below(res, nr, kk + 1u - first, g);
    const uint32_t dst = (high ? slot_hi : slot_lo) + j + ahead;
    posi[dst] = p;
    velr[dst] = rec[1];
    key[dst] = kk;
}

// arrivals appended behind the owned range for the pass over all particles (launch_merge_arrivals): any owned layer
// is fine there, but a particle that is not inside this slab at all cannot be represented (its key is clamped into a
// ghost layer): flagged, the step reports SPH_E_STATE
__global__ __launch_bounds__(256) void k_slab_unpack(const float4* __restrict__ rec, uint32_t n, float4* __restrict__ posi,
                                                     float4* __restrict__ velr, uint32_t* __restrict__ key, GridDesc g, uint32_t G,
                                                     volatile uint32_t* __restrict__ err_host) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = rec[2 * i];
    posi[i] = p;
    velr[i] = rec[2 * i + 1];
    const uint32_t k = cell_key(g, p.x, p.y, p.z);
    if (key) key[i] = k;
    const uint32_t lz = k / (g.g[0] * g.g[1]);
    if (lz < G || lz + G >= g.zl) err_host[SLAB_ERR_ARRIVAL_OUTSIDE] = 1u;
}

// Arrivals join a BOUNDARY LAYER in place.  A particle a neighbour sent lies in the cell layer next to the cut it
// crossed (the sender counted the exceptions -- "far" leavers -- into its header, and a step with any of those does
// not come here), i.e. in the first or the last layer of the sorted owned range; the space in front of / behind the
// owned range is free (the ghosts of the last step are gone, those of this step have not come yet).  So only that
// layer is re-merged: the layer's nl particles and the k arrivals go, in key order, to the slots [d0, d0 + nl + k) of
// the scratch arrays, d0 = l0 - k on the low side and l0 on the high side, and are copied back; every other slot stays
// where it is.  EQUAL KEYS follow the order of the whole-domain sort, which is stable: a cell's particles are ordered
// by their slot BEFORE the sort, and a particle that came up from the slab below had a smaller slot than every
// resident of its new cell, one that came down from the slab above a larger one.  So on the low side the arrivals go
// IN FRONT of the residents of their cell (`arrivals_first`), on the high side behind them, arrivals among themselves
// in arrival order (= the sender's slot order) -- an N-slab run then holds every cell in exactly the order of the
// one-context run, and every sum over neighbours has the same bits (tests/test_gpu_slabs.py compare with array_equal).
// One thread per resident (counts the arrivals that go in front of it, from LDS) and per arrival (ranks itself among
// the arrivals, binary search among the residents).  An arrival whose key is NOT in the expected layer would break
// the order: it is flagged (the step then fails with SPH_E_STATE).
constexpr uint32_t SLAB_INSERT_MAX = 2048;
__global__ __launch_bounds__(256) void k_slab_insert(const float4* __restrict__ posi, const float4* __restrict__ velr,
                                                     const uint32_t* __restrict__ keyS, uint32_t l0, uint32_t nl,
                                                     const float4* __restrict__ rec, uint32_t k, GridDesc g,
                                                     float4* __restrict__ posi_o, float4* __restrict__ velr_o,
                                                     uint32_t* __restrict__ key_o, uint32_t d0, uint32_t want_layer,
                                                     bool arrivals_first, volatile uint32_t* __restrict__ err_host) {
    __shared__ uint32_t s_ak[SLAB_INSERT_MAX];
    const uint32_t layer = g.g[0] * g.g[1];
    for (uint32_t r = threadIdx.x; r < k; r += 256u) {
        const float4 p = rec[2 * r];
        const uint32_t key = cell_key(g, p.x, p.y, p.z);
        s_ak[r] = key;
        if (blockIdx.x == 0 && key / layer != want_layer) err_host[SLAB_ERR_INSERT_LAYER] = 1u;
    }
    __syncthreads();
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < nl) {
        const uint32_t key = keyS[l0 + t];
        uint32_t less = 0;                              // arrivals in front of this resident
        const uint32_t bound = key + (arrivals_first ? 1u : 0u);            // s_ak < key + 1  <=>  s_ak <= key
        for (uint32_t r = 0; r < k; r++) less += s_ak[r] < bound ? 1u : 0u;
        const uint32_t dst = d0 + t + less;
        posi_o[dst] = posi[l0 + t];
        velr_o[dst] = velr[l0 + t];
        key_o[dst] = key;
    } else if (t - nl < k) {
        const uint32_t r = t - nl, key = s_ak[r];
        uint32_t among = 0;
        for (uint32_t q = 0; q < k; q++) among += (s_ak[q] < key || (s_ak[q] == key && q < r)) ? 1u : 0u;
        uint32_t lo = 0, hi = nl;                       // residents in front of me: key < mine (low side) / <= mine (high side)
        const uint32_t bound = key + (arrivals_first ? 0u : 1u);
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (keyS[l0 + mid] < bound) lo = mid + 1; else hi = mid;
        }
        const uint32_t dst = d0 + among + lo;
        posi_o[dst] = rec[2 * r];
        velr_o[dst] = rec[2 * r + 1];
        key_o[dst] = key;
    }
}

// the merged boundary layer back from the scratch arrays (one launch instead of three copies)
__global__ __launch_bounds__(256) void k_slab_copy_back(const float4* __restrict__ p_src, const float4* __restrict__ v_src,
                                                        const uint32_t* __restrict__ k_src, float4* __restrict__ p_dst,
                                                        float4* __restrict__ v_dst, uint32_t* __restrict__ k_dst, uint32_t first,
                                                        uint32_t count) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= count) return;
    p_dst[first + t] = p_src[first + t];
    v_dst[first + t] = v_src[first + t];
    k_dst[first + t] = k_src[first + t];
}

// ---- re-cut (sph_slab_recut): whole layers change owner on the device --------------------------------------------------
// Where a particle of the sorted owned range goes under the NEW cuts: 0 = to the lower neighbour (its true cell layer is
// below new_z_lo), 2 = to the upper one, 1 = stays.  By POSITION, not by key: a leaver's key is clamped into a ghost layer.
constexpr uint32_t RECUT_BLOCK = 1024;          // slots per block of the three-way partition
__device__ __forceinline__ uint32_t recut_dest(const float4& p, const GridDesc& g, uint32_t new_lo, uint32_t new_hi) {
    const uint32_t lz = cell_coord(p.z, g.box_min[2], g.box_dims[2], g.inv_dims[2], g.gf[2], g.g[2]);
    return lz < new_lo ? 0u : (lz >= new_hi ? 2u : 1u);
}
__global__ __launch_bounds__(256) void k_recut_count(const float4* __restrict__ posi, uint32_t n, GridDesc g, uint32_t new_lo,
                                                     uint32_t new_hi, uint32_t* __restrict__ blk) {
    __shared__ uint32_t s_c[2];
    if (threadIdx.x < 2) s_c[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t c0 = 0, c2 = 0;
    for (uint32_t k = threadIdx.x; k < RECUT_BLOCK; k += 256u) {
        const uint32_t i = blockIdx.x * RECUT_BLOCK + k;
        if (i < n) { const uint32_t d = recut_dest(posi[i], g, new_lo, new_hi); c0 += d == 0u; c2 += d == 2u; }
    }
    if (c0) atomicAdd(&s_c[0], c0);
    if (c2) atomicAdd(&s_c[1], c2);
    __syncthreads();
    if (threadIdx.x < 2) blk[2 * blockIdx.x + threadIdx.x] = s_c[threadIdx.x];
}
// exclusive scan of the per-block {down, up} counts by one block; the totals go to mapped host memory
__global__ __launch_bounds__(1024) void k_recut_scan(uint32_t* __restrict__ blk, uint32_t nblk, volatile uint32_t* __restrict__ host) {
    __shared__ uint32_t part[2][1024];
    const uint32_t per = (nblk + 1023u) / 1024u, lo = min(threadIdx.x * per, nblk), hi = min(lo + per, nblk);
    uint32_t s0 = 0, s2 = 0;
    for (uint32_t b = lo; b < hi; b++) { s0 += blk[2 * b]; s2 += blk[2 * b + 1]; }
    part[0][threadIdx.x] = s0; part[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t a = threadIdx.x >= (uint32_t)o ? part[0][threadIdx.x - o] : 0u, b = threadIdx.x >= (uint32_t)o ? part[1][threadIdx.x - o] : 0u;
        __syncthreads();
        part[0][threadIdx.x] += a; part[1][threadIdx.x] += b;
        __syncthreads();
    }
    uint32_t r0 = part[0][threadIdx.x] - s0, r2 = part[1][threadIdx.x] - s2;
    for (uint32_t b = lo; b < hi; b++) { const uint32_t c0 = blk[2 * b], c2 = blk[2 * b + 1]; blk[2 * b] = r0; blk[2 * b + 1] = r2; r0 += c0; r2 += c2; }
    if (threadIdx.x == 1023) { host[0] = part[0][1023]; host[1] = part[1][1023]; }
}
// stable three-way partition: who stays -> [keep0 + rank), who goes down -> [down0 + rank), up -> [up0 + rank) of the
// ping-pong arrays, each class in slot order (one wave per 64 slots: ranks by ballot, the waves of a block in sequence)
__global__ __launch_bounds__(64) void k_recut_scatter(const float4* __restrict__ posi, const float4* __restrict__ velr, uint32_t n,
                                                      GridDesc g, uint32_t new_lo, uint32_t new_hi, const uint32_t* __restrict__ blk,
                                                      float4* __restrict__ posi_o, float4* __restrict__ velr_o, uint32_t keep0,
                                                      uint32_t down0, uint32_t up0) {
    const uint32_t lane = threadIdx.x;
    uint32_t r0 = blk[2 * blockIdx.x], r2 = blk[2 * blockIdx.x + 1];
    uint32_t r1 = blockIdx.x * RECUT_BLOCK - r0 - r2;                      // slots in front of this block that stay
    for (uint32_t k = 0; k < RECUT_BLOCK; k += 64u) {
        const uint32_t i = blockIdx.x * RECUT_BLOCK + k + lane;
        const bool live = i < n;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), v = p;
        uint32_t d = 3u;
        if (live) { p = posi[i]; v = velr[i]; d = recut_dest(p, g, new_lo, new_hi); }
        const uint64_t m0 = __ballot(d == 0u), m1 = __ballot(d == 1u), m2 = __ballot(d == 2u), below = (1ull << lane) - 1ull;
        if (live) {
            const uint32_t dst = d == 0u ? down0 + r0 + (uint32_t)__popcll(m0 & below)
                               : d == 1u ? keep0 + r1 + (uint32_t)__popcll(m1 & below) : up0 + r2 + (uint32_t)__popcll(m2 & below);
            posi_o[dst] = p; velr_o[dst] = v;
        }
        r0 += (uint32_t)__popcll(m0); r1 += (uint32_t)__popcll(m1); r2 += (uint32_t)__popcll(m2);
    }
}
// one chunk of both leaving lists -> 8-float records ; arrived records -> their final slots ; the kept run -> its final place
__global__ __launch_bounds__(256) void k_recut_pack(const float4* __restrict__ p, const float4* __restrict__ v, uint32_t first_lo,
                                                    uint32_t n_lo, uint32_t first_hi, uint32_t n_hi, float4* __restrict__ rec_lo,
                                                    float4* __restrict__ rec_hi) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_lo) { rec_lo[2 * t] = p[first_lo + t]; rec_lo[2 * t + 1] = v[first_lo + t]; }
    else if (t - n_lo < n_hi) { const uint32_t i = t - n_lo; rec_hi[2 * i] = p[first_hi + i]; rec_hi[2 * i + 1] = v[first_hi + i]; }
}
__global__ __launch_bounds__(256) void k_recut_unpack(const float4* __restrict__ rec_lo, uint32_t n_lo, uint32_t slot_lo,
                                                      const float4* __restrict__ rec_hi, uint32_t n_hi, uint32_t slot_hi,
                                                      float4* __restrict__ p, float4* __restrict__ v) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_lo) { p[slot_lo + t] = rec_lo[2 * t]; v[slot_lo + t] = rec_lo[2 * t + 1]; }
    else if (t - n_lo < n_hi) { const uint32_t i = t - n_lo; p[slot_hi + i] = rec_hi[2 * i]; v[slot_hi + i] = rec_hi[2 * i + 1]; }
}
__global__ __launch_bounds__(256) void k_recut_copy(const float4* __restrict__ ps, const float4* __restrict__ vs, uint32_t src,
                                                    float4* __restrict__ pd, float4* __restrict__ vd, uint32_t dst, uint32_t count) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < count) { pd[dst + t] = ps[src + t]; vd[dst + t] = vs[src + t]; }
}
__global__ void k_recut_counts_out(float4* __restrict__ out_lo, float4* __restrict__ out_hi, uint32_t down, uint32_t up) {
    if (threadIdx.x == 0) out_lo[0] = make_float4(__uint_as_float(down), 0.f, 0.f, 0.f);
    if (threadIdx.x == 1) out_hi[0] = make_float4(__uint_as_float(up), 0.f, 0.f, 0.f);
}

// ---- loop transport (sph_loop_transport_create): ONE slab whose two neighbours are its own periodic images ------------
// what leaves through the top comes back in at the bottom, shifted down by the slab's height, and the other way round:
// records carry a position float4 at every even float4 index (z = component 2); the (rho, p) message is a plain copy
__global__ __launch_bounds__(256) void k_loop_copy(const float4* __restrict__ up_src, float4* __restrict__ lo_dst, uint32_t n_up,
                                                   const float4* __restrict__ down_src, float4* __restrict__ hi_dst, uint32_t n_down,
                                                   uint32_t first_record4, float shift, int shift_z) {
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n_up + n_down; t += gridDim.x * 256u) {
        const bool up = t < n_up;                            // sent upwards: arrives from below, one slab height lower
        const uint32_t i = up ? t : t - n_up;
        float4 v = up ? up_src[i] : down_src[i];
        if (shift_z && i >= first_record4 && ((i - first_record4) & 1u) == 0u) v.z += up ? -shift : shift;
        (up ? lo_dst : hi_dst)[i] = v;
    }
}
// what a message of that size would spend on a link: a one-wave spin on the 100 MHz wall clock
__global__ void k_loop_delay(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// ---- neighbour ping (sph_slab_ping): a message whose every word says who sent it, towards which side, in which round --
__device__ __forceinline__ uint32_t ping_word(uint32_t rank, uint32_t side, uint32_t rep, uint32_t i) {
    uint32_t x = (rank * 2u + side) * 0x9E3779B9u + rep * 0x85EBCA6Bu + i;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
    return x;
}
__global__ __launch_bounds__(256) void k_slab_ping_fill(uint32_t* __restrict__ lo, uint32_t* __restrict__ hi, uint32_t words,
                                                        uint32_t rank, uint32_t rep) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) {
        if (lo) lo[i] = ping_word(rank, 0u, rep, i);
        if (hi) hi[i] = ping_word(rank, 1u, rep, i);
    }
}
// what came from below was sent UP by rank - 1 (its side 1), what came from above was sent DOWN by rank + 1 (its side 0)
__global__ __launch_bounds__(256) void k_slab_ping_check(const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi,
                                                         uint32_t words, uint32_t rank, uint32_t rep, uint32_t* __restrict__ bad) {
    uint32_t wrong = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < words; i += gridDim.x * 256u) {
        if (lo) wrong += lo[i] != ping_word(rank - 1u, 1u, rep, i) ? 1u : 0u;
        if (hi) wrong += hi[i] != ping_word(rank + 1u, 0u, rep, i) ? 1u : 0u;
    }
    if (wrong) atomicAdd(bad, wrong);
}

}  // namespace sph

using namespace sph;

// ---- transports ------------------------------------------------------------------------------------------------
namespace {

// RCCL through dlopen: libsph_hip.so does not depend on librccl for single-GPU users, and a process that already
// carries a copy (PyTorch bundles one) keeps using that one.
struct Id128 { char bytes[128]; };          // ncclUniqueId
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128 /* by value */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;       // optional
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;            // optional: what the communicator itself says it is
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*CommCuDevice)(void*, int*) = nullptr;
    int (*CommGetAsyncError)(void*, int*) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return SPH_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD); if (h) break; }     // a copy already in the process
    for (const char* nm : names) { if (h) break; h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); }
    SPH_REQUIRE(h, SPH_E_DEVICE, "librccl not found (%s)", dlerror());
#define SPH_SYM(field, name)                                                           \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                          \
    SPH_REQUIRE(g_rccl.field, SPH_E_DEVICE, "librccl lacks %s", name)
    SPH_SYM(GetUniqueId, "ncclGetUniqueId");
    SPH_SYM(CommInitRank, "ncclCommInitRank");
    SPH_SYM(CommDestroy, "ncclCommDestroy");
    *(void**)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
    SPH_SYM(Send, "ncclSend");
    SPH_SYM(Recv, "ncclRecv");
    SPH_SYM(GroupStart, "ncclGroupStart");
    SPH_SYM(GroupEnd, "ncclGroupEnd");
    SPH_SYM(GetErrorString, "ncclGetErrorString");
#undef SPH_SYM
    *(void**)(&g_rccl.CommCount) = dlsym(h, "ncclCommCount");
    *(void**)(&g_rccl.CommUserRank) = dlsym(h, "ncclCommUserRank");
    *(void**)(&g_rccl.CommCuDevice) = dlsym(h, "ncclCommCuDevice");
    *(void**)(&g_rccl.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
    g_rccl.lib = h;
    return SPH_OK;
}

struct RcclLink {
    void* comm = nullptr;
    int rank = 0, world = 1;
    bool aborted = false;
};

// A receive whose sender stopped never completes: it sits on the comm stream and blocks every later synchronisation
// of that stream (sph_slab_destroy, sph_slab_sync).  ncclCommAbort takes the communicator's queued operations down.
void rccl_abort(void* self) {
    RcclLink* L = (RcclLink*)self;
    if (!L || L->aborted || !L->comm) return;
    L->aborted = true;
    if (g_rccl.CommAbort) g_rccl.CommAbort(L->comm);
    L->comm = nullptr;
}

#define SPH_NCCL(call)                                                                                   \
    do {                                                                                                 \
        int r__ = (call);                                                                                \
        if (r__ != 0) { set_error("RCCL error %d (%s): %s", r__, g_rccl.GetErrorString(r__), #call); return SPH_E_DEVICE; } \
    } while (0)

// both neighbours in one group: no ordering between the four transfers, no deadlock
int rccl_exchange(void* self, int, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
                  const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes, void* stream) {
    RcclLink* L = (RcclLink*)self;
    SPH_REQUIRE(L->comm && !L->aborted, SPH_E_DEVICE, "the RCCL communicator of rank %d was aborted", L->rank);
    hipStream_t s = (hipStream_t)stream;
    const int ncclChar = 0;
    SPH_NCCL(g_rccl.GroupStart());
    // a call that fails inside the group must not leave the group open (the thread's next RCCL call -- ncclCommAbort among
    // them -- would be queued into it): remember the first error, close the group, then report
    int err = 0; const char* what = "";
    auto op = [&](int r, const char* name) { if (r != 0 && err == 0) { err = r; what = name; } };
    if (L->rank > 0) {
        if (send_lo_bytes) op(g_rccl.Send(send_lo, send_lo_bytes, ncclChar, L->rank - 1, L->comm, s), "ncclSend to rank - 1");
        if (recv_lo_bytes) op(g_rccl.Recv(recv_lo, recv_lo_bytes, ncclChar, L->rank - 1, L->comm, s), "ncclRecv from rank - 1");
    }
    if (L->rank + 1 < L->world) {
        if (send_hi_bytes) op(g_rccl.Send(send_hi, send_hi_bytes, ncclChar, L->rank + 1, L->comm, s), "ncclSend to rank + 1");
        if (recv_hi_bytes) op(g_rccl.Recv(recv_hi, recv_hi_bytes, ncclChar, L->rank + 1, L->comm, s), "ncclRecv from rank + 1");
    }
    op(g_rccl.GroupEnd(), "ncclGroupEnd");
    if (err) { set_error("RCCL error %d (%s): %s (rank %d of %d)", err, g_rccl.GetErrorString(err), what, L->rank, L->world); return SPH_E_DEVICE; }
    return SPH_OK;
}

// ---- device-to-device transport between the slabs of ONE process (several ranks sharing one GPU, one thread each) ----
// What RCCL does between GPUs, restated for ranks that share a device: the buffers are DEVICE pointers, the transfers
// are queued on the caller's comm stream and nothing waits for them on the host -- so the step's stream/event edges
// (after_main / after_comm, the ghost unpack on the comm stream, the (rho, p) copies) are exercised exactly as with
// the product transport, which a host-staged test transport (it drains the comm stream inside every exchange) cannot
// do.  Per directed link: the sender records an event behind the kernels that filled its buffer and posts {pointer,
// size, tag}; the receiver makes its stream wait for that event, queues the copy, records a `done` event and marks the
// post consumed; the sender then makes its own stream wait for `done` (its buffer is free again once the copy ran).
// The host threads only rendezvous (mutex + condition variable, bounded waits); sizes and tags of both ends are
// compared, so a disagreement is an error message instead of a hang.
struct LocalLink {
    uint64_t posted = 0, consumed = 0;
    const void* ptr = nullptr;
    size_t bytes = 0;
    int tag = 0;
    hipEvent_t ready = nullptr, done = nullptr;
};

}  // namespace

struct sph_local_hub {
    int world = 0;
    double timeout_s = 120.0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<LocalLink> links;        // [2 r] = r -> r + 1, [2 r + 1] = r -> r - 1
};

namespace {

struct LocalEnd { sph_local_hub* hub; int rank; };

int local_exchange(void* self, int tag, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
                   const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes, void* stream) {
    LocalEnd* E = (LocalEnd*)self;
    sph_local_hub* H = E->hub;
    hipStream_t st = (hipStream_t)stream;
    const int r = E->rank;
    const bool lo = r > 0, hi = r + 1 < H->world;
    LocalLink* out[2] = {lo ? &H->links[2 * r + 1] : nullptr, hi ? &H->links[2 * r] : nullptr};             // r -> r-1, r -> r+1
    LocalLink* in[2] = {lo ? &H->links[2 * (r - 1)] : nullptr, hi ? &H->links[2 * (r + 1) + 1] : nullptr};   // r-1 -> r, r+1 -> r
    const void* sp[2] = {send_lo, send_hi};
    const size_t sb[2] = {lo ? send_lo_bytes : 0, hi ? send_hi_bytes : 0};
    void* rp[2] = {recv_lo, recv_hi};
    const size_t rb[2] = {lo ? recv_lo_bytes : 0, hi ? recv_hi_bytes : 0};
    const auto limit = std::chrono::duration<double>(H->timeout_s);
    for (int k = 0; k < 2; k++) {                       // post what I send
        if (!sb[k]) continue;
        SPH_HIP(hipEventRecord(out[k]->ready, st));
        std::lock_guard<std::mutex> g(H->mu);
        out[k]->ptr = sp[k]; out[k]->bytes = sb[k]; out[k]->tag = tag;
        out[k]->posted++;
        H->cv.notify_all();
    }
    for (int k = 0; k < 2; k++) {                       // take what the neighbours posted
        if (!rb[k]) continue;
        LocalLink* L = in[k];
        const void* src; size_t bytes; int ptag;
        {
            std::unique_lock<std::mutex> g(H->mu);
            if (!H->cv.wait_for(g, limit, [&] { return L->posted > L->consumed; })) {
                set_error("local transport: rank %d waited %.0f s for a %zu-byte message (tag %d) its neighbour never sent", r,
                          H->timeout_s, rb[k], tag);
                return SPH_E_DEVICE;
            }
            src = L->ptr; bytes = L->bytes; ptag = L->tag;
        }
        if (bytes != rb[k] || ptag != tag) {
            set_error("local transport: rank %d expects %zu bytes (tag %d) from its %s neighbour, which sent %zu (tag %d): the two "
                      "ends of a link disagree on a message size", r, rb[k], tag, k == 0 ? "lower" : "upper", bytes, ptag);
            return SPH_E_STATE;
        }
        SPH_HIP(hipStreamWaitEvent(st, L->ready, 0));
        SPH_HIP(hipMemcpyAsync(rp[k], src, bytes, hipMemcpyDeviceToDevice, st));
        SPH_HIP(hipEventRecord(L->done, st));
        std::lock_guard<std::mutex> g(H->mu);
        L->consumed++;
        H->cv.notify_all();
    }
    for (int k = 0; k < 2; k++) {                       // my buffers are free once the neighbours' copies ran
        if (!sb[k]) continue;
        LocalLink* L = out[k];
        {
            std::unique_lock<std::mutex> g(H->mu);
            if (!H->cv.wait_for(g, limit, [&] { return L->consumed == L->posted; })) {
                set_error("local transport: rank %d waited %.0f s for its neighbour to take a %zu-byte message (tag %d)", r,
                          H->timeout_s, sb[k], tag);
                return SPH_E_DEVICE;
            }
        }
        SPH_HIP(hipStreamWaitEvent(st, L->done, 0));
    }
    return SPH_OK;
}

}  // namespace

namespace {
struct LoopEnd { float shift; double gbs, latency_us; };

int loop_exchange(void* self, int tag, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
                  const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes, void* stream) {
    LoopEnd* E = (LoopEnd*)self;
    hipStream_t st = (hipStream_t)stream;
    SPH_REQUIRE(send_hi_bytes == recv_lo_bytes && send_lo_bytes == recv_hi_bytes, SPH_E_STATE,
                "loop transport (tag %d): what goes up (%zu bytes) is not what is expected from below (%zu), or down %zu / from above %zu: "
                "the two ends of a link disagree on a message size", tag, send_hi_bytes, recv_lo_bytes, send_lo_bytes, recv_hi_bytes);
    if (E->latency_us > 0.0 || E->gbs > 0.0) {           // both links carry their message at the same time
        const size_t big = send_hi_bytes > send_lo_bytes ? send_hi_bytes : send_lo_bytes;
        const double us = E->latency_us + (E->gbs > 0.0 ? (double)big / (E->gbs * 1e3) : 0.0);
        hipLaunchKernelGGL(k_loop_delay, dim3(1), dim3(1), 0, st, (unsigned long long)(us * 100.0));
    }
    const bool recs = tag == SPH_TAG_MIGRANTS || tag == SPH_TAG_MIGRANTS_REST || tag == SPH_TAG_HALO_A || tag == SPH_TAG_ONE ||
                      tag == SPH_TAG_ONE_REST;
    if (!recs) {                                          // (rho, p) pairs, pings: bytes as they are
        if (send_hi_bytes) SPH_HIP(hipMemcpyAsync(recv_lo, send_hi, send_hi_bytes, hipMemcpyDeviceToDevice, st));
        if (send_lo_bytes) SPH_HIP(hipMemcpyAsync(recv_hi, send_lo, send_lo_bytes, hipMemcpyDeviceToDevice, st));
        return SPH_OK;
    }
    SPH_REQUIRE((send_hi_bytes | send_lo_bytes) % 32 == 0, SPH_E_INVALID, "loop transport: a particle message is made of 32-byte records");
    const uint32_t n_up = (uint32_t)(send_hi_bytes / 16), n_down = (uint32_t)(send_lo_bytes / 16);
    if (n_up + n_down) {
        hipLaunchKernelGGL(k_loop_copy, dim3(min(ceil_div(n_up + n_down, 256u), 2048u)), dim3(256), 0, st, (const float4*)send_hi,
                           (float4*)recv_lo, n_up, (const float4*)send_lo, (float4*)recv_hi, n_down, (tag == SPH_TAG_MIGRANTS || tag == SPH_TAG_ONE) ? 2u : 0u,
                           E->shift, recs ? 1 : 0);
        SPH_HIP(hipGetLastError());
    }
    return SPH_OK;
}
}  // namespace

struct sph_slab {
    sph_ctx* c = nullptr;
    int rank = 0, world = 1;
    bool has_lo = false, has_hi = false;
    sph_transport tr{};
    bool host_staged = false;            // the transport wants host buffers (tests); else device pointers on the comm stream
    hipStream_t comm = nullptr;
    hipEvent_t ev_main = nullptr, ev_comm = nullptr, ev_deep = nullptr;
    // The two cross-stream edges every step takes several times -- "comm goes on behind main" (after_main) and the reverse
    // (after_comm) -- as a sequence number one stream WRITES into a word of device memory and the other WAITS for
    // (hipStreamWriteValue32 / hipStreamWaitValue32) instead of an event: a chain of small kernels alternating between two
    // streams runs at 7.4-8.6 us per kernel and hop that way against 14.4-15.9 with an event per hop (profiles/hop_chain.hip,
    // profiles/r06_cross_stream_hops.txt).  Used where the device supports it and a write / wait pair at create time worked
    // (SPH_SLAB_HOPS=event in the environment: events, for A/B runs); the rarer edges keep their events.
    uint32_t* hop_mem = nullptr;         // one word per edge, a cache line apart: HOP_* below
    uint32_t hop_seq[5] = {0u, 0u, 0u, 0u, 0u};
    bool hops_by_value = false;
    hipStream_t early = nullptr;         // the early force launch's own stream (behind the deep density by event): its tail then runs
    hipEvent_t ev_early_go = nullptr, ev_early_done = nullptr;   // beside the interior launch instead of in front of it
    bool early_own_stream = true;        // SPH_SLAB_EARLY_STREAM=0: on the main stream (A/B)
    uint32_t gcap = 0, mcap = 0;         // halo / migrant capacity per side, in records
    uint32_t* d_lb = nullptr;            // DL_* words: layer bounds, deep-interior range, far counts, block counters (device)
    volatile uint32_t* h_lb = nullptr;   // HL_* words (pinned, mapped): what the step's one wait reads
    uint32_t* h_lb_dev = nullptr;        // its device view
    uint32_t seq = 0;                    // step number the device echoes into h_lb[HL_SEQ]
    double wait_timeout_s = 120.0;       // bound of the poll (sph_slab_set_wait_timeout; SPH_SLAB_TIMEOUT_S)
    int device = 0;                      // copied from the context: destroy does not touch it
    float4* mig_send[2] = {nullptr, nullptr};   // (1 + mcap) records of 2 float4
    float4* mig_recv[2] = {nullptr, nullptr};
    float4* halo_send[2] = {nullptr, nullptr};  // gcap records
    float4* halo_recv[2] = {nullptr, nullptr};
    float2* dens_send[2] = {nullptr, nullptr};
    float2* dens_recv[2] = {nullptr, nullptr};
    char* stage_send[2] = {nullptr, nullptr};   // pinned staging for host-staged transports
    char* stage_recv[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    uint64_t steps = 0, migrants = 0, resorts = 0, ghosts = 0, host_waits = 0, inserts = 0, far_steps = 0, rest_msgs = 0;
    uint64_t exchanges = 0;              // transport calls so far (3 in a usual step: migrants, halo A, halo B; 1 in the one-message step)
    // The step's protocol (sph_slab_set_protocol): 3 = MIGRANTS / HALO A / HALO B, three dependent message groups; 1 = ONE group --
    // header, leavers and the residents of the two layers next to each cut in one message per neighbour, its size fixed in
    // advance from the counts both ends saw in the PREVIOUS step's headers (one_prev_*; the first step after a create / re-cut
    // runs the three-group protocol to learn them), ghost densities recomputed here.  Needs two ghost layers in the context.
    int protocol = 3;
    bool one_ready = false;              // one_prev_* are those of the step before this one
    uint32_t one_prev_s[2] = {0, 0};     // records (leavers + two layers) I packed towards each side in the last step
    uint32_t one_prev_r[2] = {0, 0};     // ... and what each neighbour packed towards me
    uint32_t msg_cap = 0;                // records a migrant / one-message buffer holds behind its header
    uint64_t one_steps = 0, one_rest_msgs = 0;
    bool early_force = true;             // the force pass of the innermost layers runs in front of the step's wait, on a stream of its own
                                         // (sph_slab_set_early_force: ~3 us per step when the links are fast, -30 at 40 us per group)
    uint32_t early_cap = 1u << 20;       // slots of that launch at most (~150 us of k_force): what a link's latency needs, no more
    uint64_t early_launches = 0, early_used = 0;
    uint32_t early_span = 0;             // slots the last step's early range held: sizes this step's grid (what a grid misses, the interior launch computes)
    bool early_span_known = false;       // false: no step has reported a range yet (the first launch is sized for the whole slab)
    uint32_t* recut_blk = nullptr;       // sph_slab_recut: {down, up} counts per 1024-slot block, then their scan
    uint64_t recuts = 0, recut_moved = 0;
    // failure: the first error of this slab (sticky), its message, and whether the transport may still be used
    int failed = 0;
    char fail_msg[512] = {0};
    bool transport_dead = false;
    // what the step in flight has exchanged so far and what it still owes its neighbours (slab_fail)
    // ---- where a step's time goes (sph_slab_timing_get).  Host side, always on (three clock reads per step): the one
    //      wait, the host time in front of it (hash / sort / bounds / migrant exchange queued) and the whole call.
    //      Device side, only while sph_slab_timing_enable(1): an event pair around every transport call on the comm
    //      stream (an event recorded on a stream costs the device ~5 us at its next dispatch: not in timed runs).
    struct Acc { uint64_t n = 0; double sum = 0.0, max = 0.0; void add(double v) { n++; sum += v; if (v > max) max = v; } };
    Acc t_wait, t_pre, t_post, t_host;
    uint64_t waits_ready = 0;            // waits whose sequence word was there at the first look: the HOST is behind the device
    Acc t_group[9];                      // by tag - 1: MIGRANTS, HALO_A, HALO_B, MIGRANTS_REST, PING, (re-cut x2,) ONE, ONE_REST
    bool time_groups = false;
    struct Pending { int tag; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> ev_free;
    struct Progress {
        bool mig_posted = false;                     // this step's first exchange (MIGRANTS, or ONE) has been handed to the transport
        bool one = false;                            // the one-message protocol: that exchange carries one_s / one_r records behind the header
        uint32_t one_s[2] = {0, 0}, one_r[2] = {0, 0};
        bool next_known = false;                     // this step's headers are in: the NEXT step's one-message sizes follow from them
        uint32_t next_s[2] = {0, 0}, next_r[2] = {0, 0};
        bool headers = false, rest = false, halo_a = false, halo_b = false;
        bool peer_dead[2] = {false, false};          // that neighbour's header said "abort": nothing more to or from it
        uint32_t rest_s[2] = {0, 0}, rest_r[2] = {0, 0};   // records of the second migrant message (send, receive) per side
        uint32_t h[2] = {0, 0}, g[2] = {0, 0};       // boundary-layer records I send / ghost records I receive per side
    } pg;
};

namespace {

void slab_free(sph_slab* s) {
    if (!s) return;
    for (int k = 0; k < 2; k++) {
        hipFree(s->mig_send[k]); hipFree(s->mig_recv[k]); hipFree(s->halo_send[k]); hipFree(s->halo_recv[k]);
        hipFree(s->dens_send[k]); hipFree(s->dens_recv[k]);
        if (s->stage_send[k]) hipHostFree(s->stage_send[k]);
        if (s->stage_recv[k]) hipHostFree(s->stage_recv[k]);
    }
    hipFree(s->d_lb);
    hipFree(s->recut_blk);
    if (s->h_lb) hipHostFree((void*)s->h_lb);
    if (s->hop_mem) hipFree(s->hop_mem);
    if (s->ev_main) hipEventDestroy(s->ev_main);
    if (s->ev_comm) hipEventDestroy(s->ev_comm);
    if (s->ev_deep) hipEventDestroy(s->ev_deep);
    if (s->ev_early_go) hipEventDestroy(s->ev_early_go);
    if (s->ev_early_done) hipEventDestroy(s->ev_early_done);
    if (s->early) hipStreamDestroy(s->early);
    for (auto& pd : s->pending) { hipEventDestroy(pd.a); hipEventDestroy(pd.b); }
    for (hipEvent_t e : s->ev_free) hipEventDestroy(e);
    if (s->comm) hipStreamDestroy(s->comm);
    delete s;
}

// hand the four buffers to the transport.  Device transports get device pointers and the comm stream; host-staged
// ones get pinned host copies (the comm stream is drained first: a test transport, not the product path).
hipEvent_t slab_timing_event(sph_slab* s) {
    hipEvent_t e = nullptr;
    if (!s->ev_free.empty()) { e = s->ev_free.back(); s->ev_free.pop_back(); return e; }
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

int slab_exchange_raw(sph_slab* s, int tag, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
                      const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes);

// the transport call of one message group; with sph_slab_timing_enable an event pair brackets it on the comm stream: the
// time between the two is the group as the DEVICE sees it -- waiting for the neighbour's half included, which is the point
int slab_exchange(sph_slab* s, int tag, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
                  const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes) {
    hipEvent_t a = nullptr, b = nullptr;
    if (s->time_groups && tag >= 1 && tag <= 9 && tag != SPH_TAG_RECUT_COUNTS && tag != SPH_TAG_RECUT) {
        a = slab_timing_event(s); b = slab_timing_event(s);
        if (a && b) SPH_HIP(hipEventRecord(a, s->comm));
    }
    const int rc = slab_exchange_raw(s, tag, send_lo, send_lo_bytes, recv_lo, recv_lo_bytes, send_hi, send_hi_bytes, recv_hi,
                                     recv_hi_bytes);
    if (a && b) {
        if (rc == SPH_OK && hipEventRecord(b, s->comm) == hipSuccess) s->pending.push_back({tag, a, b});
        else { s->ev_free.push_back(a); s->ev_free.push_back(b); }
    }
    return rc;
}

int slab_exchange_raw(sph_slab* s, int tag, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
                      const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes) {
    if (!s->has_lo || s->pg.peer_dead[0]) send_lo_bytes = recv_lo_bytes = 0;
    if (!s->has_hi || s->pg.peer_dead[1]) send_hi_bytes = recv_hi_bytes = 0;
    s->exchanges++;
    if (!s->host_staged) {
        int rc = s->tr.exchange(s->tr.self, tag, send_lo, send_lo_bytes, recv_lo, recv_lo_bytes, send_hi, send_hi_bytes, recv_hi,
                                recv_hi_bytes, (void*)s->comm);
        if (rc < 0 && sph_last_error()[0] == 0) set_error("slab transport failed (tag %d)", tag);
        if (rc < 0) s->transport_dead = true;
        return rc;
    }
    SPH_REQUIRE(send_lo_bytes <= s->stage_bytes && send_hi_bytes <= s->stage_bytes && recv_lo_bytes <= s->stage_bytes &&
                    recv_hi_bytes <= s->stage_bytes, SPH_E_CAPACITY, "slab message exceeds the staging buffers");
    if (send_lo_bytes) SPH_HIP(hipMemcpyAsync(s->stage_send[0], send_lo, send_lo_bytes, hipMemcpyDeviceToHost, s->comm));
    if (send_hi_bytes) SPH_HIP(hipMemcpyAsync(s->stage_send[1], send_hi, send_hi_bytes, hipMemcpyDeviceToHost, s->comm));
    SPH_HIP(hipStreamSynchronize(s->comm));
    int rc = s->tr.exchange(s->tr.self, tag, s->stage_send[0], send_lo_bytes, s->stage_recv[0], recv_lo_bytes, s->stage_send[1],
                            send_hi_bytes, s->stage_recv[1], recv_hi_bytes, nullptr);
    if (rc < 0) { if (sph_last_error()[0] == 0) set_error("slab transport failed (tag %d)", tag); s->transport_dead = true; return rc; }
    if (recv_lo_bytes) SPH_HIP(hipMemcpyAsync(recv_lo, s->stage_recv[0], recv_lo_bytes, hipMemcpyHostToDevice, s->comm));
    if (recv_hi_bytes) SPH_HIP(hipMemcpyAsync(recv_hi, s->stage_recv[1], recv_hi_bytes, hipMemcpyHostToDevice, s->comm));
    return SPH_OK;
}

// comm stream continues after everything queued on main so far / main after comm
// One edge "stream `to` goes on behind what `from` holds now", in two halves so that the wait can be queued later than the mark:
// hop_mark returns the edge's sequence number (the write is ALWAYS queued before any wait that needs it: no wait without its
// write), hop_wait queues the wait for it.  Events where write / wait value is not available (sph_slab::hops_by_value).
enum { HOP_MAIN_COMM = 0, HOP_COMM_MAIN = 1, HOP_MAIN_EARLY = 2, HOP_EARLY_MAIN = 3, HOP_DEEP = 4 };
int hop_mark(sph_slab* s, int edge, hipStream_t from, hipEvent_t ev, uint32_t* value) {
    if (s->hops_by_value) {
        *value = ++s->hop_seq[edge];
        SPH_HIP(hipStreamWriteValue32(from, s->hop_mem + 16 * edge, *value, 0));
        return SPH_OK;
    }
    *value = 0u;
    SPH_HIP(hipEventRecord(ev, from));
    return SPH_OK;
}
int hop_wait(sph_slab* s, int edge, hipStream_t to, hipEvent_t ev, uint32_t value) {
    if (s->hops_by_value) {
        SPH_HIP(hipStreamWaitValue32(to, s->hop_mem + 16 * edge, value, hipStreamWaitValueGte, 0xFFFFFFFFu));
        return SPH_OK;
    }
    SPH_HIP(hipStreamWaitEvent(to, ev, 0));
    return SPH_OK;
}
int after_main(sph_slab* s) {
    uint32_t v;
    int rc = hop_mark(s, HOP_MAIN_COMM, s->c->stream, s->ev_main, &v);
    return rc ? rc : hop_wait(s, HOP_MAIN_COMM, s->comm, s->ev_main, v);
}
int after_comm(sph_slab* s) {
    uint32_t v;
    int rc = hop_mark(s, HOP_COMM_MAIN, s->comm, s->ev_comm, &v);
    return rc ? rc : hop_wait(s, HOP_COMM_MAIN, s->c->stream, s->ev_comm, v);
}

// launchers use the context's stream: this runs them on the comm stream instead
struct OnComm {
    sph_ctx* c; hipStream_t saved;
    OnComm(sph_slab* s) : c(s->c), saved(s->c->stream) { c->stream = s->comm; }
    ~OnComm() { c->stream = saved; }
};

// the step's one wait: poll the sequence word the comm stream writes behind the migrant exchange.  Polled, not slept
// on (hipEventSynchronize hands the thread to the kernel and comes back tens of microseconds late); BOUNDED: a
// neighbour that left its step with an error never sends, and this rank must report that instead of spinning forever.
int slab_wait_headers(sph_slab* s) {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    if (s->h_lb[HL_SEQ] == s->seq) s->waits_ready++;          // the device got here first: this step is paced by the host
    while (s->h_lb[HL_SEQ] != s->seq) {
        __builtin_ia32_pause();
        if ((++spins & 0x3FFFu) == 0u) {
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (dt > s->wait_timeout_s) {
                set_error("rank %d: no migrant header after %.1f s (step %llu): a neighbour stopped, or the transport is stuck",
                          s->rank, dt, (unsigned long long)s->steps);
                s->transport_dead = true;
                return SPH_E_DEVICE;
            }
            hipError_t q = hipStreamQuery(s->comm);            // a device fault shows up here, not in the mapped word
            if (q != hipSuccess && q != hipErrorNotReady) SPH_HIP(q);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    s->host_waits++;
    s->t_wait.add(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    return SPH_OK;
}

int slab_check_device_flags(sph_slab* s) {
    SPH_REQUIRE(s->h_lb[HL_ERR + SLAB_ERR_INSERT_LAYER] == 0u, SPH_E_STATE,
                "rank %d: an arriving particle was not in the boundary layer next to the cut it crossed (the sorted order of "
                "that layer is invalid)", s->rank);
    SPH_REQUIRE(s->h_lb[HL_ERR + SLAB_ERR_ARRIVAL_OUTSIDE] == 0u, SPH_E_STATE,
                "rank %d: an arriving particle lies outside this slab altogether (it crossed a whole slab in one step)", s->rank);
    return SPH_OK;
}

// elapsed times of the bracketed transport calls so far (drains the comm stream)
int slab_timing_collect(sph_slab* s) {
    if (s->pending.empty()) return SPH_OK;
    SPH_HIP(hipStreamSynchronize(s->comm));
    for (auto& pd : s->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pd.a, pd.b) == hipSuccess) s->t_group[pd.tag - 1].add((double)ms * 1e3);
        s->ev_free.push_back(pd.a); s->ev_free.push_back(pd.b);
    }
    s->pending.clear();
    return SPH_OK;
}

int slab_step_body(sph_slab* s, float dt) {
    sph_ctx* c = s->c;
    int rc;
    const auto t_begin = std::chrono::steady_clock::now();
    s->pg = sph_slab::Progress();
    const size_t rec = 2 * sizeof(float4);
    // The ONE-MESSAGE step (sph_slab_set_protocol(s, 1); two ghost layers): header, leavers and the residents of the two layers
    // next to each cut travel in ONE message per neighbour, before the host knows any count -- so its size is fixed by a rule on
    // the counts both ends saw in the PREVIOUS step's headers (a margin of 1/16 + 1024 records; what does not fit follows in an
    // exact second message after the wait: the first step of a burst).  The first step after a create / re-cut has no such
    // counts and runs the three-group protocol.
    const bool p1 = s->protocol == 1 && s->one_ready && s->world > 1;
    auto one_cap = [](uint32_t prev) { return (prev + prev / 16u + 1024u + 63u) & ~63u; };
    const uint32_t S_s[2] = {p1 && s->has_lo ? one_cap(s->one_prev_s[0]) : 0u, p1 && s->has_hi ? one_cap(s->one_prev_s[1]) : 0u};
    const uint32_t S_r[2] = {p1 && s->has_lo ? one_cap(s->one_prev_r[0]) : 0u, p1 && s->has_hi ? one_cap(s->one_prev_r[1]) : 0u};
    if (p1) {
        SPH_REQUIRE(S_s[0] <= s->msg_cap && S_s[1] <= s->msg_cap && S_r[0] <= s->msg_cap && S_r[1] <= s->msg_cap, SPH_E_CAPACITY,
                    "rank %d: the one-message step would carry %u/%u (send) %u/%u (receive) records; the buffers hold %u", s->rank, S_s[0], S_s[1],
                    S_r[0], S_r[1], s->msg_cap);
        s->pg.one = true;
        for (int k = 0; k < 2; k++) { s->pg.one_s[k] = S_s[k]; s->pg.one_r[k] = S_r[k]; }
    }
    rc = slab_check_device_flags(s); if (rc) return rc;
    // ---- hash + sort the owned particles (leavers end up at the two ends of the owned range) -----------------------
    c->defer_ghost_clear = true;                  // the old ghosts' cells are cleared by the sort's first kernel, not by one of their own
    rc = step_hash(c);
    c->defer_ghost_clear = false;
    if (rc) return rc;
    c->owned_cells_in_bounds = true;              // the sort leaves the table of the owned slots to the bounds kernel below
    c->owned_cells_pending = false;
    rc = step_sort(c);
    c->owned_cells_in_bounds = false;
    if (rc) return rc;
    const uint32_t layer = c->grid.g[0] * c->grid.g[1];
    const uint32_t n0 = c->n, off0 = c->own_off;
    // ---- layer bounds, leavers and headers in ONE kernel; the comm stream ships the fixed-size part ------------------
    s->seq++;
    // a few blocks: every block finds the bounds for itself, the leavers (few, at most mcap) are packed in a grid-stride loop;
    // behind them the blocks that build the cell table of the owned slots when the sort left it pending (not on a skipped sort)
    // a few blocks (many when they also copy two layers): every block finds the bounds for itself, the leavers and residents are packed in
    // grid-stride loops; behind them the blocks that build the cell table of the owned slots when the sort left it pending (not on a skipped sort)
    const uint32_t pack_blocks = p1 ? 128u : min(ceil_div(s->mcap, 256u), 16u), build_blocks = c->owned_cells_pending ? cells_build_blocks(n0) : 0u;
    c->owned_cells_pending = false;
    hipLaunchKernelGGL(k_slab_bounds_pack, dim3(pack_blocks + build_blocks), dim3(256), 0, c->stream, c->keyS + off0, c->posi + off0,
                       c->velr + off0, n0, off0, layer, s->mcap, c->grid, s->early_cap, s->d_lb, s->mig_send[0], s->mig_send[1], pack_blocks,
                       c->keyS, c->cells, c->mm_count_host_dev + 1, c->ghost_layers, p1 ? 1u : 0u, s->gcap);
    SPH_HIP(hipGetLastError());
    rc = after_main(s); if (rc) return rc;
    // ---- the density of the deep interior goes into the main stream's queue BEFORE the host waits: its slot range
    //      comes from device memory (k_slab_bounds_pack wrote it).  Layers >= 4 from either cut see neither ghosts nor
    //      arrivals (those land in the boundary layers), and no slot of them moves before the force pass.
    const uint32_t G = c->ghost_layers;           // ghost layers per side: the owned layers are the local layers [G, zl - G)
    bool deep_valid = c->grid.zl >= 2u * G + 7u;
    if (deep_valid) {
        PhaseTimer t(c, SPH_PH_DENS);
        rc = launch_density_dev_range(c, s->d_lb + DL_DEEP, n0);
        if (rc) return rc;
    }
    // ---- and behind it the fused force pass of the innermost layers (local layers [6, zl-6): every density they read is
    //      the deep launch's), also from a range in device memory.  More work that needs nothing from a link: the main
    //      stream stays busy while the migrant message, the host's wake-up and halo A are on their way.  The owned range
    //      does not have its final start yet (leavers go, arrivals may be merged in front), so the launch leaves its keys by
    //      ABSOLUTE slot in the sort's scratch keys and marks no movers; k_slab_early_finish does both once the start is
    //      known.  A step whose arrivals take the pass over all particles throws the result away (every slot moves).
    bool early_launched = false;
    // the grid: the range's size is on the device; the host sizes the launch from the LAST step's range plus a margin (a range
    // changes by a few slots a step) -- whatever a too small grid leaves out is computed by the interior launch below
    const uint32_t early_grid_slots = s->early_span_known ? min(n0, ((s->early_span + s->early_span / 64u + 1023u) & ~255u)) : n0;
    bool early_pending = false;          // the launch runs on its own stream: the main stream has not waited for it yet
    uint32_t early_done_v = 0u;          // (the sequence number of that edge: hop_mark)
    // "the last step's range was EMPTY" (layer 5 not inside the deep range, a sparse slab) is not "unknown": no launch then --
    // a full-size grid of blocks that leave at once, two events and a stream hop bought nothing, every step.  The bounds
    // kernel reports the range whether or not a launch used it, so the launch comes back one step after the range does.
    const bool early_empty = s->early_span_known && s->early_span == 0u;
    if (deep_valid && s->early_force && s->world > 1 && c->grid.zl >= 2u * G + 11u && !early_empty) {          // (no neighbour, no latency to fill)
        if (s->early_own_stream && s->early) {
            uint32_t go;                                                    // behind the deep density
            rc = hop_mark(s, HOP_MAIN_EARLY, c->stream, s->ev_early_go, &go); if (rc) return rc;
            rc = hop_wait(s, HOP_MAIN_EARLY, s->early, s->ev_early_go, go); if (rc) return rc;
            hipStream_t saved = c->stream;
            c->stream = s->early;
            { PhaseTimer t(c, SPH_PH_FORCE); rc = launch_force_dev_range(c, s->d_lb + DL_EARLY, early_grid_slots, dt); }
            c->stream = saved;
            if (rc) return rc;
            rc = hop_mark(s, HOP_EARLY_MAIN, s->early, s->ev_early_done, &early_done_v); if (rc) return rc;
            early_pending = true;
        } else {
            PhaseTimer t(c, SPH_PH_FORCE);
            rc = launch_force_dev_range(c, s->d_lb + DL_EARLY, early_grid_slots, dt);
            if (rc) return rc;
        }
        early_launched = true;
        s->early_launches++;
    }
    // whoever re-sorts or re-writes the ping-pong arrays on the main stream must come behind the early launch
    auto join_early = [&]() -> int {
        if (early_pending) { const int rj = hop_wait(s, HOP_EARLY_MAIN, c->stream, s->ev_early_done, early_done_v); if (rj) return rj; early_pending = false; }
        return SPH_OK;
    };
    struct JoinOnExit {                  // (an error return leaves the step half done: the next sort must still come behind the launch)
        sph_slab* s; bool* pending; uint32_t* value;
        ~JoinOnExit() { if (*pending) (void)hop_wait(s, HOP_EARLY_MAIN, s->c->stream, s->ev_early_done, *value); }
    } join_on_exit{s, &early_pending, &early_done_v};
    const uint32_t inl = min(MIG_INLINE, s->mcap);
    const size_t mig_bytes = (size_t)(1 + inl) * rec;
    s->pg.mig_posted = true;                                    // (also when the call fails: the transport is dead then)
    if (p1) rc = slab_exchange(s, SPH_TAG_ONE, s->mig_send[0], (1 + (size_t)S_s[0]) * rec, s->mig_recv[0], (1 + (size_t)S_r[0]) * rec,
                               s->mig_send[1], (1 + (size_t)S_s[1]) * rec, s->mig_recv[1], (1 + (size_t)S_r[1]) * rec);
    else rc = slab_exchange(s, SPH_TAG_MIGRANTS, s->mig_send[0], mig_bytes, s->mig_recv[0], mig_bytes, s->mig_send[1], mig_bytes,
                            s->mig_recv[1], mig_bytes);
    if (rc) return rc;
    hipLaunchKernelGGL(k_slab_post_headers, dim3(1), dim3(64), 0, s->comm, s->d_lb, s->has_lo ? s->mig_recv[0] : (float4*)nullptr,
                       s->has_hi ? s->mig_recv[1] : (float4*)nullptr, s->h_lb_dev, s->seq);
    SPH_HIP(hipGetLastError());
    // ---- the one host wait of the step ---------------------------------------------------------------------------
    s->t_pre.add(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
    rc = slab_wait_headers(s); if (rc) return rc;
    const auto t_waited = std::chrono::steady_clock::now();
    s->pg.headers = true;
    const uint32_t lb0 = s->h_lb[HL_LB], lb1 = s->h_lb[HL_LB + 1], lb2 = s->h_lb[HL_LB + 2], lb3 = s->h_lb[HL_LB + 3];
    const uint32_t deep_lo = s->h_lb[HL_DEEP], deep_hi = s->h_lb[HL_DEEP + 1];
    const uint32_t near_lo = s->h_lb[HL_NEAR], near_hi = s->h_lb[HL_NEAR + 1];     // first slot of layer 3 / of layer zl-3
    const uint32_t early_lo = s->h_lb[HL_EARLY];                                       // what the early force launch covered:
    const uint32_t early_hi = min(s->h_lb[HL_EARLY + 1], early_lo + early_grid_slots);      // its range, as far as its grid reached
    s->early_span = s->h_lb[HL_EARLY + 1] - early_lo;
    s->early_span_known = true;
    const uint32_t far_lo = s->has_lo ? s->h_lb[HL_FAR] : 0u, far_hi = s->has_hi ? s->h_lb[HL_FAR + 1] : 0u;
    const uint32_t m_lo = lb0, m_hi = n0 - lb3;
    uint32_t own_lo = lb1 - lb0, own_hi = lb3 - lb2;
    const uint32_t in_lo = s->has_lo ? s->h_lb[HL_HDR_LO] : 0u, peer_own_lo = s->has_lo ? s->h_lb[HL_HDR_LO + 1] : 0u;
    const uint32_t in_hi = s->has_hi ? s->h_lb[HL_HDR_HI] : 0u, peer_own_hi = s->has_hi ? s->h_lb[HL_HDR_HI + 1] : 0u;
    const uint32_t far_in_lo = s->has_lo ? s->h_lb[HL_HDR_LO + 2] : 0u, far_in_hi = s->has_hi ? s->h_lb[HL_HDR_HI + 2] : 0u;
    // the SECOND layers next to the cuts (the one-message step's ghosts-of-ghosts; every header carries the counts, so that the
    // step after a three-group step can size its message): mine from the bounds, the neighbours' from their headers
    // (the same expressions as the kernel's header words: k_slab_bounds_pack)
    const uint32_t n2_lo = s->has_lo ? max(near_lo - off0, lb1) - lb1 : 0u;
    const uint32_t n2_hi = s->has_hi ? lb2 - min(near_hi - off0, lb2) : 0u;
    const uint32_t n2p_lo = s->has_lo ? s->h_lb[HL_HDR2_LO] : 0u, n2p_hi = s->has_hi ? s->h_lb[HL_HDR2_HI] : 0u;
    const uint32_t vfar_mine = s->h_lb[HL_VFAR], vfar_peer = (s->has_lo ? s->h_lb[HL_HDR2_LO + 1] : 0u) + (s->has_hi ? s->h_lb[HL_HDR2_HI + 1] : 0u);
    // records either end of a link packed this step: what the NEXT one-message step is sized from
    const uint32_t tot_s[2] = {s->has_lo ? m_lo + own_lo + n2_lo : 0u, s->has_hi ? m_hi + own_hi + n2_hi : 0u};
    const uint32_t tot_r[2] = {s->has_lo ? in_lo + peer_own_lo + n2p_lo : 0u, s->has_hi ? in_hi + peer_own_hi + n2p_hi : 0u};
    {   // What this step still owes its neighbours -- the second migrant message, halo A, halo B -- in numbers BOTH ends of
        // a link see (mine in my header, the neighbour's in its header), clamped to the buffers: if this rank fails from
        // here on it still sends and takes exactly these (slab_fail), so that no neighbour is left waiting for a message.
        sph_slab::Progress& g = s->pg;
        auto umin = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
        g.peer_dead[0] = s->has_lo && s->h_lb[HL_HDR_LO + 3] != 0u;
        g.peer_dead[1] = s->has_hi && s->h_lb[HL_HDR_HI + 3] != 0u;
        const uint32_t inl0 = umin(MIG_INLINE, s->mcap);
        const uint32_t mm[2] = {umin(m_lo, s->mcap), umin(m_hi, s->mcap)}, ii[2] = {umin(in_lo, s->mcap), umin(in_hi, s->mcap)};
        const uint32_t own[2] = {own_lo, own_hi}, peer[2] = {peer_own_lo, peer_own_hi}, fo[2] = {umin(far_lo, m_lo), umin(far_hi, m_hi)},
                       fi[2] = {umin(far_in_lo, in_lo), umin(far_in_hi, in_hi)}, mraw[2] = {m_lo, m_hi}, iraw[2] = {in_lo, in_hi};
        const bool any_rest = m_lo > inl0 || m_hi > inl0 || in_lo > inl0 || in_hi > inl0;
        for (int k = 0; k < 2; k++) {
            const bool has = k == 0 ? s->has_lo : s->has_hi;
            if (p1) {                     // all that can still be owed is the part of the one message that did not fit its fixed size
                g.rest_s[k] = tot_s[k] > S_s[k] ? umin(tot_s[k], s->msg_cap) - S_s[k] : 0u;
                g.rest_r[k] = tot_r[k] > S_r[k] ? umin(tot_r[k], s->msg_cap) - S_r[k] : 0u;
                continue;
            }
            g.rest_s[k] = any_rest && mm[k] > inl0 ? mm[k] - inl0 : 0u;
            g.rest_r[k] = any_rest && ii[k] > inl0 ? ii[k] - inl0 : 0u;
            g.h[k] = has ? umin(own[k] + iraw[k] - fi[k], s->gcap) : 0u;
            g.g[k] = has ? umin(peer[k] + mraw[k] - fo[k], s->gcap) : 0u;
        }
        if (p1) g.halo_a = g.halo_b = true;       // (no such messages in this protocol)
        g.next_known = true;
        for (int k = 0; k < 2; k++) { g.next_s[k] = one_cap(tot_s[k]); g.next_r[k] = one_cap(tot_r[k]); }
        SPH_REQUIRE(!g.peer_dead[0] && !g.peer_dead[1], SPH_E_PEER, "rank %d: its %s neighbour reported a failure and stopped (step %llu)",
                    s->rank, g.peer_dead[0] ? (g.peer_dead[1] ? "lower and upper" : "lower") : "upper", (unsigned long long)s->steps);
    }
    SPH_REQUIRE(s->has_lo || m_lo == 0, SPH_E_STATE, "rank %d: %u particles below the lowest slab", s->rank, m_lo);
    SPH_REQUIRE(s->has_hi || m_hi == 0, SPH_E_STATE, "rank %d: %u particles above the highest slab", s->rank, m_hi);
    // both ends of a link see the same numbers (mine in my header, the neighbour's in its header): they fail together
    SPH_REQUIRE(m_lo <= s->mcap && m_hi <= s->mcap && in_lo <= s->mcap && in_hi <= s->mcap, SPH_E_CAPACITY,
                "rank %d: a burst of %u/%u leaving, %u/%u arriving particles exceeds the migrant capacity %u", s->rank, m_lo, m_hi,
                in_lo, in_hi, s->mcap);
    SPH_REQUIRE(far_in_lo <= in_lo && far_in_hi <= in_hi && far_lo <= m_lo && far_hi <= m_hi, SPH_E_STATE,
                "rank %d: inconsistent migrant headers", s->rank);
    // capacity for what arrives, checked BEFORE anything is dropped or merged: on this error the owned range is still the
    // sorted range of this step
    SPH_REQUIRE(n0 - m_lo - m_hi + in_lo + in_hi <= c->cap && (uint64_t)off0 + n0 - m_hi + in_lo + in_hi <= c->tot, SPH_E_CAPACITY,
                "rank %d: %u + %u arriving particles exceed the capacity %u", s->rank, n0 - m_lo - m_hi, in_lo + in_hi, c->cap);
    if (p1) {
        SPH_REQUIRE(vfar_mine == 0u && vfar_peer == 0u, SPH_E_STATE,
                    "rank %d: %u leaving / %u arriving particles crossed more than TWO cell layers in one step: the one-message step keeps two "
                    "ghost layers (the three-group protocol takes such particles as long as they land in an interior layer)", s->rank,
                    vfar_mine, vfar_peer);
        SPH_REQUIRE(tot_s[0] <= s->msg_cap && tot_s[1] <= s->msg_cap && tot_r[0] <= s->msg_cap && tot_r[1] <= s->msg_cap &&
                        peer_own_lo + n2p_lo <= s->gcap && peer_own_hi + n2p_hi <= s->gcap, SPH_E_CAPACITY,
                    "rank %d: two layers of %u+%u / %u+%u residents (+ %u / %u leavers) exceed the message buffers (%u records) or the ghost "
                    "capacity %u", s->rank, peer_own_lo, n2p_lo, peer_own_hi, n2p_hi, in_lo, in_hi, s->msg_cap, s->gcap);
        // what did not fit the size fixed in advance (a burst: many more leavers than in the step before), exact
        const sph_slab::Progress& g = s->pg;
        if (g.rest_s[0] | g.rest_s[1] | g.rest_r[0] | g.rest_r[1]) {
            rc = slab_exchange(s, SPH_TAG_ONE_REST, s->mig_send[0] + 2 * (1 + (size_t)S_s[0]), g.rest_s[0] * rec, s->mig_recv[0] + 2 * (1 + (size_t)S_r[0]),
                               g.rest_r[0] * rec, s->mig_send[1] + 2 * (1 + (size_t)S_s[1]), g.rest_s[1] * rec,
                               s->mig_recv[1] + 2 * (1 + (size_t)S_r[1]), g.rest_r[1] * rec);
            if (rc) return rc;
            s->one_rest_msgs++;
        }
        s->one_steps++;
    } else
    // ---- more leavers than ride in the fixed-size message: the rest, exact size (both ends know both counts) ----------
    if (m_lo > inl || m_hi > inl || in_lo > inl || in_hi > inl) {
        const size_t s_lo = m_lo > inl ? (size_t)(m_lo - inl) * rec : 0, s_hi = m_hi > inl ? (size_t)(m_hi - inl) * rec : 0;
        const size_t r_lo = in_lo > inl ? (size_t)(in_lo - inl) * rec : 0, r_hi = in_hi > inl ? (size_t)(in_hi - inl) * rec : 0;
        rc = slab_exchange(s, SPH_TAG_MIGRANTS_REST, s->mig_send[0] + 2 * (1 + inl), s_lo, s->mig_recv[0] + 2 * (1 + inl), r_lo,
                           s->mig_send[1] + 2 * (1 + inl), s_hi, s->mig_recv[1] + 2 * (1 + inl), r_hi);
        if (rc) return rc;
        s->rest_msgs++;
    }
    s->pg.rest = true;
    // A step without arrivals (the usual one) hands the rest of the halo work to the COMM stream at once: the main
    // stream is busy with the deep density, and pack -> HALO A -> ghost unpack need nothing from it (the slices they
    // read have been final since the sort).  The ghosts are then in place when the deep density ends, and everything
    // that is left of the density pass is ONE launch.
    const bool early_halo = deep_valid && in_lo == 0 && in_hi == 0;
    // ---- drop the leavers (their cells hold nothing else until the ghosts arrive; the clearing must precede the
    //      ghost cells, so it runs on the stream that builds those) ---------------------------------------------------
    if (m_lo || m_hi) {
        if (c->cells_valid && c->cells_lo == c->own_off && c->cells_hi == c->own_off + c->n) {
            if (early_halo) {
                OnComm on(s);
                rc = launch_cells_clear_2ranges(c, c->own_off, c->own_off + m_lo, c->own_off + c->n - m_hi, c->own_off + c->n);
            } else {
                rc = launch_cells_clear_2ranges(c, c->own_off, c->own_off + m_lo, c->own_off + c->n - m_hi, c->own_off + c->n);
            }
            if (rc) return rc;
            c->cells_lo += m_lo;
            c->cells_hi -= m_hi;
        }
        c->own_off += m_lo;
        c->n -= m_lo + m_hi;
        s->migrants += m_lo + m_hi;
    }
    // ---- arrivals become owned particles.  They land in the boundary layer next to the cut they crossed -- except the
    //      `far` ones the sender counted, which land deeper -- so the boundary counts are known without counting again ---
    if (in_lo || in_hi) {
        SPH_REQUIRE(c->n + in_lo + in_hi <= c->cap && c->own_off + c->n + in_lo + in_hi <= c->tot, SPH_E_CAPACITY,
                    "rank %d: %u + %u arriving particles exceed the capacity %u", s->rank, c->n, in_lo + in_hi, c->cap);
        rc = after_comm(s); if (rc) return rc;                  // the received records are in mig_recv
        const bool merge = c->sort_merge && c->order_valid && c->cells_valid && c->cells_lo == c->own_off &&
                           c->cells_hi == c->own_off + c->n;
        const bool in_place = merge && far_in_lo == 0 && far_in_hi == 0 && in_lo <= SLAB_INSERT_MAX &&
                              in_hi <= SLAB_INSERT_MAX && in_lo <= c->own_off && own_lo + own_hi <= c->n;
        if (in_place) {
            // only the two boundary layers are touched (see k_slab_insert): their cells leave the table, the merged
            // layers come back from the scratch arrays, their cells are built again
            for (int side = 0; side < 2; side++) {
                const uint32_t k = side == 0 ? in_lo : in_hi;
                if (!k) continue;
                const uint32_t nl = side == 0 ? own_lo : own_hi;
                const uint32_t l0 = side == 0 ? c->own_off : c->own_off + c->n - own_hi;
                const uint32_t d0 = side == 0 ? l0 - k : l0;
                rc = launch_cells_clear_range(c, l0, l0 + nl); if (rc) return rc;
                hipLaunchKernelGGL(k_slab_insert, dim3(ceil_div(nl + k, 256u)), dim3(256), 0, c->stream, c->posi, c->velr, c->keyS, l0,
                                   nl, s->mig_recv[side] + 2, k, c->grid, c->posi2, c->velr2, c->keyS2, d0,
                                   side == 0 ? G : c->grid.zl - G - 1u, side == 0, s->h_lb_dev + HL_ERR);
                hipLaunchKernelGGL(k_slab_copy_back, dim3(ceil_div(nl + k, 256u)), dim3(256), 0, c->stream, c->posi2, c->velr2, c->keyS2,
                                   c->posi, c->velr, c->keyS, d0, nl + k);
                SPH_HIP(hipGetLastError());
                if (side == 0) c->own_off = d0;
                c->n += k;
                c->cells_lo = c->own_off; c->cells_hi = c->own_off + c->n;
                rc = launch_cells_build_range(c, d0, d0 + nl + k); if (rc) return rc;
            }
            own_lo += in_lo;
            own_hi += in_hi;
            s->inserts++;
            s->resorts++;
        } else {
            // behind the sorted owned range, with their cell keys: the merge path takes them in as movers without an old
            // slot (one pass over the particles; a full radix sort when the merge path is switched off).  Every slot
            // moves: the density of the deep interior is computed again with the rest.
            // Equal keys must come out in the whole-domain order (see k_slab_insert): what came up from below in front
            // of the residents of its cell, what came down from above behind them.  The merge takes that as a rule on
            // the appended slots (launch_merge_arrivals: the first in_lo of them are `front` movers); the full radix
            // sort is stable, so there the lower neighbour's particles are put physically IN FRONT of the owned range.
            deep_valid = false;
            rc = join_early(); if (rc) return rc;     // every slot moves, through the arrays the early launch writes
            uint32_t appended = 0;
            const bool front_slots = !merge && in_lo <= c->own_off;
            for (int side = 0; side < 2; side++) {
                const uint32_t cnt = side == 0 ? in_lo : in_hi;
                if (!cnt) continue;
                if (side == 0 && front_slots) {
                    hipLaunchKernelGGL(k_slab_unpack, dim3(ceil_div(cnt, 256u)), dim3(256), 0, c->stream, s->mig_recv[0] + 2, cnt,
                                       c->posi + c->own_off - cnt, c->velr + c->own_off - cnt, (uint32_t*)nullptr, c->grid, G,
                                       s->h_lb_dev + HL_ERR);
                    continue;
                }
                const uint32_t at = c->own_off + c->n + appended;
                hipLaunchKernelGGL(k_slab_unpack, dim3(ceil_div(cnt, 256u)), dim3(256), 0, c->stream, s->mig_recv[side] + 2, cnt,
                                   c->posi + at, c->velr + at, merge ? c->k0 + c->n + appended : (uint32_t*)nullptr, c->grid, G,
                                   s->h_lb_dev + HL_ERR);
                appended += cnt;
            }
            SPH_HIP(hipGetLastError());
            if (merge) {
                rc = launch_merge_arrivals(c, appended, in_lo); if (rc) return rc;   // the first in_lo came up from below
            } else {
                if (front_slots) { c->own_off -= in_lo; c->n += in_lo; }
                c->n += appended;
                c->keys_fresh = false;
                c->order_valid = false;
                c->stage = sph_ctx::ST_LOADED;
                rc = step_hash(c); if (rc) return rc;
                rc = step_sort(c); if (rc) return rc;
            }
            own_lo += in_lo - far_in_lo;          // the far ones went past the boundary layer
            own_hi += in_hi - far_in_hi;
            s->resorts++;
            if (far_in_lo || far_in_hi) {
                // rare (a particle crossed more than one cell layer in a step): count the layers again.  The neighbours
                // size their ghost messages from the headers, so what the count must confirm is that every far arrival
                // stayed clear of the OTHER boundary layer and of the ghost layers.
                s->far_steps++;
                if (getenv("SPH_SLAB_DEBUG"))
                    fprintf(stderr, "[slab %d] step %llu: far arrivals %u/%u of %u/%u; leavers %u/%u (far %u/%u); bounds %u %u %u %u of %u; "
                            "peer boundary %u/%u\n", s->rank, (unsigned long long)s->steps, far_in_lo, far_in_hi, in_lo, in_hi, m_lo, m_hi,
                            far_lo, far_hi, lb0, lb1, lb2, lb3, n0, peer_own_lo, peer_own_hi);
                hipLaunchKernelGGL(k_slab_bounds, dim3(1), dim3(64), 0, c->stream, c->keyS + c->own_off, c->n, layer, c->grid.zl, G,
                                   s->d_lb, s->h_lb_dev);
                SPH_HIP(hipGetLastError());
                SPH_HIP(hipStreamSynchronize(c->stream));
                s->host_waits++;
                rc = slab_check_device_flags(s); if (rc) return rc;
                const uint32_t r0 = s->h_lb[0], r1 = s->h_lb[1], r2 = s->h_lb[2], r3 = s->h_lb[3];
                SPH_REQUIRE(r0 == 0u && r3 == c->n && r1 - r0 == own_lo && r3 - r2 == own_hi, SPH_E_STATE,
                            "rank %d: a particle that crossed several cell layers in one step landed in a boundary or ghost "
                            "layer (boundary layers %u/%u, expected %u/%u): the time step is too large for this slab width",
                            s->rank, r1 - r0, r3 - r2, own_lo, own_hi);
            }
        }
    }
    const uint32_t n = c->n;
    // ghosts = what stayed in the neighbour's boundary layer + what I just sent INTO that layer; the one-message step keeps
    // the neighbour's second layer as well (g1: the inner ghost layer, whose densities this rank computes itself)
    const uint32_t g1_lo = s->has_lo ? peer_own_lo + m_lo - far_lo : 0u, g1_hi = s->has_hi ? peer_own_hi + m_hi - far_hi : 0u;
    const uint32_t g_lo = p1 && s->has_lo ? peer_own_lo + n2p_lo + m_lo : g1_lo;
    const uint32_t g_hi = p1 && s->has_hi ? peer_own_hi + n2p_hi + m_hi : g1_hi;
    const uint32_t h_lo = s->has_lo ? own_lo : 0u, h_hi = s->has_hi ? own_hi : 0u;
    SPH_REQUIRE(own_lo <= n && own_hi <= n, SPH_E_STATE, "rank %d: inconsistent boundary counts", s->rank);
    SPH_REQUIRE(h_lo <= s->gcap && h_hi <= s->gcap && g_lo <= s->gcap && g_hi <= s->gcap && g_lo <= c->own_off &&
                    c->own_off + n + g_hi <= c->tot, SPH_E_CAPACITY,
                "rank %d: a boundary layer of %u/%u (ghosts %u/%u) exceeds the ghost capacity %u", s->rank, h_lo, h_hi, g_lo, g_hi,
                s->gcap);
    // interior = everything but the two boundary layers, in whole 64-slot chunks (the fused force pass marks the
    // movers of the next sort per chunk)
    uint32_t a = c->own_off + ((h_lo + 63u) & ~63u), b = h_hi ? c->own_off + ((n - h_hi) & ~63u) : c->own_off + n;
    if (b < a || a > c->own_off + n) { a = c->own_off; b = c->own_off; }     // a thin slab: everything is "boundary"
    // Early mode runs the boundary chunks' force launch on the comm stream with no event behind the deep density launch
    // -- allowed only while none of its targets can see a density that launch writes: [own_off, a) must end inside layer
    // 2 (it then reads layers <= 3; the deep interior starts at 4), [b, end) must start inside layer zl-3 (a side
    // without a boundary layer has no such piece at all).  Sparse layers can break that (64 slots may span several
    // layers): then the event is recorded now -- the main stream's queue holds nothing behind the deep launch yet.
    // (An unconditional record costs the device ~6 us of idle per step; a MISSING one cost 18 NaN particles in one of
    // three 2-rank rehearsals of round 3: the high-priority comm stream overtook a deep launch that queued behind
    // another rank's kernels.)
    const bool need_deep_event = early_halo && !(a <= near_lo && b >= near_hi);
    uint32_t deep_v = 0u;
    if (need_deep_event) { rc = hop_mark(s, HOP_DEEP, c->stream, s->ev_deep, &deep_v); if (rc) return rc; }
    // the slots the non-deep density launches cover: the owned range -- and, in the one-message step, the inner ghost layers
    // on either side (what HALO B carries in the three-group step), rounded down to a whole 64-slot chunk in front so that the
    // hole these launches leave for the deep range stays on the chunk boundaries the deep launch used; the few slots of the
    // OUTER ghost layer that catches get a density nobody reads (their own neighbourhood is not complete here).  ONE launch with
    // the owned slots: giving the ghost layers launches of their own behind the main stream's release was measured and is
    // slower -- two partly filled rounds of workgroups beside the interior force launch: +63 us of density kernel time, the
    // step +4 ... +9 us at every link setting (profiles/r06b_periodic_slab_protocols_ghost_density_split.txt).
    uint32_t dens_lo = c->own_off, dens_hi = c->own_off + n;
    if (p1) {
        const uint32_t back = (g1_lo + 63u) & ~63u;
        dens_lo = c->own_off - (back <= g_lo ? back : g1_lo);
        dens_hi = c->own_off + n + g1_hi;
    }
    // ---- halo A: boundary layers -> neighbours' ghost layers -----------------------------------------------------------
    hipStream_t pack_stream = early_halo ? s->comm : c->stream;
    if (!p1 && h_lo + h_hi)
        hipLaunchKernelGGL(k_slab_pack2, dim3(ceil_div(h_lo + h_hi, 256u)), dim3(256), 0, pack_stream, c->posi, c->velr, c->own_off, h_lo,
                           c->own_off + n - h_hi, h_hi, s->halo_send[0], s->halo_send[1]);
    SPH_HIP(hipGetLastError());
    if (!early_halo) {
        rc = after_main(s); if (rc) return rc;                  // the comm stream may start once the slices are packed
        // the interior density runs while the halo travels: queued BEFORE the transfers are handed to the transport.
        // What the deep launch already did is left out.
        PhaseTimer t(c, SPH_PH_DENS);
        rc = deep_valid ? launch_density_hole(c, a, b, deep_lo, deep_hi) : launch_density_range(c, a, b);
        if (rc) return rc;
    }
    if (!p1) {
        rc = slab_exchange(s, SPH_TAG_HALO_A, s->halo_send[0], h_lo * rec, s->halo_recv[0], g_lo * rec, s->halo_send[1], h_hi * rec,
                           s->halo_recv[1], g_hi * rec);
        if (rc) return rc;
        s->pg.halo_a = true;
        // ghosts go directly in front of / behind the owned range, already in key order; their cells join the table of
        // the owned slots (comm stream, one kernel: none of it is touched by the interior passes)
        if (g_lo + g_hi)
            hipLaunchKernelGGL(k_slab_unpack_ghosts, dim3(ceil_div(g_lo + g_hi, 256u)), dim3(256), 0, s->comm, s->halo_recv[0], g_lo,
                               c->own_off - g_lo, s->halo_recv[1], g_hi, c->own_off + n, c->posi, c->velr, c->keyS, c->cells, c->grid);
    } else if (g_lo + g_hi) {
        // The one-message step: the neighbours' two layers came with their headers and leavers (behind those, in the same
        // buffer); no message here.  A side this rank sent nobody to is unpacked as it is; else its own leavers -- still in the
        // send buffer -- are merged in where the neighbour will put them (k_slab_unpack_ghosts_merge), then the cells.
        const float4* res_lo = s->mig_recv[0] + 2 * (1 + (size_t)in_lo);
        const float4* res_hi = s->mig_recv[1] + 2 * (1 + (size_t)in_hi);
        const uint32_t nr_lo = s->has_lo ? peer_own_lo + n2p_lo : 0u, nr_hi = s->has_hi ? peer_own_hi + n2p_hi : 0u;
        if (m_lo + m_hi == 0u) {
            hipLaunchKernelGGL(k_slab_unpack_ghosts, dim3(ceil_div(g_lo + g_hi, 256u)), dim3(256), 0, s->comm, res_lo, g_lo, c->own_off - g_lo,
                               res_hi, g_hi, c->own_off + n, c->posi, c->velr, c->keyS, c->cells, c->grid);
        } else {
            hipLaunchKernelGGL(k_slab_unpack_ghosts_merge, dim3(ceil_div(g_lo + g_hi, 256u)), dim3(256), 0, s->comm, res_lo, nr_lo,
                               s->mig_send[0] + 2, s->has_lo ? m_lo : 0u, c->own_off - g_lo, res_hi, nr_hi, s->mig_send[1] + 2, s->has_hi ? m_hi : 0u,
                               c->own_off + n, c->posi, c->velr, c->keyS, c->grid);
            OnComm on(s);
            rc = launch_cells_build_2ranges(c, c->own_off - g_lo, c->own_off, c->own_off + n, c->own_off + n + g_hi);
            if (rc) return rc;
        }
    }
    SPH_HIP(hipGetLastError());
    c->n_glo = g_lo; c->n_ghi = g_hi;
    s->ghosts += g_lo + g_hi;
    c->cells_lo = c->own_off - g_lo; c->cells_hi = c->own_off + n + g_hi; c->cells_valid = true;
    c->stage = sph_ctx::ST_CELLS;
    if (early_halo) {
        // all that is left of the density pass (the boundary layers and the two layers next to them), queued on the COMM
        // stream behind the ghosts: it runs BESIDE the tail of the deep launch instead of behind it (a launch this
        // small is one partly filled round of workgroups; back to back the two launches cost a round more).  The
        // boundary layers' (rho, p) come out of this launch, so their halo-B message is packed on the same stream: no
        // event recorded on the main stream in between (each costs the device ~5 us of idle at the next dispatch).
        OnComm on(s);
        {
            PhaseTimer t(c, SPH_PH_DENS);
            rc = launch_density_hole(c, dens_lo, dens_hi, deep_lo, deep_hi);
            if (rc) return rc;
        }
        if (!p1 && h_lo + h_hi)
            hipLaunchKernelGGL(k_slab_copy_dp2, dim3(ceil_div(h_lo + h_hi, 256u)), dim3(256), 0, c->stream, c->dp + c->own_off,
                               s->dens_send[0], h_lo, c->dp + c->own_off + n - h_hi, s->dens_send[1], h_hi);
        SPH_HIP(hipGetLastError());
    }
    rc = after_comm(s); if (rc) return rc;                      // main: the ghosts (and, early, the boundary densities) are in
    if (!early_halo) {
        {
            PhaseTimer t(c, SPH_PH_DENS);                        // the two boundary layers
            rc = launch_density_hole(c, dens_lo, dens_hi, a, b);
            if (rc) return rc;
        }
        // ---- halo B: (density, pressure) of the same boundary particles, same order ---------------------------------
        if (!p1 && h_lo + h_hi)
            hipLaunchKernelGGL(k_slab_copy_dp2, dim3(ceil_div(h_lo + h_hi, 256u)), dim3(256), 0, c->stream, c->dp + c->own_off,
                               s->dens_send[0], h_lo, c->dp + c->own_off + n - h_hi, s->dens_send[1], h_hi);
        SPH_HIP(hipGetLastError());
        rc = after_main(s); if (rc) return rc;
    }
    c->have_dens = true;
    // the interior forces run while halo B travels
    const bool mark = force_begin(c, true);
    {
        PhaseTimer t(c, SPH_PH_FORCE);
        // what the early launch did is left out in whole 64-slot chunks of the FINAL owned range (the up to 63 slots at
        // either end of its range are simply computed again: same inputs, same bits)
        uint32_t h0 = 0, h1 = 0;
        if (early_launched && deep_valid && early_hi > early_lo && early_lo >= c->own_off) {
            h0 = c->own_off + ((early_lo - c->own_off + 63u) & ~63u);
            h1 = c->own_off + ((early_hi - c->own_off) & ~63u);
        }
        if (h1 > h0 && h0 >= a && h1 <= b) {
            rc = launch_force_hole(c, a, b, h0, h1, true, true, true, dt, mark);                     // interior: queued before the transfer
            if (rc) return rc;
            rc = join_early(); if (rc) return rc;                   // (its tail has run beside the launch above)
            hipLaunchKernelGGL(k_slab_early_finish, dim3(ceil_div(h1 - h0, 256u)), dim3(256), 0, c->stream, c->keyS2, c->keyS, h0, h1,
                               c->own_off, c->k0, mark ? c->mm_mask : (uint64_t*)nullptr, c->mm_tile_cnt);
            SPH_HIP(hipGetLastError());
            s->early_used++;
        } else {
            rc = join_early(); if (rc) return rc;                   // (both write the same slots of the ping-pong arrays: same values, but in order)
            rc = launch_force_range(c, a, b, true, true, true, dt, mark);
        }
    }
    if (rc) return rc;
    if (!p1) {
        rc = slab_exchange(s, SPH_TAG_HALO_B, s->dens_send[0], h_lo * sizeof(float2), s->dens_recv[0], g_lo * sizeof(float2),
                           s->dens_send[1], h_hi * sizeof(float2), s->dens_recv[1], g_hi * sizeof(float2));
        if (rc) return rc;
        s->pg.halo_b = true;
        if (g_lo + g_hi)
            hipLaunchKernelGGL(k_slab_unpack_dp2, dim3(ceil_div(g_lo + g_hi, 256u)), dim3(256), 0, s->comm, s->dens_recv[0],
                               c->dp + c->own_off - g_lo, c->cw + c->own_off - g_lo, g_lo, s->dens_recv[1], c->dp + c->own_off + n,
                               c->cw + c->own_off + n, g_hi, c->phys);
        SPH_HIP(hipGetLastError());
    }   // (the one-message step: the inner ghost layers' densities came out of this rank's own density launches above)
    // the boundary layers' force pass: on the comm stream, behind the ghosts' (rho, p) -- beside the interior launch
    // (it reads what that one reads and writes other slots of the ping-pong arrays), not behind it
    {
        if (need_deep_event) { rc = hop_wait(s, HOP_DEEP, s->comm, s->ev_deep, deep_v); if (rc) return rc; }  // (non-early: ev_main above covers the densities)
        OnComm on(s);
        PhaseTimer t(c, SPH_PH_FORCE);
        rc = launch_force_hole(c, c->own_off, c->own_off + n, a, b, true, true, true, dt, mark);
        if (rc) return rc;
    }
    rc = after_comm(s); if (rc) return rc;                      // the mover count (force_finish) needs both launches
    force_finish(c, true, mark);
    c->have_force = c->have_coll = false;
    // (no event here: the comm stream's first action of the next step waits for an event the main stream records behind
    // k_slab_bounds_pack, i.e. behind everything queued above)
    s->steps++;
    // what either end of each link packed this step: the size rule of the next one-message step (both ends hold the same numbers)
    for (int k = 0; k < 2; k++) { s->one_prev_s[k] = tot_s[k]; s->one_prev_r[k] = tot_r[k]; }
    s->one_ready = true;
    if (c->timing) { c->timed_steps++; if (c->events.size() > 3 * 4096) timing_collect(c); }
    {
        const auto t_end = std::chrono::steady_clock::now();
        s->t_post.add(std::chrono::duration<double, std::micro>(t_end - t_waited).count());
        s->t_host.add(std::chrono::duration<double, std::micro>(t_end - t_begin).count());
    }
    if (s->pending.size() > 4096) { rc = slab_timing_collect(s); if (rc) return rc; }
    return SPH_OK;
}

// A rank that fails does not simply return: its neighbours have sized this step's messages from the headers and are
// about to post them, and with RCCL a receive whose sender never sends stays on the comm stream for ever.  So the rank
// (1) still exchanges what the step owes -- the numbers of Progress, contents irrelevant: the run is over --, (2) puts
// "abort" into the header of its NEXT migrant message and exchanges that too, (3) is marked failed: every later call
// returns the first error.  A neighbour reads the abort word at its next wait, returns SPH_E_PEER and does the same
// towards ITS other neighbour: the failure reaches rank r +- k after k steps, nobody waits for a timeout.  A step that
// fails before its own migrant message was posted sends the abort header AS that message (same step).  If the
// transport itself failed (or a wait timed out: the neighbour is gone) nothing more is exchanged and the transport is
// aborted (RCCL: ncclCommAbort), so that destroy / sync do not block on a receive that will never complete.
int slab_fail(sph_slab* s, int rc) {
    if (s->failed) return s->failed;
    s->failed = rc;
    snprintf(s->fail_msg, sizeof s->fail_msg, "%s", sph_last_error());
    sph_slab::Progress& g = s->pg;
    const size_t rec = 2 * sizeof(float4);
    const uint32_t inl = min(MIG_INLINE, s->mcap);
    int e = SPH_OK;
    if (!s->transport_dead && !g.mig_posted && s->world > 1) {
        // the step failed BEFORE its migrant message went out (a device-side flag of the last step, the sort, a launch):
        // the neighbours are about to post theirs, so the abort header travels as THIS step's migrant message -- they
        // read it at this step's wait and stop; nothing else is owed (no header of this rank promised anything)
        g.mig_posted = true;
        after_main(s);          // (a bounds kernel of this step may be queued on the main stream: it writes the same header words)
        hipLaunchKernelGGL(k_slab_abort_headers, dim3(1), dim3(64), 0, s->comm, s->mig_send[0], s->mig_send[1]);
        const size_t mig_bytes = (size_t)(1 + inl) * rec;
        if (g.one)              // the neighbours have posted this step's ONE message at the sizes the rule gave all of us
            e = slab_exchange(s, SPH_TAG_ONE, s->mig_send[0], (1 + (size_t)g.one_s[0]) * rec, s->mig_recv[0], (1 + (size_t)g.one_r[0]) * rec,
                              s->mig_send[1], (1 + (size_t)g.one_s[1]) * rec, s->mig_recv[1], (1 + (size_t)g.one_r[1]) * rec);
        else
            e = slab_exchange(s, SPH_TAG_MIGRANTS, s->mig_send[0], mig_bytes, s->mig_recv[0], mig_bytes, s->mig_send[1], mig_bytes,
                              s->mig_recv[1], mig_bytes);
    } else if (!s->transport_dead && g.headers && s->world > 1) {
        if (!g.rest && g.one && (g.rest_s[0] | g.rest_s[1] | g.rest_r[0] | g.rest_r[1]))
            e = slab_exchange(s, SPH_TAG_ONE_REST, s->mig_send[0] + 2 * (1 + (size_t)g.one_s[0]), g.rest_s[0] * rec,
                              s->mig_recv[0] + 2 * (1 + (size_t)g.one_r[0]), g.rest_r[0] * rec, s->mig_send[1] + 2 * (1 + (size_t)g.one_s[1]),
                              g.rest_s[1] * rec, s->mig_recv[1] + 2 * (1 + (size_t)g.one_r[1]), g.rest_r[1] * rec);
        else if (!g.rest && (g.rest_s[0] | g.rest_s[1] | g.rest_r[0] | g.rest_r[1]))
            e = slab_exchange(s, SPH_TAG_MIGRANTS_REST, s->mig_send[0] + 2 * (1 + inl), g.rest_s[0] * rec, s->mig_recv[0] + 2 * (1 + inl),
                              g.rest_r[0] * rec, s->mig_send[1] + 2 * (1 + inl), g.rest_s[1] * rec, s->mig_recv[1] + 2 * (1 + inl),
                              g.rest_r[1] * rec);
        if (!e && !g.halo_a)
            e = slab_exchange(s, SPH_TAG_HALO_A, s->halo_send[0], g.h[0] * rec, s->halo_recv[0], g.g[0] * rec, s->halo_send[1],
                              g.h[1] * rec, s->halo_recv[1], g.g[1] * rec);
        if (!e && !g.halo_b)
            e = slab_exchange(s, SPH_TAG_HALO_B, s->dens_send[0], g.h[0] * sizeof(float2), s->dens_recv[0], g.g[0] * sizeof(float2),
                              s->dens_send[1], g.h[1] * sizeof(float2), s->dens_recv[1], g.g[1] * sizeof(float2));
        if (!e) {                                               // the next step's header: "abort"
            hipLaunchKernelGGL(k_slab_abort_headers, dim3(1), dim3(64), 0, s->comm, s->mig_send[0], s->mig_send[1]);
            const size_t mig_bytes = (size_t)(1 + inl) * rec;
            // (the neighbours, for whom this step went through, take their next step under the slab's protocol: a one-message
            // step is sized from THIS step's headers, which both ends have)
            if (s->protocol == 1 && g.next_known && g.next_s[0] <= s->msg_cap && g.next_s[1] <= s->msg_cap && g.next_r[0] <= s->msg_cap &&
                g.next_r[1] <= s->msg_cap)
                e = slab_exchange(s, SPH_TAG_ONE, s->mig_send[0], (1 + (size_t)(s->has_lo ? g.next_s[0] : 0u)) * rec, s->mig_recv[0],
                                  (1 + (size_t)(s->has_lo ? g.next_r[0] : 0u)) * rec, s->mig_send[1], (1 + (size_t)(s->has_hi ? g.next_s[1] : 0u)) * rec,
                                  s->mig_recv[1], (1 + (size_t)(s->has_hi ? g.next_r[1] : 0u)) * rec);
            else
                e = slab_exchange(s, SPH_TAG_MIGRANTS, s->mig_send[0], mig_bytes, s->mig_recv[0], mig_bytes, s->mig_send[1], mig_bytes,
                                  s->mig_recv[1], mig_bytes);
        }
    }
    if ((s->transport_dead || e) && s->tr.abort) { s->tr.abort(s->tr.self); s->transport_dead = true; }
    set_error("%s", s->fail_msg);
    return rc;
}

int slab_step_once(sph_slab* s, float dt) {
    if (s->failed) { set_error("%s", s->fail_msg); return s->failed; }
    const int rc = slab_step_body(s, dt);
    return rc ? slab_fail(s, rc) : SPH_OK;
}

}  // namespace

extern "C" {

int sph_rccl_unique_id(uint8_t id[128]) {
    SPH_REQUIRE(id, SPH_E_INVALID, "null argument");
    int rc = rccl_load();
    if (rc) return rc;
    SPH_NCCL(g_rccl.GetUniqueId(id));
    return SPH_OK;
}

int sph_rccl_transport_create(sph_transport** out, const uint8_t id[128], int rank, int world, int device) {
    SPH_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, SPH_E_INVALID, "bad argument");
    *out = nullptr;
    int rc = rccl_load();
    if (rc) return rc;
    if (device < 0) device = sph_selected_device();
    SPH_HIP(hipSetDevice(device));
    RcclLink* L = new (std::nothrow) RcclLink();
    sph_transport* t = new (std::nothrow) sph_transport();
    if (!L || !t) { delete L; delete t; set_error("out of host memory"); return SPH_E_NOMEM; }
    L->rank = rank; L->world = world;
    Id128 uid;
    memcpy(uid.bytes, id, 128);
    int r = g_rccl.CommInitRank(&L->comm, world, uid, rank);
    if (r != 0) {
        set_error("ncclCommInitRank failed: %d (%s)", r, g_rccl.GetErrorString(r));
        delete L; delete t;
        return SPH_E_DEVICE;
    }
    t->self = L;
    t->exchange = rccl_exchange;
    t->host_buffers = 0;
    t->abort = rccl_abort;
    *out = t;
    return SPH_OK;
}

void sph_rccl_transport_destroy(sph_transport* t) {
    if (!t) return;
    RcclLink* L = (RcclLink*)t->self;
    if (L) { if (L->comm && !L->aborted && g_rccl.CommDestroy) g_rccl.CommDestroy(L->comm); delete L; }
    delete t;
}

// One rank is enough to drive real bytes through the loaded ncclSend/ncclRecv: two messages to SELF in one group (a
// send to the own rank is matched with the receive from the own rank, in the order they were posted), on a stream of
// its own, compared on the host.  What a wrong signature, datatype value or struct-by-value convention of the dlopen
// binding would break shows up here, on a one-GPU box; the neighbour pattern itself needs >= 2 GPUs.
int sph_rccl_transport_selftest(sph_transport* t, size_t bytes) {
    SPH_REQUIRE(t && t->self && t->exchange == rccl_exchange, SPH_E_INVALID, "not an RCCL transport");
    SPH_REQUIRE(bytes >= 16 && bytes <= ((size_t)1 << 30), SPH_E_INVALID, "bad size");
    RcclLink* L = (RcclLink*)t->self;
    char *sa = nullptr, *sb = nullptr, *ra = nullptr, *rb = nullptr;
    hipStream_t st = nullptr;
    int rc = SPH_OK;
    std::vector<char> ha(bytes), hb(bytes);
    auto fail = [&](const char* what, hipError_t e) { set_error("selftest: %s: %s", what, hipGetErrorString(e)); rc = SPH_E_DEVICE; };
    hipError_t e;
    if ((e = hipMalloc((void**)&sa, bytes)) || (e = hipMalloc((void**)&sb, bytes)) || (e = hipMalloc((void**)&ra, bytes)) ||
        (e = hipMalloc((void**)&rb, bytes)) || (e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)))
        fail("allocation", e);
    if (!rc) {
        for (size_t i = 0; i < bytes; i++) { ha[i] = (char)(i * 7u + 1u); hb[i] = (char)(i * 13u + 5u); }
        if ((e = hipMemcpy(sa, ha.data(), bytes, hipMemcpyHostToDevice)) || (e = hipMemcpy(sb, hb.data(), bytes, hipMemcpyHostToDevice)) ||
            (e = hipMemset(ra, 0, bytes)) || (e = hipMemset(rb, 0, bytes)))
            fail("fill", e);
    }
    if (!rc) {
        const int ncclChar = 0;
        int r = g_rccl.GroupStart();
        if (!r) r = g_rccl.Send(sa, bytes, ncclChar, L->rank, L->comm, st);
        if (!r) r = g_rccl.Recv(ra, bytes, ncclChar, L->rank, L->comm, st);
        if (!r) r = g_rccl.Send(sb, bytes / 2, ncclChar, L->rank, L->comm, st);
        if (!r) r = g_rccl.Recv(rb, bytes / 2, ncclChar, L->rank, L->comm, st);
        const int r2 = g_rccl.GroupEnd();
        if (r || r2) { set_error("selftest: RCCL error %d (%s)", r ? r : r2, g_rccl.GetErrorString(r ? r : r2)); rc = SPH_E_DEVICE; }
    }
    if (!rc && (e = hipStreamSynchronize(st))) fail("synchronize", e);
    if (!rc) {
        std::vector<char> ga(bytes), gb(bytes);
        if ((e = hipMemcpy(ga.data(), ra, bytes, hipMemcpyDeviceToHost)) || (e = hipMemcpy(gb.data(), rb, bytes, hipMemcpyDeviceToHost)))
            fail("read back", e);
        else if (memcmp(ga.data(), ha.data(), bytes) != 0 || memcmp(gb.data(), hb.data(), bytes / 2) != 0) {
            set_error("selftest: received bytes differ from the bytes sent");
            rc = SPH_E_STATE;
        } else {
            for (size_t i = bytes / 2; i < bytes; i++)
                if (gb[i] != 0) { set_error("selftest: a %zu-byte receive wrote past its end", bytes / 2); rc = SPH_E_STATE; break; }
        }
    }
    if (st) hipStreamDestroy(st);
    hipFree(sa); hipFree(sb); hipFree(ra); hipFree(rb);
    return rc;
}

int sph_local_hub_create(sph_local_hub** out, int world, int device) {
    SPH_REQUIRE(out && world >= 1, SPH_E_INVALID, "bad argument");
    *out = nullptr;
    if (device < 0) device = sph_selected_device();
    SPH_HIP(hipSetDevice(device));
    sph_local_hub* H = new (std::nothrow) sph_local_hub();
    SPH_REQUIRE(H, SPH_E_NOMEM, "out of host memory");
    H->world = world;
    if (const char* e = getenv("SPH_SLAB_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) H->timeout_s = v; }
    H->links.resize(2 * (size_t)world);
    for (LocalLink& L : H->links)
        if (hipEventCreateWithFlags(&L.ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.done, hipEventDisableTiming) != hipSuccess) {
            set_error("sph_local_hub_create: hipEventCreate failed");
            sph_local_hub_destroy(H);
            return SPH_E_DEVICE;
        }
    *out = H;
    return SPH_OK;
}

void sph_local_hub_destroy(sph_local_hub* H) {
    if (!H) return;
    for (LocalLink& L : H->links) { if (L.ready) hipEventDestroy(L.ready); if (L.done) hipEventDestroy(L.done); }
    delete H;
}

int sph_local_hub_set_timeout(sph_local_hub* H, double seconds) {
    SPH_REQUIRE(H && seconds > 0.0, SPH_E_INVALID, "bad argument");
    H->timeout_s = seconds;
    return SPH_OK;
}

int sph_loop_transport_create(sph_transport** out, float z_shift, double link_gbs, double latency_us) {
    SPH_REQUIRE(out && z_shift > 0.f && link_gbs >= 0.0 && latency_us >= 0.0, SPH_E_INVALID, "bad argument");
    *out = nullptr;
    LoopEnd* E = new (std::nothrow) LoopEnd{z_shift, link_gbs, latency_us};
    sph_transport* t = new (std::nothrow) sph_transport();
    if (!E || !t) { delete E; delete t; set_error("out of host memory"); return SPH_E_NOMEM; }
    t->self = E;
    t->exchange = loop_exchange;
    t->host_buffers = 0;
    t->abort = nullptr;
    *out = t;
    return SPH_OK;
}

void sph_loop_transport_destroy(sph_transport* t) {
    if (!t) return;
    delete (LoopEnd*)t->self;
    delete t;
}

int sph_local_transport_create(sph_transport** out, sph_local_hub* hub, int rank) {
    SPH_REQUIRE(out && hub && rank >= 0 && rank < hub->world, SPH_E_INVALID, "bad argument");
    *out = nullptr;
    LocalEnd* E = new (std::nothrow) LocalEnd{hub, rank};
    sph_transport* t = new (std::nothrow) sph_transport();
    if (!E || !t) { delete E; delete t; set_error("out of host memory"); return SPH_E_NOMEM; }
    t->self = E;
    t->exchange = local_exchange;
    t->host_buffers = 0;
    t->abort = nullptr;               // (its waits are host-side and bounded: nothing stays queued on a stream)
    *out = t;
    return SPH_OK;
}

void sph_local_transport_destroy(sph_transport* t) {
    if (!t) return;
    delete (LocalEnd*)t->self;
    delete t;
}

int sph_slab_create(sph_slab** out, sph_ctx* ctx, int rank, int world, const sph_transport* transport,
                    uint32_t migrant_capacity) {
    SPH_REQUIRE(out && ctx && transport && transport->exchange, SPH_E_INVALID, "null argument");
    *out = nullptr;
    SPH_REQUIRE(ctx->slab, SPH_E_INVALID, "sph_slab_create needs a context made by sph_create_slab");
    SPH_REQUIRE(world >= 1 && rank >= 0 && rank < world, SPH_E_INVALID, "bad rank %d of %d", rank, world);
    SPH_REQUIRE(world == 1 || ctx->z_hi - ctx->z_lo >= 2, SPH_E_INVALID,
                "a slab needs at least two cell layers (its two boundary layers must be different layers)");
    SPH_HIP(hipSetDevice(ctx->device));
    sph_slab* s = new (std::nothrow) sph_slab();
    SPH_REQUIRE(s, SPH_E_NOMEM, "out of host memory");
    s->c = ctx; s->rank = rank; s->world = world;
    s->has_lo = rank > 0; s->has_hi = rank + 1 < world;
    s->device = ctx->device;
    if (const char* e = getenv("SPH_SLAB_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) s->wait_timeout_s = v; }
    if (const char* e = getenv("SPH_SLAB_EARLY_STREAM")) s->early_own_stream = atoi(e) != 0;
    if (const char* e = getenv("SPH_SLAB_EARLY_SPAN")) { const long v = atol(e); if (v > 0) s->early_cap = (uint32_t)v; }
    s->tr = *transport;
    s->host_staged = transport->host_buffers != 0;
    s->gcap = ctx->gcap;
    // default: half a ghost layer's capacity -- a whole lattice layer can cross a cut in one step; only the first 255 records
    // of a side travel every step (the fixed-size message), so the capacity costs memory, not bandwidth
    s->mcap = migrant_capacity ? migrant_capacity : (ctx->gcap / 2u + 1024u);
    if (s->mcap > ctx->gcap) s->mcap = ctx->gcap;
    int lo_pri = 0, hi_pri = 0;
    hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri);
    bool ok = hipStreamCreateWithPriority(&s->comm, hipStreamNonBlocking, hi_pri) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_comm, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_deep, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&s->early, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_early_go, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_early_done, hipEventDisableTiming) == hipSuccess &&
              hipMalloc((void**)&s->d_lb, DL_WORDS * sizeof(uint32_t)) == hipSuccess &&
              hipMemset(s->d_lb, 0, DL_WORDS * sizeof(uint32_t)) == hipSuccess &&
              hipHostMalloc((void**)&s->h_lb, HL_WORDS * sizeof(uint32_t), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void**)&s->h_lb_dev, (void*)s->h_lb, 0) == hipSuccess;
    // (a context with two ghost layers may run the one-message step: leavers AND two layers of residents in one buffer)
    s->msg_cap = ctx->ghost_layers >= 2u ? s->mcap + s->gcap + 64u : s->mcap;
    const size_t mig_bytes = (size_t)(1 + s->msg_cap) * 2 * sizeof(float4), halo_bytes = (size_t)(s->gcap + 1) * 2 * sizeof(float4);
    for (int k = 0; k < 2 && ok; k++)
        ok = hipMalloc((void**)&s->mig_send[k], mig_bytes) == hipSuccess && hipMalloc((void**)&s->mig_recv[k], mig_bytes) == hipSuccess &&
             hipMalloc((void**)&s->halo_send[k], halo_bytes) == hipSuccess && hipMalloc((void**)&s->halo_recv[k], halo_bytes) == hipSuccess &&
             hipMalloc((void**)&s->dens_send[k], (size_t)(s->gcap + 1) * sizeof(float2)) == hipSuccess &&
             hipMalloc((void**)&s->dens_recv[k], (size_t)(s->gcap + 1) * sizeof(float2)) == hipSuccess &&
             hipMemset(s->mig_send[k], 0, mig_bytes) == hipSuccess && hipMemset(s->mig_recv[k], 0, mig_bytes) == hipSuccess;
    if (ok && s->host_staged) {
        s->stage_bytes = mig_bytes > halo_bytes ? mig_bytes : halo_bytes;
        for (int k = 0; k < 2 && ok; k++)
            ok = hipHostMalloc((void**)&s->stage_send[k], s->stage_bytes) == hipSuccess &&
                 hipHostMalloc((void**)&s->stage_recv[k], s->stage_bytes) == hipSuccess;
    }
    if (!ok) { set_error("sph_slab_create: allocation failed"); slab_free(s); return SPH_E_NOMEM; }
    {   // cross-stream edges by write / wait value where that works on this device (see hop_mem)
        const char* e = getenv("SPH_SLAB_HOPS");
        int can = 0;
        if (!(e && e[0] == 'e') && hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) == hipSuccess && can &&
            hipMalloc((void**)&s->hop_mem, 5 * 16 * sizeof(uint32_t)) == hipSuccess && hipMemset(s->hop_mem, 0, 5 * 16 * sizeof(uint32_t)) == hipSuccess &&
            hipDeviceSynchronize() == hipSuccess) {
            // one pair each way, waited for here: a runtime that accepts the calls but cannot serve them must not be found out mid-step
            s->hops_by_value = true;
            const bool fine = after_main(s) == SPH_OK && after_comm(s) == SPH_OK && hipStreamSynchronize(s->comm) == hipSuccess &&
                              hipStreamSynchronize(ctx->stream) == hipSuccess;
            if (!fine) { s->hops_by_value = false; (void)hipGetLastError(); }
        }
        if (!s->hops_by_value && s->hop_mem) { hipFree(s->hop_mem); s->hop_mem = nullptr; }
    }
    for (int k = 0; k < HL_WORDS; k++) s->h_lb[k] = 0u;
    ctx->host_paced = true;            // sph_slab_step waits for the device once per step
    *out = s;
    return SPH_OK;
}

// Draining the comm stream of a slab that FAILED must not wait for ever: its last act was to queue an "abort" migrant
// message for the neighbours' next step, and if they never take another step (the failure came on the run's last step, or
// they failed themselves) that send / receive pair is never matched.  Poll, bounded by the step's wait time-out, then take
// the transport down (RCCL: ncclCommAbort) and only then synchronise.
static void slab_drain_comm(sph_slab* s) {
    if (s->transport_dead && s->tr.abort) s->tr.abort(s->tr.self);     // (idempotent) nothing of a dead link stays queued
    if (s->failed && !s->transport_dead) {
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipStreamQuery(s->comm)) == hipErrorNotReady) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->wait_timeout_s) {
                s->transport_dead = true;
                if (s->tr.abort) s->tr.abort(s->tr.self);
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    hipStreamSynchronize(s->comm);
}

// must run BEFORE sph_destroy of its context (it drains the context's stream, on which its kernels run)
void sph_slab_destroy(sph_slab* s) {
    if (!s) return;
    hipSetDevice(s->device);
    slab_drain_comm(s);
    if (s->early) hipStreamSynchronize(s->early);
    hipStreamSynchronize(s->c->stream);
    s->c->host_paced = false;
    slab_free(s);
}

int sph_slab_set_wait_timeout(sph_slab* s, double seconds) {
    SPH_REQUIRE(s && seconds > 0.0, SPH_E_INVALID, "bad argument");
    s->wait_timeout_s = seconds;
    return SPH_OK;
}

int sph_slab_step(sph_slab* s, float dt, uint32_t n_steps) {
    SPH_REQUIRE(s, SPH_E_INVALID, "null slab");
    SPH_HIP(hipSetDevice(s->c->device));
    for (uint32_t k = 0; k < n_steps; k++) {
        int rc = slab_step_once(s, dt);
        if (rc) return rc;
    }
    return SPH_OK;
}

int sph_slab_sync(sph_slab* s) {
    SPH_REQUIRE(s, SPH_E_INVALID, "null slab");
    SPH_HIP(hipSetDevice(s->c->device));
    slab_drain_comm(s);
    SPH_HIP(hipStreamQuery(s->comm));
    if (s->early) SPH_HIP(hipStreamSynchronize(s->early));
    int rc = sph_sync(s->c);
    if (rc) return rc;
    return slab_check_device_flags(s);
}

int sph_slab_set_protocol(sph_slab* s, int groups) {
    SPH_REQUIRE(s, SPH_E_INVALID, "null slab");
    SPH_REQUIRE(groups == 1 || groups == 3, SPH_E_INVALID, "the slab step has a one-message and a three-group protocol, not %d", groups);
    if (groups == 1) {
        SPH_REQUIRE(s->c->ghost_layers >= 2u, SPH_E_STATE, "the one-message step needs a context with two ghost layers (sph_create_slab_layers)");
        // the two layers sent to either neighbour must be four different layers: what arrives from one side (even a particle
        // that crossed two layers) then never lands in a layer the OTHER neighbour holds a copy of
        SPH_REQUIRE(s->world == 1 || s->c->z_hi - s->c->z_lo >= 4u, SPH_E_STATE,
                    "the one-message step needs slabs of at least four cell layers (this one: %u)", s->c->z_hi - s->c->z_lo);
    }
    s->protocol = groups;
    s->one_ready = false;                                   // the next step runs the three-group protocol and learns the sizes
    return SPH_OK;
}

int sph_slab_protocol(const sph_slab* s, uint64_t out[3]) {
    SPH_REQUIRE(s && out, SPH_E_INVALID, "null argument");
    out[0] = (uint64_t)s->protocol; out[1] = s->one_steps; out[2] = s->one_rest_msgs;
    return SPH_OK;
}

int sph_slab_set_early_force(sph_slab* s, int on) {
    SPH_REQUIRE(s, SPH_E_INVALID, "null slab");
    s->early_force = on != 0;
    if (on > 1) s->early_cap = (uint32_t)on;
    return SPH_OK;
}

int sph_slab_early_force_stats(const sph_slab* s, uint64_t out[2]) {
    SPH_REQUIRE(s && out, SPH_E_INVALID, "null argument");
    out[0] = s->early_launches; out[1] = s->early_used;
    return SPH_OK;
}

int sph_slab_timing_enable(sph_slab* s, int on) {
    SPH_REQUIRE(s, SPH_E_INVALID, "null slab");
    s->time_groups = on != 0;
    return SPH_OK;
}

int sph_slab_timing_reset(sph_slab* s) {
    SPH_REQUIRE(s, SPH_E_INVALID, "null slab");
    SPH_HIP(hipSetDevice(s->c->device));
    int rc = slab_timing_collect(s);
    if (rc) return rc;
    s->t_wait = s->t_pre = s->t_post = s->t_host = sph_slab::Acc();
    for (auto& g : s->t_group) g = sph_slab::Acc();
    s->waits_ready = 0;
    return SPH_OK;
}

int sph_slab_timing_get(sph_slab* s, double out[SPH_SLAB_T_WORDS]) {
    SPH_REQUIRE(s && out, SPH_E_INVALID, "null argument");
    SPH_HIP(hipSetDevice(s->c->device));
    int rc = slab_timing_collect(s);
    if (rc) return rc;
    for (int k = 0; k < SPH_SLAB_T_WORDS; k++) out[k] = 0.0;
    out[SPH_SLAB_T_STEPS] = (double)s->t_host.n;
    out[SPH_SLAB_T_WAITS_READY] = (double)s->waits_ready;
    const sph_slab::Acc* host[4] = {&s->t_wait, &s->t_pre, &s->t_post, &s->t_host};
    for (int k = 0; k < 4; k++) { out[SPH_SLAB_T_WAIT + 2 * k] = host[k]->sum; out[SPH_SLAB_T_WAIT + 2 * k + 1] = host[k]->max; }
    for (int g = 0; g < 4; g++) {
        out[SPH_SLAB_T_GROUPS + 3 * g] = (double)s->t_group[g].n;
        out[SPH_SLAB_T_GROUPS + 3 * g + 1] = s->t_group[g].sum;
        out[SPH_SLAB_T_GROUPS + 3 * g + 2] = s->t_group[g].max;
    }
    for (int g = 0; g < 2; g++) {
        const sph_slab::Acc& a = s->t_group[SPH_TAG_ONE - 1 + g];
        out[SPH_SLAB_T_GROUPS_ONE + 3 * g] = (double)a.n; out[SPH_SLAB_T_GROUPS_ONE + 3 * g + 1] = a.sum; out[SPH_SLAB_T_GROUPS_ONE + 3 * g + 2] = a.max;
    }
    return SPH_OK;
}

// One slab_exchange-shaped group of `bytes` per direction to rank - 1 and rank + 1, `reps` times (+ one untimed round
// first: RCCL connects to a new peer on first use), on the comm stream, through the slab's own transport and buffers.
// Every word of a message names its sender, direction and round and is checked on arrival, so a ring wired the wrong
// way round fails here and not as wrong physics.  COLLECTIVE: every rank of the chain calls it with the same arguments.
int sph_slab_ping(sph_slab* s, size_t bytes, uint32_t reps, double out[3]) {
    SPH_REQUIRE(s && out && reps >= 1, SPH_E_INVALID, "bad argument");
    SPH_REQUIRE(!s->failed, s->failed, "%s", s->fail_msg);
    const size_t cap = (size_t)(s->gcap + 1) * 2 * sizeof(float4);
    SPH_REQUIRE(bytes >= 4 && bytes % 4 == 0 && bytes <= cap, SPH_E_INVALID, "ping of %zu bytes: must be a multiple of 4 up to the halo "
                "buffer's %zu", bytes, cap);
    SPH_HIP(hipSetDevice(s->c->device));
    out[0] = out[1] = out[2] = 0.0;
    if (s->world < 2) return SPH_OK;
    const uint32_t words = (uint32_t)(bytes / 4), grid = min(ceil_div(words, 256u), 1024u);
    uint32_t* bad = s->d_lb + DL_PING;
    SPH_HIP(hipMemsetAsync(bad, 0, sizeof(uint32_t), s->comm));
    const uint64_t exchanges0 = s->exchanges;
    const bool timed0 = s->time_groups;
    const sph_slab::Acc acc0 = s->t_group[4];
    int rc = slab_timing_collect(s);
    s->t_group[4] = sph_slab::Acc();
    for (uint32_t rep = 0; rep <= reps && !rc; rep++) {
        s->time_groups = rep > 0;                             // round 0 is the connection set-up
        hipLaunchKernelGGL(k_slab_ping_fill, dim3(grid), dim3(256), 0, s->comm, s->has_lo ? (uint32_t*)s->halo_send[0] : nullptr,
                           s->has_hi ? (uint32_t*)s->halo_send[1] : nullptr, words, (uint32_t)s->rank, rep);
        rc = slab_exchange(s, SPH_TAG_PING, s->halo_send[0], bytes, s->halo_recv[0], bytes, s->halo_send[1], bytes, s->halo_recv[1], bytes);
        if (rc) break;
        hipLaunchKernelGGL(k_slab_ping_check, dim3(grid), dim3(256), 0, s->comm, s->has_lo ? (const uint32_t*)s->halo_recv[0] : nullptr,
                           s->has_hi ? (const uint32_t*)s->halo_recv[1] : nullptr, words, (uint32_t)s->rank, rep, bad);
        if (hipGetLastError() != hipSuccess) { set_error("ping: kernel launch failed"); rc = SPH_E_DEVICE; }
    }
    s->time_groups = timed0;
    s->exchanges = exchanges0;                                // (the step counters keep counting steps' messages only)
    if (!rc) {
        // bounded: a neighbour that never answers must be an error message, not a hang
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipStreamQuery(s->comm)) == hipErrorNotReady) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->wait_timeout_s) {
                set_error("rank %d: ping of %zu bytes: no answer from a neighbour after %.0f s", s->rank, bytes, s->wait_timeout_s);
                s->transport_dead = true;
                if (s->tr.abort) s->tr.abort(s->tr.self);
                rc = SPH_E_DEVICE;
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        if (!rc && q != hipSuccess) { set_error("ping: %s", hipGetErrorString(q)); rc = SPH_E_DEVICE; }
    }
    uint32_t wrong = 0;
    if (!rc && hipMemcpy(&wrong, bad, sizeof wrong, hipMemcpyDeviceToHost) != hipSuccess) { set_error("ping: read back failed"); rc = SPH_E_DEVICE; }
    if (!rc) rc = slab_timing_collect(s);
    if (!rc) {
        out[0] = s->t_group[4].n ? s->t_group[4].sum / (double)s->t_group[4].n : 0.0;
        out[1] = s->t_group[4].max;
        out[2] = (double)wrong;
        if (wrong) {
            set_error("rank %d: ping of %zu bytes: %u words did not come from the neighbour and direction they should have come from",
                      s->rank, bytes, wrong);
            rc = SPH_E_STATE;
        }
    }
    s->t_group[4] = acc0;
    if (rc) { s->failed = rc; snprintf(s->fail_msg, sizeof s->fail_msg, "%s", sph_last_error()); }
    return rc;
}

// a stream of this slab, drained with a bound: a neighbour that never takes part must be an error, not a hang
static int slab_wait_stream(sph_slab* s, hipStream_t st, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t q;
    while ((q = hipStreamQuery(st)) == hipErrorNotReady) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > s->wait_timeout_s) {
            set_error("rank %d: %s: no answer from a neighbour after %.0f s", s->rank, what, s->wait_timeout_s);
            s->transport_dead = true;
            if (s->tr.abort) s->tr.abort(s->tr.self);
            return SPH_E_DEVICE;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    if (q != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(q)); return SPH_E_DEVICE; }
    return SPH_OK;
}

// Re-cut: this slab takes over the cell layers [new_z_lo, new_z_hi).  Everything stays on the device and the context is
// kept: the owned particles are sorted once more (so that order and positions agree), split by their TRUE layer into
// "goes down" | "stays" | "goes up" (stable), the two leaving runs travel to rank - 1 / rank + 1 in chunks of the halo
// buffers (one small message tells the counts first), what arrives is put in front of / behind what stays -- exactly
// the order in which the whole-domain sorted array lists the new owner's particles -- and the context is re-keyed for
// its new layers; the next step starts with a full (stable) radix sort, as after an upload.  COLLECTIVE over the
// chain; every particle may move ONE rank per call: the caller keeps new cut r within [old cut r-1, old cut r+1].
static int slab_recut_body(sph_slab* s, uint32_t new_lo, uint32_t new_hi) {
    sph_ctx* c = s->c;
    int rc;
    const size_t rec = 2 * sizeof(float4);
    SPH_HIP(hipStreamSynchronize(s->comm));
    rc = step_hash(c); if (rc) return rc;
    rc = step_sort(c); if (rc) return rc;                 // (a slab's sort also builds the table of its owned slots)
    rc = launch_cells_clear(c); if (rc) return rc;        // no entry of the old numbering survives
    mm_drop_marks(c);
    const uint32_t n = c->n, off = c->own_off, nblk = ceil_div(max(n, 1u), RECUT_BLOCK);
    if (!s->recut_blk) SPH_HIP(hipMalloc((void**)&s->recut_blk, (size_t)2 * (ceil_div(c->cap, RECUT_BLOCK) + 1) * sizeof(uint32_t)));
    s->h_lb[HL_RECUT] = s->h_lb[HL_RECUT + 1] = 0u;
    hipLaunchKernelGGL(k_recut_count, dim3(nblk), dim3(256), 0, c->stream, c->posi + off, n, c->grid, new_lo, new_hi, s->recut_blk);
    hipLaunchKernelGGL(k_recut_scan, dim3(1), dim3(1024), 0, c->stream, s->recut_blk, nblk, s->h_lb_dev + HL_RECUT);
    SPH_HIP(hipGetLastError());
    SPH_HIP(hipStreamSynchronize(c->stream));
    const uint32_t down = s->h_lb[HL_RECUT], up = s->h_lb[HL_RECUT + 1];
    SPH_REQUIRE(down + up <= n, SPH_E_STATE, "rank %d: re-cut counts %u + %u exceed the %u owned particles", s->rank, down, up, n);
    SPH_REQUIRE((s->has_lo || down == 0u) && (s->has_hi || up == 0u), SPH_E_INVALID,
                "rank %d: the new layers [%u, %u) leave %u / %u particles without an owner", s->rank, new_lo, new_hi, down, up);
    const uint32_t keep = n - down - up;
    // [stays | goes down | goes up] in the ping-pong arrays, from the canonical offset
    const uint32_t keep0 = c->gcap, down0 = keep0 + keep, up0 = down0 + down;
    hipLaunchKernelGGL(k_recut_scatter, dim3(nblk), dim3(64), 0, c->stream, c->posi + off, c->velr + off, n, c->grid, new_lo, new_hi,
                       s->recut_blk, c->posi2, c->velr2, keep0, down0, up0);
    // the counts: one record each way (the first record of the migrant buffers)
    hipLaunchKernelGGL(k_recut_counts_out, dim3(1), dim3(64), 0, c->stream, s->mig_send[0], s->mig_send[1], down, up);
    SPH_HIP(hipGetLastError());
    rc = after_main(s); if (rc) return rc;
    rc = slab_exchange(s, SPH_TAG_RECUT_COUNTS, s->mig_send[0], rec, s->mig_recv[0], rec, s->mig_send[1], rec, s->mig_recv[1], rec);
    if (rc) return rc;
    uint32_t in_lo = 0, in_hi = 0;
    if (s->has_lo) SPH_HIP(hipMemcpyAsync((void*)&s->h_lb[HL_RECUT + 2], s->mig_recv[0], sizeof(uint32_t), hipMemcpyDeviceToHost, s->comm));
    if (s->has_hi) SPH_HIP(hipMemcpyAsync((void*)&s->h_lb[HL_RECUT + 3], s->mig_recv[1], sizeof(uint32_t), hipMemcpyDeviceToHost, s->comm));
    rc = slab_wait_stream(s, s->comm, "re-cut (counts)"); if (rc) return rc;
    if (s->has_lo) in_lo = s->h_lb[HL_RECUT + 2];
    if (s->has_hi) in_hi = s->h_lb[HL_RECUT + 3];
    const uint64_t n_new = (uint64_t)keep + in_lo + in_hi;
    // A rank that cannot hold what is coming does NOT simply return: its neighbours have sized the particle rounds from the
    // counts and are about to post them -- they would sit in their bounded waits and abort their transports.  It takes part
    // in every round (its leaving runs go out, what arrives is received and dropped), THEN reports the error and is failed;
    // its neighbours complete their re-cut and hear of the failure in the header of its next migrant message, as after a
    // failed step (slab_fail).
    const bool over = n_new > c->cap;
    // what stays goes to its final place in the (now free) primary arrays: behind what comes from below
    const uint32_t base = c->gcap;
    if (keep && !over)
        hipLaunchKernelGGL(k_recut_copy, dim3(ceil_div(keep, 256u)), dim3(256), 0, c->stream, c->posi2, c->velr2, keep0, c->posi, c->velr,
                           base + in_lo, keep);
    // the two leaving runs, chunk by chunk; both ends of a link know both counts, so they agree on every size
    const uint32_t chunk = s->gcap;
    const uint32_t rounds = max(max(ceil_div(down, chunk), ceil_div(up, chunk)), max(ceil_div(in_lo, chunk), ceil_div(in_hi, chunk)));
    for (uint32_t j = 0; j < rounds; j++) {
        auto part = [&](uint32_t total) { return total > j * chunk ? min(chunk, total - j * chunk) : 0u; };
        const uint32_t s_lo = part(down), s_hi = part(up), r_lo = part(in_lo), r_hi = part(in_hi);
        if (s_lo + s_hi)
            hipLaunchKernelGGL(k_recut_pack, dim3(ceil_div(s_lo + s_hi, 256u)), dim3(256), 0, c->stream, c->posi2, c->velr2, down0 + j * chunk,
                               s_lo, up0 + j * chunk, s_hi, s->halo_send[0], s->halo_send[1]);
        SPH_HIP(hipGetLastError());
        rc = after_main(s); if (rc) return rc;
        rc = slab_exchange(s, SPH_TAG_RECUT, s->halo_send[0], s_lo * rec, s->halo_recv[0], r_lo * rec, s->halo_send[1], s_hi * rec,
                           s->halo_recv[1], r_hi * rec);
        if (rc) return rc;
        if (r_lo + r_hi && !over)
            hipLaunchKernelGGL(k_recut_unpack, dim3(ceil_div(r_lo + r_hi, 256u)), dim3(256), 0, s->comm, s->halo_recv[0], r_lo,
                               base + j * chunk, s->halo_recv[1], r_hi, base + in_lo + keep + j * chunk, c->posi, c->velr);
        SPH_HIP(hipGetLastError());
        rc = after_comm(s); if (rc) return rc;            // the send buffers are free again, the arrivals are in
    }
    rc = slab_wait_stream(s, s->comm, "re-cut (particles)"); if (rc) return rc;
    SPH_HIP(hipStreamSynchronize(c->stream));
    SPH_REQUIRE(!over, SPH_E_CAPACITY, "rank %d: the layers [%u, %u) hold %llu particles, the capacity is %u (the neighbours' re-cut "
                "went through; this rank's particles are lost: the slab is failed)", s->rank, new_lo, new_hi, (unsigned long long)n_new, c->cap);
    c->own_off = base;
    c->n = (uint32_t)n_new;
    rc = set_slab_range(c, new_lo, new_hi); if (rc) return rc;
    s->one_ready = false;                                   // (the layers changed: the next step learns the message sizes again)
    s->recuts++;
    s->recut_moved += down + up;
    return SPH_OK;
}

int sph_slab_recut(sph_slab* s, uint32_t new_z_lo, uint32_t new_z_hi) {
    SPH_REQUIRE(s, SPH_E_INVALID, "null slab");
    if (s->failed) { set_error("%s", s->fail_msg); return s->failed; }
    SPH_REQUIRE(new_z_lo < new_z_hi && new_z_hi <= s->c->params.grid[2], SPH_E_INVALID, "bad layer range [%u, %u)", new_z_lo, new_z_hi);
    SPH_REQUIRE(s->world == 1 || new_z_hi - new_z_lo >= 2, SPH_E_INVALID, "a slab needs at least two cell layers");
    SPH_REQUIRE(s->world == 1 || s->protocol != 1 || new_z_hi - new_z_lo >= 4, SPH_E_INVALID,
                "the one-message step needs slabs of at least four cell layers (asked for [%u, %u))", new_z_lo, new_z_hi);
    SPH_REQUIRE((s->has_lo || new_z_lo == 0u) && (s->has_hi || new_z_hi == s->c->params.grid[2]), SPH_E_INVALID,
                "rank %d of %d: the outer slabs reach to the ends of the grid", s->rank, s->world);
    // a particle moves ONE rank per call: a new range that does not even touch the old one would hand particles to a
    // neighbour that does not own their layer either (the full single-hop rule -- new cut r within [old cut r-1, old cut
    // r+1] -- needs the neighbours' old cuts: slab.py single_hop_cuts; this is the part one rank can see)
    SPH_REQUIRE(new_z_lo <= s->c->z_hi && new_z_hi >= s->c->z_lo, SPH_E_INVALID,
                "rank %d: the new layers [%u, %u) do not touch the old ones [%u, %u): move the cuts one hop at a time", s->rank, new_z_lo,
                new_z_hi, s->c->z_lo, s->c->z_hi);
    SPH_HIP(hipSetDevice(s->c->device));
    const uint64_t exchanges0 = s->exchanges;
    const bool timed0 = s->time_groups;
    s->time_groups = false;
    s->pg = sph_slab::Progress();
    const int rc = slab_recut_body(s, new_z_lo, new_z_hi);
    s->time_groups = timed0;
    s->exchanges = exchanges0;                            // (the step counters count the steps' messages)
    if (rc) {                                             // a half-done re-cut cannot be resumed: the slab is failed
        s->failed = rc;
        snprintf(s->fail_msg, sizeof s->fail_msg, "%s", sph_last_error());
        if (s->transport_dead && s->tr.abort) s->tr.abort(s->tr.self);
    }
    return rc;
}

int sph_slab_recut_stats(const sph_slab* s, uint64_t out[2]) {
    SPH_REQUIRE(s && out, SPH_E_INVALID, "null argument");
    out[0] = s->recuts; out[1] = s->recut_moved;
    return SPH_OK;
}

// test hook: raise one of the sticky device-side error words by hand (what k_slab_insert / k_slab_unpack do when an
// arrival is where it cannot be), so that the failure path of a step that has not sent anything yet can be exercised
int sph_slab_test_raise_flag(sph_slab* s, int flag) {
    SPH_REQUIRE(s && flag >= 0 && flag < 2, SPH_E_INVALID, "bad argument");
    s->h_lb[HL_ERR + flag] = 1u;
    return SPH_OK;
}

int sph_rccl_transport_info(const sph_transport* t, int out[4]) {
    SPH_REQUIRE(t && t->self && t->exchange == rccl_exchange && out, SPH_E_INVALID, "not an RCCL transport");
    const RcclLink* L = (const RcclLink*)t->self;
    out[0] = out[1] = out[2] = -1; out[3] = 0;
    SPH_REQUIRE(L->comm && !L->aborted, SPH_E_DEVICE, "the RCCL communicator of rank %d was aborted", L->rank);
    if (g_rccl.CommCount) SPH_NCCL(g_rccl.CommCount(L->comm, &out[0]));
    if (g_rccl.CommUserRank) SPH_NCCL(g_rccl.CommUserRank(L->comm, &out[1]));
    if (g_rccl.CommCuDevice) SPH_NCCL(g_rccl.CommCuDevice(L->comm, &out[2]));
    if (g_rccl.CommGetAsyncError) SPH_NCCL(g_rccl.CommGetAsyncError(L->comm, &out[3]));
    return SPH_OK;
}

uint64_t sph_slab_in_place_merges(const sph_slab* s) { return s ? s->inserts : 0; }

int sph_slab_counters(const sph_slab* s, uint64_t out[8]) {
    SPH_REQUIRE(s && out, SPH_E_INVALID, "null argument");
    out[0] = s->steps; out[1] = s->migrants; out[2] = s->resorts; out[3] = s->ghosts; out[4] = s->host_waits;
    out[5] = s->inserts; out[6] = s->far_steps; out[7] = s->rest_msgs;
    return SPH_OK;
}

uint64_t sph_slab_exchanges(const sph_slab* s) { return s ? s->exchanges : 0; }

int sph_slab_failed(const sph_slab* s) {
    return s ? s->failed : SPH_E_INVALID;
}

int sph_slab_stats(const sph_slab* s, uint64_t out[5]) {
    SPH_REQUIRE(s && out, SPH_E_INVALID, "null argument");
    out[0] = s->steps; out[1] = s->migrants; out[2] = s->resorts; out[3] = s->ghosts; out[4] = s->host_waits;
    return SPH_OK;
}

}  // extern "C"
