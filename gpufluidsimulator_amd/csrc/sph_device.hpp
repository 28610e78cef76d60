// sph_device.hpp -- device-side helpers (gfx950, wave64).
#pragma once

#include "sph_common.hpp"

namespace sph {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// ---- which slots a workgroup of the pair kernels takes ---------------------------------------------------------------
// Workgroups are dealt round-robin to the 8 XCDs (block b runs on XCD b % 8), each with its own 4 MB L2, and an XCD
// starts its blocks in the order of their ids.  A block stages rows of three z layers; with blocks walking the sorted
// particles front to back on all XCDs at once, a row is fetched through the fabric again for every layer that uses it and
// by every XCD: 3.1x the algorithmic bytes for k_force at C3 (profiles/pmc_traffic.json, rounds 2-4).  Two remappings of
// the block id, both bijections on [0, nb), so that only the ORDER of the work changes (results do not depend on it):
//   xcd   every XCD walks ONE contiguous eighth of the slots: FETCH_SIZE of k_force -24 %, of k_density -19 %, SQ cycles
//         -1.2 % / +0.3 % (round 1 had measured "no gain" in time; the traffic it had not looked at);
//   zt    inside that eighth, strips of 2^s_sh blocks are walked through all 2^nl_sh z layers of the eighth (2^lb_sh
//         blocks per layer, from the host's estimate) before the next strip: the rows of layers z-1, z, z+1 of a strip
//         are still in the XCD's L2 when the next layer's blocks of the same strip come by.  k_force FETCH_SIZE -51 %
//         against the plain order (1.8x algorithmic traffic instead of 3.1x) and -3.6 % wall clock; k_density fetches
//         -55..63 % but ran 2.8 % SLOWER until the XCDs stopped walking the SAME strip at the same time (`xrot`: XCD x
//         starts at strip x * strips / 8 -- eight XCDs x sixteen layers at one offset modulo the 2 MB layer stride had
//         been hitting the same memory channels): with it the time is that of the plain order.
//         profiles/r04_block_order_traffic.txt, r04_block_order_wallclock_ab.txt.
// Powers of two only (shifts, no divisions per wave): the estimate need not be exact, a tile that straddles two layers
// only loses some of the reuse.
struct BlockOrder {
    uint32_t xcd;                   // 1: contiguous eighth per XCD
    uint32_t lb_sh, s_sh, nl_sh;    // zt: log2 of blocks per layer, per strip, layers per XCD range; nl_sh = 0: off
    uint32_t xrot;                  // zt: XCD x starts at strip x * (strips / 8) instead of strip 0
};

__device__ __forceinline__ uint32_t ordered_block(uint32_t b, uint32_t nb, const BlockOrder& o) {
    if (!o.xcd) return b;
    const uint32_t q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    const uint32_t base = xcd * q + (xcd < r ? xcd : r);
    uint32_t j = b >> 3;
    if (o.nl_sh && j < (1u << (o.lb_sh + o.nl_sh))) {               // inside the whole layers of this XCD's range
        const uint32_t per_sh = o.s_sh + o.nl_sh;                    // blocks per strip over all layers
        uint32_t st = j >> per_sh;
        const uint32_t rr = j & ((1u << per_sh) - 1u);
        if (o.xrot) { const uint32_t ns_sh = o.lb_sh - o.s_sh; st = (st + ((xcd << ns_sh) >> 3)) & ((1u << ns_sh) - 1u); }
        const uint32_t l = rr >> o.s_sh, col = (st << o.s_sh) + (rr & ((1u << o.s_sh) - 1u));
        j = (l << o.lb_sh) + col;
    }
    return base + j;
}

// ---- wave64 reductions over DPP (no LDS): row reductions by quad_perm / mirrors, then
//      row_bcast15 / row_bcast31 carry the partial results to lane 63 (gfx9 DPP controls).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// min and max of one value over the wave.  The per-lane range lengths these are used on are equal in
// all 64 lanes more often than not (every cell of a resting lattice holds the same count), and that
// case costs a readfirstlane, a compare and a scalar test instead of two 7-step DPP reductions.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v);
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v);
__device__ __forceinline__ void wave_min_max_u32(uint32_t v, uint32_t& vmin, uint32_t& vmax) {
    const uint32_t v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    if (__ballot(v != v0) == 0ull) { vmin = v0; vmax = v0; return; }     // wave-uniform branch
    vmin = wave_min_u32(v);
    vmax = wave_max_u32(v);
}

__device__ __forceinline__ uint32_t wave_min_u32_uniform_first(uint32_t v) {
    const uint32_t v0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    if (__ballot(v != v0) == 0ull) return v0;                             // wave-uniform branch
    return wave_min_u32(v);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_u32<0xB1>(0xFFFFFFFFu, v));        // quad_perm [1,0,3,2]
    v = min(v, dpp_u32<0x4E>(0xFFFFFFFFu, v));        // quad_perm [2,3,0,1]
    v = min(v, dpp_u32<0x141>(0xFFFFFFFFu, v));       // row_half_mirror
    v = min(v, dpp_u32<0x140>(0xFFFFFFFFu, v));       // row_mirror
    v = min(v, dpp_u32<0x142, 0xA>(0xFFFFFFFFu, v));  // row_bcast15 -> rows 1,3
    v = min(v, dpp_u32<0x143, 0xC>(0xFFFFFFFFu, v));  // row_bcast31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, dpp_u32<0xB1>(0u, v));
    v = max(v, dpp_u32<0x4E>(0u, v));
    v = max(v, dpp_u32<0x141>(0u, v));
    v = max(v, dpp_u32<0x140>(0u, v));
    v = max(v, dpp_u32<0x142, 0xA>(0u, v));
    v = max(v, dpp_u32<0x143, 0xC>(0u, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- cell table: {start, end} per occupied cell from boundary flags on the sorted keys (k_cells_build, sph_pairs.hip; the slab
//      step's bounds kernel runs the same build in its spare blocks, sph_slab.hip) ------------------------------------------------
// Thread t of the build takes CELLS_SPT = 4 consecutive slots from lo + 4 t: six key reads for four slots instead of twelve, a
// quarter of the threads (one slot per thread took 51 us for 16.7 M slots: far from any bandwidth).  No atomics: the first /
// last slot of a cell's run writes .x / .y.  ends_host (the whole owned range only): the first and the last key, and the sort's
// sequence number, into mapped host memory (block_order's estimate; launch_sort's throttle).
__device__ __forceinline__ void cells_build_thread(const uint32_t* __restrict__ key, uint32_t lo, uint32_t hi, uint2* __restrict__ cells,
                                                   volatile uint32_t* __restrict__ ends_host, uint32_t seq, uint32_t t) {
    const uint32_t s0 = lo + t * CELLS_SPT;
    if (s0 >= hi) return;
    if (ends_host && s0 == lo) {
        ends_host[0] = key[lo]; ends_host[1] = key[hi - 1u];
        if (seq) ends_host[2] = seq;
    }
    uint32_t kk[CELLS_SPT + 2];                          // key[s0 - 1 .. s0 + 4]
    kk[1] = key[s0];
    kk[0] = s0 > lo ? key[s0 - 1] : ~kk[1];
#pragma unroll
    for (uint32_t j = 1; j <= CELLS_SPT; j++) kk[j + 1] = s0 + j < hi ? key[s0 + j] : ~kk[j];
#pragma unroll
    for (uint32_t j = 0; j < CELLS_SPT; j++) {
        const uint32_t s = s0 + j;
        if (s >= hi) break;
        const uint32_t k = kk[j + 1];
        if (kk[j] != k) cells[k].x = s;
        if (kk[j + 2] != k) cells[k].y = s + 1;
    }
}

// ---- what the force pass needs of a neighbour j (sph_pairs.hip: k_force) --------------------------------------------
// cp_j = (spiky/visc) * p_j and w_j = visc * 1/rho_j (v_rcp_f32), so that the pair loop needs w = w_j * (h - r) for the
// viscosity and (cp_i + cp_j) * w * (h - r) / r for the pressure.  rho = 0 (padding entries) -> weight 0.  Computed ONCE
// per particle, by whoever writes its (rho, p) -- the density pass, the unpacking of a ghost layer's densities --
// instead of at every staging of every wave that has the particle among its candidates (9 x 2 per wave in k_force).
__device__ __forceinline__ float2 neighbour_terms(const Phys& ph, float rho, float p) {
    return make_float2(ph.cp_scale * p, rho > 0.f ? ph.visc_coef * __builtin_amdgcn_rcpf(rho) : 0.f);
}

// ---- cell hashing -------------------------------------------------------------------------
// get_Z_index of the reference (particleSystem.cu:93-103): subtract boxMin, DIVIDE by the box
// dimension, THEN multiply by the grid size, floor.  IEEE division (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt) so that the cell coordinates are bit-identical to the
// CPU.  Unlike the reference, out-of-box positions are clamped into the grid instead of producing
// out-of-range Morton codes.
__device__ __forceinline__ uint32_t cell_coord(float p, float bmin, float bdim, float gf, uint32_t g) {
    float rel = p - bmin;
    float q = (rel / bdim) * gf;
    int c = (int)floorf(q);
    c = c < 0 ? 0 : c;
    c = c >= (int)g ? (int)g - 1 : c;
    return (uint32_t)c;
}

// The same with the division replaced by an exact scaling where the box edge is a power of two (inv != 0: GridDesc::
// inv_dims): x / 2^k and x * 2^-k are the same float, and an IEEE division is ~10 instructions (three of them in the
// integrate epilogue of every particle).  Other edges keep the division.
__device__ __forceinline__ uint32_t cell_coord(float p, float bmin, float bdim, float inv, float gf, uint32_t g) {
    const float rel = p - bmin;
    const float q = (inv != 0.f ? rel * inv : rel / bdim) * gf;              // (wave-uniform choice)
    int c = (int)floorf(q);
    c = c < 0 ? 0 : c;
    c = c >= (int)g ? (int)g - 1 : c;
    return (uint32_t)c;
}

// Local cell key: x fastest, then y, then the local z layer (global z - z_off; in a slab context
// layer 0 and layer zl-1 are the ghost layers, a whole-domain context has none).  The reference numbers cells in Morton order
// (particleSystem.cu:68-91); the numbering is internal: results are reported by creation index.
__device__ __forceinline__ uint32_t cell_key(const GridDesc& g, float x, float y, float z) {
    uint32_t cx = cell_coord(x, g.box_min[0], g.box_dims[0], g.inv_dims[0], g.gf[0], g.g[0]);
    uint32_t cy = cell_coord(y, g.box_min[1], g.box_dims[1], g.inv_dims[1], g.gf[1], g.g[1]);
    uint32_t cz = cell_coord(z, g.box_min[2], g.box_dims[2], g.inv_dims[2], g.gf[2], g.g[2]);
    // particles outside the local layers cannot be represented: clamp into the outermost ones
    int lz = (int)cz - g.z_off;
    lz = lz < 0 ? 0 : lz;
    lz = lz >= (int)g.zl ? (int)g.zl - 1 : lz;
    return ((uint32_t)lz * g.g[1] + cy) * g.g[0] + cx;
}

}  // namespace sph
