// sph_common.hpp -- shared declarations of libsph_hip.so (host side + device helpers).
// MI355X / gfx950 only: wave64, no portability layer.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/sph_hip.h"

namespace sph {

constexpr int WAVE = 64;
#ifndef SPH_SORT_KPT
#define SPH_SORT_KPT 16
#endif
constexpr uint32_t SORT_TILE_KEYS = 256 * SPH_SORT_KPT;   // keys per radix-sort tile (sph_sort.hip)
#ifndef SPH_PAIR_SMALL_SLOTS_DEFAULT
#define SPH_PAIR_SMALL_SLOTS_DEFAULT 524288u
#endif
constexpr uint32_t MM_TILE_CHUNKS = 256;   // merge sort: 64-slot chunks per scan tile (sph_sort.hip)

// Grid description passed by value to kernels (replaces the device-resident SimParams*
// every reference kernel dereferences, particleSystem.cu:93-103,127-130).
struct GridDesc {
    float box_min[3];
    float box_dims[3];     // box_max - box_min
    float inv_dims[3];     // 1 / box_dims where that is a power of two (the division is then an exact scaling), else 0
    uint32_t g[3];         // global cells per axis
    float gf[3];           // (float)g
    int32_t z_off;         // local z layer = global z layer - z_off (slab: z_lo - ghost layers; whole domain: 0)
    uint32_t zl;           // local z layers (slab: owned + 2 x ghost layers; whole domain: g[2])
    uint32_t ncells;       // g[0]*g[1]*zl
};

struct Phys {
    float h, h2;
    float mass;
    float rest_density, gas_constant;
    float poly6_mass;      // MASS * 315/(65*pi*h^9)        (particleSystem.cu:30,35)
    float spiky_half_mass; // MASS * 45/(pi*h^6) / 2         (particleSystem.cu:41,47; sign folded)
    float visc_coef;       // VISC * MASS * 45/(pi*h^6)      (particleSystem.cu:42,48)
    float cp_scale;        // spiky_half_mass / visc_coef: the pressure coefficient relative to the viscous one (sph_pairs.hip)
    float gravity_y;
    float wall_eps, wall_damping;
    float coll_dist2;      // (COLLISION_PARAM * 2 * radius)^2 (particleSystem.cu:61)
    float coll_mass;       // MASS * (1 + RESTITUTION)        (particleSystem.cu:62)
    float box_min[3], box_max[3];
};

}  // namespace sph

// The opaque context of include/sph_hip.h.
struct sph_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    sph_params params{};
    int precision = SPH_PRECISION_F32;   // sph_set_precision
    sph::GridDesc grid{};
    sph::Phys phys{};

    uint32_t cap = 0;        // owned capacity
    uint32_t gcap = 0;       // ghost capacity per side (0: whole-domain context)
    uint32_t tot = 0;        // gcap + cap + gcap
    bool slab = false;
    uint32_t z_lo = 0, z_hi = 0;   // owned global cell layers
    // ghost cell layers a slab context keeps on either side of its owned layers: 1 (the three-message step: the neighbours'
    // boundary layers, whose densities arrive in a message of their own) or 2 (the one-message step: two layers of the
    // neighbour's particles, the densities of the inner one recomputed here) -- sph_create_slab_layers
    uint32_t ghost_layers = 1;

    // counts
    uint32_t n = 0;          // owned particles
    uint32_t own_off = 0;    // offset of the owned range inside posi/velr (gcap after a sort)
    uint32_t n_glo = 0, n_ghi = 0;   // ghosts installed below / above
    uint32_t halo_n[2] = {0, 0};     // boundary-layer counts of the last halo pack
    bool halo_n_valid = false;

    // sorted SoA state, `tot` entries each; the owned range starts at own_off
    float4* posi = nullptr;   // x, y, z, creation index (bits)
    float4* velr = nullptr;   // vx, vy, vz, unused
    float4* posi2 = nullptr;  // ping-pong targets of reorder / integrate
    float4* velr2 = nullptr;
    uint32_t* keyS = nullptr; // cell key per slot (same indexing as posi)
    uint32_t* keyS2 = nullptr;// ping-pong target of the sort
    float2* dp = nullptr;     // density, pressure
    float2* cw = nullptr;     // what the force pass needs of a NEIGHBOUR: cp_j = cp_scale * p_j, w_j = visc_coef / rho_j (0 where
                              // rho_j = 0: padding), written by the density pass next to dp (neighbour_terms, sph_device.hpp)
    float4* fpress = nullptr; // phase API outputs
    float4* fvisc = nullptr;
    float4* dvel = nullptr;   // delta_velocity xyz, collision count
    float4* pos_out = nullptr;// (x,y,z,1) by creation index: the gl_pos analogue
    uint32_t pos_out_cap = 0;

    // pair kernels: a (dz, dy) row whose staged hull would exceed this many slots is read straight from global memory
    // by every lane instead (sph_pairs.hip: traverse; sph_set_direct_hull)
    uint32_t direct_hull = 512;
    // pair kernels: a context with fewer owned particles than this launches blocks of 128 threads instead of 256 (sph_pairs.hip:
    // SMALL_THREADS_PAIR; same results bit for bit; sph_set_pair_small_launch)
    uint32_t pair_small_slots = SPH_PAIR_SMALL_SLOTS_DEFAULT;
    // block order of the pair kernels (sph_device.hpp: BlockOrder; sph_set_block_order): every XCD walks a contiguous eighth
    // of the slots; the fused force pass also walks strips of 2^order_strip_sh blocks through the z layers of that eighth
    bool order_xcd = true, order_ztile = true;
    bool order_ztile_dens = true, order_xrot = true;        // (SPH_BLOCK_ORDER fields 4 and 5: the strips for the density pass too; XCD x starts at strip x * strips / 8)
    uint32_t order_strip_sh = 4;

    // cell table: {start, end} per local cell, zero = empty
    uint2* cells = nullptr;        // = cells_base + 1
    uint2* cells_base = nullptr;   // the allocation: ncells + one zero guard entry on either side
    uint32_t cells_alloc = 0;      // cells the allocation holds (a slab whose layer range grows gets a bigger table: set_slab_range)
    uint32_t cells_lo = 0, cells_hi = 0;   // slot range the table was built from
    bool cells_valid = false;
    bool cells_clear_deferred = false;   // sph_hash left the clearing of the old table to the sort (merge path)
    // the slab step builds the table of the owned slots INSIDE its bounds kernel (one dispatch less on every rank's critical
    // path): it sets `owned_cells_in_bounds` around its sort, the sort then leaves the build `pending` instead of launching it
    bool owned_cells_in_bounds = false, owned_cells_pending = false;
    // ... and the clearing of the OLD ghosts' cells (a kernel of its own at hash time: k_cells_clear2, ~6 us at the head of every
    // rank's step) to spare blocks of the sort's k_mm_compact: the slab step sets `defer_ghost_clear` around its hash, the hash
    // then only notes the two ranges, and launch_sort clears them -- in k_mm_compact when it merges, else with the kernel
    bool defer_ghost_clear = false, ghost_clear_pending = false;
    uint32_t ghost_clear[4] = {0, 0, 0, 0};       // slot ranges [0],[1]) and [2],[3])
    bool keys_fresh = false;   // k0 already holds the keys of the current positions (written by the integrate epilogue)

    // radix sort scratch
    uint32_t* k0 = nullptr; uint32_t* v0 = nullptr;
    uint32_t* k1 = nullptr; uint32_t* v1 = nullptr;
    uint32_t* os_hist = nullptr;    // [group][4 passes][512 digits]: digit counts per group of 16 tiles
    uint32_t* os_base = nullptr;    // same shape: first output position of a (group, digit)
    uint32_t* os_tickets = nullptr; // one per pass (re-armed by k_os_scan)
    uint32_t* os_tot = nullptr;     // [4 passes][512 digits]: keys per digit
    uint32_t os_groups_cap = 0;
    unsigned long long* os_status = nullptr;   // (tile, digit) look-back words {epoch << 1 | is_prefix, count} (one-group sorts)
    uint32_t* os_status32 = nullptr;           // (tile, digit) words {epoch:19, count:13} (grouped sorts)
    uint32_t os_epoch = 0;          // changes with every pass of every sort: the status table is never cleared
    uint32_t* os_err_host = nullptr;           // pinned, mapped: set by a look-back that timed out
    uint32_t* os_err_dev = nullptr;
    uint32_t sort_blocks_cap = 0;
    uint32_t key_bits = 0;
    const uint32_t* last_perm = nullptr;   // v0 or v1: the permutation of the last sort; null = identity / not kept
    bool keep_perm = false;         // the merge path also writes the permutation (only the compat seam needs it)
    // the sort as a merge (sph_sort.hip: launch_sort_merge)
    bool sort_merge = true;         // SPH_SORT_MERGE=0 in the environment at create time turns it off
    bool sort_merge_always = false; // sph_set_sort_mode(c, 2): merge whatever the mover count (tests)
    bool order_valid = false;       // [own_off, own_off+n) is still in the order of the last sort, keyS = its keys
    bool last_sort_skipped = false;
    // A slab context is stepped by sph_slab_step, whose host waits for the device once per step: the host cannot run
    // ahead, so its run-ahead needs no bound.
    bool host_paced = false;
    // What the host learns from the device about the sorts WITHOUT events (every event recorded on a stream costs the device
    // ~5.5 us of idle at its next dispatch; rounds 1-5 recorded two per step in a whole-domain context: 11 us of a 92 us step
    // at 131,072 particles, profiles/r06base_headless_n131072_kernel_stats.csv): the kernels echo sequence numbers into the
    // mapped host block mm_count_host -- [3] the number of the last sort whose table build has started (bounds the host's
    // run-ahead to four sorts), [4] the number of the last mover count, stored AFTER the count [0] (the skip of a sort with
    // nothing to do needs the count of the CURRENT marks: it is looked at, never waited for).
    uint32_t sort_seq_issued = 0;   // sorts that queued a table build so far
    uint32_t scan_seq_issued = 0;   // mover counts queued so far (k_mm_tilescan; k_mm_compact when it counts itself)
    uint32_t cells_seq_next = 0;    // what the next whole-range table build echoes into mm_count_host[3] (0: nothing)
    bool mm_counted_valid = false;  // the last count queued (scan_seq_issued) is that of the CURRENT marks
    uint64_t sort_merges = 0, sort_calls = 0, sort_skips = 0;   // skips: merges with no mover at all (nothing done)
    // A whole-domain context picks ONE form of the movers' sort from a count up to four steps old (radix_sort_bits).  A flow
    // changes that count slowly; the caller can change it at once (new positions or velocities for everybody: an upload, a
    // kick through sph_set_by_index, another box): for the sorts up to this call number both forms are launched again, or
    // a count that went from 0 to millions would go through the one-block sort's tile-by-tile fallback (~8 ms per sort at
    // 2 M movers, four times).
    uint64_t sort_form_both_until = 0;
    uint64_t sort_forms[3] = {0, 0, 0};   // movers' sorts launched as: both forms / the one-block sort alone / the multi-block passes alone
    uint64_t* mm_mask = nullptr;    // one bit per slot: key changed since the last sort
    uint32_t* mm_M64 = nullptr;     // movers before each 64-slot chunk
    uint32_t* mm_tile_cnt = nullptr; uint32_t* mm_tile_off = nullptr;
    uint32_t* mm_k0 = nullptr; uint32_t* mm_k1 = nullptr; uint32_t* mm_v1 = nullptr;   // mover (key, slot) ping-pong (+ v0)
    uint32_t* mm_count = nullptr;           // movers of the current sort (device)
    uint32_t* mm_count_host = nullptr;      // pinned, written by the device: [0] last known count (a hint), [1], [2] the keys of
                                            // the first / last owned slot after the last table build (block_order's estimate),
                                            // [3] / [4] sequence numbers of the last table build / the last count (above); 8 words
    uint32_t* mm_count_host_dev = nullptr;  // device view of the same word
    unsigned long long* mm_total = nullptr; // movers of all sorts so far (device; sph_sort_stats)
    // the fused integrate epilogue already wrote mm_mask / mm_tile_cnt for the range it was launched on
    bool mm_marked = false;
    bool mm_scanned = false;        // ... and the scan that counts them is already queued (mm_scan_marks)
    uint32_t* mm_tileL = nullptr;   // coarse mover ranks per 4096 slots (k_mm_tile_rank)
    uint32_t* mm_tileA = nullptr;   // first key of every 4096-slot tile of the old order (k_mm_tile_rank)
    uint32_t mm_marked_off = 0, mm_marked_n = 0;
    uint32_t* d_scratch = nullptr;  // small device scratch (counts)
    uint32_t* h_scratch = nullptr;  // pinned host mirror

    // state machine for the phase API
    enum Stage { ST_LOADED = 0, ST_HASHED, ST_SORTED, ST_CELLS } stage = ST_LOADED;
    bool have_force = false, have_coll = false, have_dens = false;

    // device timing
    bool timing = false;
    std::vector<hipEvent_t> events;   // triples: start event, stop event, phase id (see PhaseTimer)
    double ph_ms[SPH_PH_COUNT] = {0};
    uint32_t timed_steps = 0;
};

namespace sph {

void set_error(const char* fmt, ...);
void mm_drop_marks(sph_ctx* c);     // sph_sort.hip
void mm_scan_marks(sph_ctx* c);     // sph_sort.hip
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SPH_HIP(call)                                                        \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return sph::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define SPH_REQUIRE(cond, code, ...)          \
    do {                                      \
        if (!(cond)) {                        \
            sph::set_error(__VA_ARGS__);      \
            return (code);                    \
        }                                     \
    } while (0)

// kernel launchers (defined in the .hip files)
int launch_hash(sph_ctx* c);
int launch_reset_lattice(sph_ctx* c, const uint32_t lattice[3], int jitter, const float jitter_dims[3], uint64_t start,
                         uint32_t count);
int launch_sort(sph_ctx* c);          // radix sort of (k0,v0)[0,n) + reorder into posi2/velr2/keyS
int launch_merge_arrivals(sph_ctx* c, uint32_t n_in, uint32_t n_front = 0);   // particles appended behind the sorted range join it
int launch_cells_clear(sph_ctx* c);
int launch_cells_clear_range(sph_ctx* c, uint32_t lo, uint32_t hi);
int launch_cells_build_range(sph_ctx* c, uint32_t lo, uint32_t hi);
int launch_cells_clear_2ranges(sph_ctx* c, uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1);
int launch_cells_build_2ranges(sph_ctx* c, uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1);
int launch_cells_build(sph_ctx* c);
constexpr uint32_t CELLS_SPT = 4;                     // slots per thread of the table build (sph_device.hpp: cells_build_thread)
inline uint32_t cells_build_blocks(uint32_t count) { return (count + 256u * CELLS_SPT - 1u) / (256u * CELLS_SPT); }
int launch_density(sph_ctx* c);
int launch_force(sph_ctx* c, bool force, bool collide, bool integrate, float dt);
// sub-range forms for the slab driver (interior first, boundary layers once the ghosts are in)
int launch_density_range(sph_ctx* c, uint32_t lo, uint32_t hi);
int launch_density_hole(sph_ctx* c, uint32_t lo, uint32_t hi, uint32_t hole_lo, uint32_t hole_hi);   // [lo, hi) minus the hole
int launch_density_dev_range(sph_ctx* c, const uint32_t* range_dev, uint32_t max_count);            // range in device memory
int launch_force_dev_range(sph_ctx* c, const uint32_t* range_dev, uint32_t max_count, float dt);   // fused pass, range in device memory
int launch_force_hole(sph_ctx* c, uint32_t lo, uint32_t hi, uint32_t hole_lo, uint32_t hole_hi, bool force, bool collide,
                      bool integrate, float dt, bool mark);
bool force_begin(sph_ctx* c, bool integrate);
int launch_force_range(sph_ctx* c, uint32_t lo, uint32_t hi, bool force, bool collide, bool integrate, float dt, bool mark);
void force_finish(sph_ctx* c, bool integrate, bool mark);
// phase bodies of sph_capi.hip (with their bookkeeping), for the slab driver
int set_slab_range(sph_ctx* c, uint32_t z_lo, uint32_t z_hi);   // sph_capi.hip: a slab context takes over another layer range (its table must be clear)
int step_hash(sph_ctx* c);
int step_sort(sph_ctx* c);
int step_cells(sph_ctx* c);
int launch_integrate(sph_ctx* c, float dt);

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// ---- device timing: a pair of HIP events on the context's stream around a phase (only while sph_timing_enable) ----
struct PhaseTimer {
    sph_ctx* c; int phase; hipEvent_t a = nullptr, b = nullptr;
    PhaseTimer(sph_ctx* c_, int ph) : c(c_), phase(ph) {
        if (!c->timing) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        hipEventRecord(a, c->stream);
    }
    ~PhaseTimer() {
        if (!a || !b) return;
        hipEventRecord(b, c->stream);
        c->events.push_back(a);
        c->events.push_back(b);
        c->events.push_back((hipEvent_t)(intptr_t)phase);   // tag
    }
};
void timing_collect(sph_ctx* c);      // sph_capi.hip: waits for the recorded events and adds them up per phase

}  // namespace sph
