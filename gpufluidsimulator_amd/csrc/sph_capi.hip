// sph_capi.hip -- the C ABI of include/sph_hip.h: context lifetime, state transfer, phase
// sequencing and device timing.  Host code only; kernels live in sph_sort.hip / sph_pairs.hip /
// sph_halo.hip.
#include "sph_common.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <new>

namespace sph {

static thread_local std::string g_err;
static int g_device = 0;        // sph_select_device: what `device < 0` means in sph_create

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return SPH_E_DEVICE;
}

static uint32_t next_pow2(uint32_t x) {   // nextPow2, SPH/particleSystem.h:34-43
    x--; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; x++;
    return x;
}

static int derive(sph_ctx* c, const sph_params* p, uint32_t z_lo, uint32_t z_hi, bool slab) {
    for (int a = 0; a < 3; a++) {
        SPH_REQUIRE(p->grid[a] >= 1 && p->grid[a] <= 4096, SPH_E_INVALID, "grid[%d] = %u out of range", a, p->grid[a]);
        SPH_REQUIRE(p->box_max[a] > p->box_min[a], SPH_E_INVALID, "empty box on axis %d", a);
    }
    SPH_REQUIRE(p->h > 0.f && p->mass > 0.f && p->particle_radius > 0.f, SPH_E_INVALID, "h, mass, radius must be > 0");
    SPH_REQUIRE(z_lo < z_hi && z_hi <= p->grid[2], SPH_E_INVALID, "bad slab [%u, %u) of %u layers", z_lo, z_hi, p->grid[2]);
    c->params = *p;
    GridDesc& g = c->grid;
    for (int a = 0; a < 3; a++) {
        g.box_min[a] = p->box_min[a];
        g.box_dims[a] = p->box_max[a] - p->box_min[a];
        {   // a power-of-two edge (every BASELINE config: 4, 8, 32, 64): x / edge == x * (1 / edge) in every bit
            int e = 0;
            const float m = frexpf(g.box_dims[a], &e);
            g.inv_dims[a] = (m == 0.5f && e > -100 && e < 100) ? 1.0f / g.box_dims[a] : 0.0f;
        }
        g.g[a] = p->grid[a];
        g.gf[a] = (float)p->grid[a];
    }
    c->z_lo = z_lo; c->z_hi = z_hi;
    const uint32_t G = c->ghost_layers;
    SPH_REQUIRE(G == 1u || G == 2u, SPH_E_INVALID, "a slab keeps 1 or 2 ghost layers per side, not %u", G);
    g.z_off = slab ? (int32_t)z_lo - (int32_t)G : 0;
    g.zl = slab ? z_hi - z_lo + 2u * G : p->grid[2];
    SPH_REQUIRE(slab || (z_lo == 0 && z_hi == p->grid[2]), SPH_E_INVALID, "a whole-domain context owns every layer");
    uint64_t nc = (uint64_t)g.g[0] * g.g[1] * g.zl;
    SPH_REQUIRE(nc < (1ull << 32), SPH_E_INVALID, "cell table too large (%llu cells)", (unsigned long long)nc);
    g.ncells = (uint32_t)nc;
    c->key_bits = 1;
    while (c->key_bits < 32 && (1ull << c->key_bits) < nc) c->key_bits++;

    // constants exactly as the reference's float expressions evaluate (particleSystem.cu:30,41,42)
    const float PI_F = 3.141592654f;
    Phys& ph = c->phys;
    ph.h = p->h;
    ph.h2 = p->h * p->h;
    ph.mass = p->mass;
    ph.rest_density = p->rest_density;
    ph.gas_constant = p->gas_constant;
    const float poly6 = 315.f / (65.f * PI_F * powf(p->h, 9.f));   // note the 65, as in the reference
    const float spiky = 45.f / (PI_F * powf(p->h, 6.f));
    ph.poly6_mass = p->mass * poly6;
    ph.spiky_half_mass = p->mass * spiky * 0.5f;
    ph.visc_coef = (p->viscosity * p->mass) * spiky;
    ph.cp_scale = ph.spiky_half_mass / ph.visc_coef;
    ph.gravity_y = p->gravity_y;
    ph.wall_eps = p->wall_eps;
    ph.wall_damping = p->wall_damping;
    // computeCollision tests (double)sqrtf(r2) <= COLLISION_PARAM * 2 * radius (particleSystem.cu:61).
    // sqrtf is monotone and correctly rounded, so that is r2 <= the LARGEST float whose square root
    // rounds to a float <= the threshold: found by walking a few ulps around cd^2.
    const double cd = (double)p->collision_param * 2.0 * (double)p->particle_radius;
    float r2max = (float)(cd * cd);
    for (int k = 0; k < 8; k++) r2max = nextafterf(r2max, INFINITY);
    while ((double)sqrtf(r2max) > cd) r2max = nextafterf(r2max, 0.f);
    ph.coll_dist2 = r2max;
    ph.coll_mass = p->mass * (1.f + p->restitution);
    for (int a = 0; a < 3; a++) { ph.box_min[a] = p->box_min[a]; ph.box_max[a] = p->box_max[a]; }
    return SPH_OK;
}

template <class T>
static int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return SPH_E_NOMEM;
    }
    return SPH_OK;
}

static void free_all(sph_ctx* c) {
    hipFree(c->posi); hipFree(c->velr); hipFree(c->posi2); hipFree(c->velr2); hipFree(c->keyS); hipFree(c->dp); hipFree(c->cw);
    hipFree(c->fpress); hipFree(c->fvisc); hipFree(c->dvel); hipFree(c->pos_out); hipFree(c->cells_base);
    hipFree(c->k0); hipFree(c->v0); hipFree(c->k1); hipFree(c->v1); hipFree(c->os_hist); hipFree(c->os_base); hipFree(c->os_tickets); hipFree(c->os_tot); hipFree(c->os_status); hipFree(c->os_status32); hipFree(c->keyS2); hipFree(c->mm_tileL); hipFree(c->mm_tileA);
    if (c->os_err_host) hipHostFree(c->os_err_host);
    hipFree(c->d_scratch);
    hipFree(c->mm_mask); hipFree(c->mm_M64); hipFree(c->mm_tile_cnt); hipFree(c->mm_tile_off);
    hipFree(c->mm_k0); hipFree(c->mm_k1); hipFree(c->mm_v1); hipFree(c->mm_count); hipFree(c->mm_total);
    if (c->mm_count_host) hipHostFree(c->mm_count_host);
    if (c->h_scratch) hipHostFree(c->h_scratch);
}

static int create_impl(sph_ctx** out, int device, uint32_t capacity, const sph_params* p, uint32_t z_lo, uint32_t z_hi,
                       uint32_t gcap, bool slab, uint32_t ghost_layers = 1) {
    SPH_REQUIRE(out && p, SPH_E_INVALID, "null argument");
    *out = nullptr;
    SPH_REQUIRE(capacity > 0 && (uint64_t)capacity + 2ull * gcap < (1ull << 31), SPH_E_INVALID, "bad capacity");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (%s): libsph_hip has no CPU fallback", hipGetErrorString(e));
        return SPH_E_DEVICE;
    }
    if (device < 0) device = g_device;       // the device chosen with sph_select_device (default 0)
    SPH_REQUIRE(device < ndev, SPH_E_DEVICE, "device %d out of range (%d devices)", device, ndev);
    SPH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SPH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libsph_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return SPH_E_DEVICE;
    }
    sph_ctx* c = new (std::nothrow) sph_ctx();
    SPH_REQUIRE(c, SPH_E_NOMEM, "out of host memory");
    c->device = device;
    c->ghost_layers = slab ? ghost_layers : 1u;
    int rc = derive(c, p, z_lo, z_hi, slab);
    if (rc) { delete c; return rc; }
    c->cap = capacity; c->gcap = gcap; c->tot = capacity + 2 * gcap; c->slab = slab;
    c->own_off = gcap;
    c->sort_blocks_cap = ceil_div(capacity, SORT_TILE_KEYS) + 1;
    c->pos_out_cap = slab ? 0 : capacity;
    // + 2*PIECE entries: the pair kernels stage whole 128-entry pieces without bounds predicates
    const size_t tot = (size_t)c->tot + 256;
    rc = dev_alloc(&c->posi, tot);
    if (!rc) rc = dev_alloc(&c->velr, tot);
    if (!rc) rc = dev_alloc(&c->posi2, tot);
    if (!rc) rc = dev_alloc(&c->velr2, tot);
    if (!rc) rc = dev_alloc(&c->keyS, tot);
    if (!rc) rc = dev_alloc(&c->keyS2, tot);
    if (!rc) rc = dev_alloc(&c->dp, tot);
    if (!rc) rc = dev_alloc(&c->cw, tot);
    if (!rc) rc = dev_alloc(&c->fpress, tot);
    if (!rc) rc = dev_alloc(&c->fvisc, tot);
    if (!rc) rc = dev_alloc(&c->dvel, tot);
    if (!rc) rc = dev_alloc(&c->pos_out, (size_t)c->pos_out_cap);
    // one guard entry on either side: the pair kernels read cells[key - 1 .. key + 1] of a row unconditionally
    if (!rc) rc = dev_alloc(&c->cells_base, (size_t)c->grid.ncells + 2);
    if (!rc) { c->cells = c->cells_base + 1; c->cells_alloc = c->grid.ncells; }
    if (!rc) rc = dev_alloc(&c->k0, (size_t)capacity);
    if (!rc) rc = dev_alloc(&c->v0, (size_t)capacity);
    if (!rc) rc = dev_alloc(&c->k1, (size_t)capacity);
    if (!rc) rc = dev_alloc(&c->v1, (size_t)capacity);
    const size_t os_groups = (size_t)c->sort_blocks_cap / 4 + 3;   // enough for groups of >= 4 tiles
    c->os_groups_cap = (uint32_t)os_groups;
    if (!rc) rc = dev_alloc(&c->os_hist, os_groups * 4 * 512);
    if (!rc) rc = dev_alloc(&c->os_base, os_groups * 4 * 512);
    if (!rc) rc = dev_alloc(&c->os_tickets, (size_t)4 * os_groups);
    if (!rc) rc = dev_alloc(&c->os_tot, (size_t)4 * 512);
    if (!rc) rc = dev_alloc(&c->os_status, (size_t)512 * c->sort_blocks_cap);
    if (!rc) rc = dev_alloc(&c->os_status32, (size_t)512 * c->sort_blocks_cap);
    if (!rc) rc = dev_alloc(&c->mm_tileL, (size_t)c->sort_blocks_cap + 2);
    if (!rc) rc = dev_alloc(&c->mm_tileA, (size_t)c->sort_blocks_cap + 2);
    if (!rc && (hipHostMalloc((void**)&c->os_err_host, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
                hipHostGetDevicePointer((void**)&c->os_err_dev, c->os_err_host, 0) != hipSuccess)) {
        set_error("hipHostMalloc(mapped) failed");
        rc = SPH_E_NOMEM;
    }
    if (!rc) {
        *c->os_err_host = 0;
        if (hipMemset(c->os_status, 0, (size_t)512 * c->sort_blocks_cap * sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(c->os_status32, 0, (size_t)512 * c->sort_blocks_cap * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(c->os_tickets, 0, 4 * os_groups * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(c->os_hist, 0, os_groups * 4 * 512 * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(c->keyS, 0, tot * sizeof(uint32_t)) != hipSuccess ||
            hipMemset(c->keyS2, 0, tot * sizeof(uint32_t)) != hipSuccess) {
            set_error("hipMemset failed");
            rc = SPH_E_DEVICE;
        }
    }
    if (!rc) rc = dev_alloc(&c->d_scratch, (size_t)64);
    if (!rc && hipHostMalloc((void**)&c->h_scratch, 64 * sizeof(uint32_t)) != hipSuccess) {
        set_error("hipHostMalloc failed");
        rc = SPH_E_NOMEM;
    }
    if (const char* env = getenv("SPH_BLOCK_ORDER")) {          // "xcd,ztile,strip_log2" at create time: A/B runs (sph_set_block_order)
        int x = 1, z = 1, sh = 4, zd = 1, xr = 1;
        if (sscanf(env, "%d,%d,%d,%d,%d", &x, &z, &sh, &zd, &xr) >= 1) {
            c->order_xcd = x != 0; c->order_ztile = z != 0; if (sh >= 1 && sh <= 10) c->order_strip_sh = (uint32_t)sh;
            c->order_ztile_dens = zd != 0; c->order_xrot = xr != 0;
        }
    }
    if (const char* env = getenv("SPH_PAIR_SMALL_SLOTS")) {     // at create time: A/B runs (sph_set_pair_small_launch)
        unsigned long v = 0;
        if (sscanf(env, "%lu", &v) == 1) c->pair_small_slots = v > 0xFFFFFFFFul ? 0xFFFFFFFFu : (uint32_t)v;
    }
    {   // merge sort scratch
        const char* env = getenv("SPH_SORT_MERGE");
        c->sort_merge = !(env && env[0] == '0');
        const size_t nchunks = (size_t)ceil_div(capacity, 64u) + 1, ntiles = nchunks / 256 + 2;
        if (!rc) rc = dev_alloc(&c->mm_mask, nchunks);
        if (!rc) rc = dev_alloc(&c->mm_M64, nchunks);
        if (!rc) rc = dev_alloc(&c->mm_tile_cnt, ntiles);
        if (!rc) rc = dev_alloc(&c->mm_tile_off, ntiles);
        if (!rc) rc = dev_alloc(&c->mm_k0, (size_t)capacity);
        if (!rc) rc = dev_alloc(&c->mm_k1, (size_t)capacity);
        if (!rc) rc = dev_alloc(&c->mm_v1, (size_t)capacity);
        if (!rc) rc = dev_alloc(&c->mm_count, (size_t)1);
        if (!rc) rc = dev_alloc(&c->mm_total, (size_t)1);
        if (!rc && (hipHostMalloc((void**)&c->mm_count_host, 8 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
                    hipHostGetDevicePointer((void**)&c->mm_count_host_dev, c->mm_count_host, 0) != hipSuccess)) {
            set_error("hipHostMalloc(mapped) failed");
            rc = SPH_E_NOMEM;
        }
        if (!rc) {
            c->mm_count_host[0] = 0;
            c->mm_count_host[1] = 1; c->mm_count_host[2] = 0;          // "no estimate yet" (first key > last key)
            for (int k = 3; k < 8; k++) c->mm_count_host[k] = 0;
            if (hipMemset(c->mm_tile_cnt, 0, ntiles * sizeof(uint32_t)) != hipSuccess ||
                hipMemset(c->mm_count, 0, sizeof(uint32_t)) != hipSuccess ||
                hipMemset(c->mm_total, 0, sizeof(unsigned long long)) != hipSuccess) {
                set_error("hipMemset failed");
                rc = SPH_E_DEVICE;
            }
        }
    }
    if (!rc && hipMemset(c->cells_base, 0, ((size_t)c->grid.ncells + 2) * sizeof(uint2)) != hipSuccess) {
        set_error("hipMemset(cells) failed");
        rc = SPH_E_DEVICE;
    }
    if (!rc && (hipMemset(c->dp, 0, tot * sizeof(float2)) != hipSuccess ||
                hipMemset(c->cw, 0, tot * sizeof(float2)) != hipSuccess ||
                hipMemset(c->posi, 0, tot * sizeof(float4)) != hipSuccess ||
                hipMemset(c->velr, 0, tot * sizeof(float4)) != hipSuccess ||
                hipMemset(c->posi2, 0, tot * sizeof(float4)) != hipSuccess ||
                hipMemset(c->velr2, 0, tot * sizeof(float4)) != hipSuccess)) {
        set_error("hipMemset failed");
        rc = SPH_E_DEVICE;
    }
    if (rc) { free_all(c); delete c; return rc; }
    *out = c;
    return SPH_OK;
}

// ---- device timing --------------------------------------------------------------------------------------
void timing_collect(sph_ctx* c) {
    for (size_t k = 0; k + 3 <= c->events.size(); k += 3) {
        hipEvent_t a = c->events[k], b = c->events[k + 1];
        int phase = (int)(intptr_t)c->events[k + 2];
        hipEventSynchronize(b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) == hipSuccess) c->ph_ms[phase] += ms;
        hipEventDestroy(a);
        hipEventDestroy(b);
    }
    c->events.clear();
}

// ---- phase bodies ------------------------------------------------------------------------------------------
static int do_hash(sph_ctx* c) {
    PhaseTimer t(c, SPH_PH_ZINDEX);
    // the table of the previous step dies with its keys -- unless the sort is going to merge: then only the
    // cells the movers leave can become empty, and the sort clears just those (the table is rebuilt by the
    // sort's reorder pass over exactly the same owned slots).  The cells of a slab's old ghosts die here.
    const uint32_t lo = c->own_off - c->n_glo, hi = c->own_off + c->n + c->n_ghi;
    c->cells_clear_deferred = c->sort_merge && c->order_valid && c->cells_valid && c->cells_lo == lo && c->cells_hi == hi;
    int rc;
    c->ghost_clear_pending = false;
    if (c->cells_clear_deferred && c->defer_ghost_clear && (c->n_glo | c->n_ghi)) {
        // (the slab step: launch_sort, called next, clears them -- sph_ctx::defer_ghost_clear; cells_lo / cells_hi keep the ghosts
        // until then, so that a full sort's clearing of the whole table range still covers them)
        c->ghost_clear[0] = lo; c->ghost_clear[1] = c->own_off; c->ghost_clear[2] = c->own_off + c->n; c->ghost_clear[3] = hi;
        c->ghost_clear_pending = true;
        rc = SPH_OK;
    } else if (c->cells_clear_deferred) {
        rc = launch_cells_clear_2ranges(c, lo, c->own_off, c->own_off + c->n, hi);      // both ghost ranges, one launch
        c->cells_lo = c->own_off; c->cells_hi = c->own_off + c->n;
    } else {
        rc = launch_cells_clear(c);
    }
    if (rc) return rc;
    rc = launch_hash(c);
    if (rc) return rc;
    c->keys_fresh = false;               // consumed (the sort reorders k0)
    c->stage = sph_ctx::ST_HASHED;
    return SPH_OK;
}

static int do_sort(sph_ctx* c) {
    PhaseTimer t(c, SPH_PH_SORT);
    int rc = launch_sort(c);
    if (rc) return rc;
    c->n_glo = c->n_ghi = 0;
    c->halo_n_valid = false;
    c->stage = sph_ctx::ST_SORTED;
    c->have_dens = c->have_force = c->have_coll = false;
    return SPH_OK;
}

static int do_cells(sph_ctx* c) {
    PhaseTimer t(c, SPH_PH_BGRID);
    const uint32_t lo = c->own_off - c->n_glo, hi = c->own_off + c->n + c->n_ghi;
    if (!(c->cells_valid && c->cells_lo == lo && c->cells_hi == hi)) {   // whole-domain: built by the sort's reorder pass
        int rc = launch_cells_build(c);      // slab: adds the ghost cells to the owned ones the sort built
        if (rc) return rc;
    }
    c->stage = sph_ctx::ST_CELLS;
    return SPH_OK;
}

int step_hash(sph_ctx* c) { return do_hash(c); }
int step_sort(sph_ctx* c) { return do_sort(c); }
int step_cells(sph_ctx* c) { return do_cells(c); }

static int do_density(sph_ctx* c) {
    PhaseTimer t(c, SPH_PH_DENS);
    int rc = launch_density(c);
    if (rc) return rc;
    c->have_dens = true;
    return SPH_OK;
}

}  // namespace sph

namespace sph {
// A slab context takes over the cell layers [z_lo, z_hi) (sph_slab_recut): local keys, the ghost layers and the width of
// the sort keys follow; the cell table -- which the caller has CLEARED (no entry of the old numbering may survive) -- is
// kept when it is large enough, else replaced.  The particles keep their slots; the next sph_hash re-keys them.
int set_slab_range(sph_ctx* c, uint32_t z_lo, uint32_t z_hi) {
    SPH_REQUIRE(c && c->slab, SPH_E_INVALID, "set_slab_range needs a slab context");
    sph_ctx tmp;
    tmp.params = c->params;
    tmp.ghost_layers = c->ghost_layers;
    int rc = derive(&tmp, &c->params, z_lo, z_hi, true);
    if (rc) return rc;
    if (tmp.grid.ncells > c->cells_alloc) {
        SPH_HIP(hipStreamSynchronize(c->stream));
        uint2* base = nullptr;
        rc = dev_alloc(&base, (size_t)tmp.grid.ncells + 2);
        if (rc) return rc;
        if (hipMemset(base, 0, ((size_t)tmp.grid.ncells + 2) * sizeof(uint2)) != hipSuccess) {
            hipFree(base);
            set_error("set_slab_range: clearing the new cell table failed");
            return SPH_E_DEVICE;
        }
        hipFree(c->cells_base);
        c->cells_base = base; c->cells = base + 1; c->cells_alloc = tmp.grid.ncells;
    }
    c->grid = tmp.grid; c->z_lo = z_lo; c->z_hi = z_hi; c->key_bits = tmp.key_bits;
    c->cells_valid = false; c->cells_clear_deferred = false;
    c->keys_fresh = false; c->order_valid = false;
    c->stage = sph_ctx::ST_LOADED;
    c->have_dens = c->have_force = c->have_coll = false;
    c->n_glo = c->n_ghi = 0;
    c->halo_n_valid = false;
    return SPH_OK;
}
}  // namespace sph

using namespace sph;

// one thread per owned slot: particles whose creation index falls in the range take their new data
__global__ __launch_bounds__(256) void k_set_by_index(float4* __restrict__ posi, float4* __restrict__ velr, uint32_t n,
                                                      uint32_t first, uint32_t count, const float* __restrict__ pos,
                                                      const float* __restrict__ vel, float4* __restrict__ pos_by_index) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = posi[i];
    const uint32_t k = __float_as_uint(p.w) - first;
    if (k >= count) return;                               // also rejects indices below `first` (wraps)
    if (pos) {
        p.x = pos[3 * k]; p.y = pos[3 * k + 1]; p.z = pos[3 * k + 2];
        posi[i] = p;
        if (pos_by_index) pos_by_index[first + k] = make_float4(p.x, p.y, p.z, 1.0f);
    }
    if (vel) velr[i] = make_float4(vel[3 * k], vel[3 * k + 1], vel[3 * k + 2], 0.f);
}


__global__ __launch_bounds__(256) void k_gather_cells(const uint32_t* __restrict__ keys, uint32_t m,
                                                      const uint2* __restrict__ cells, uint2* __restrict__ out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < m) out[k] = cells[keys[k]];
}

extern "C" {

int sph_abi_version(void) { return SPH_ABI_VERSION; }

const char* sph_last_error(void) { return g_err.c_str(); }

int sph_device_count(int* is_gfx950) {
    int n = 0;
    if (is_gfx950) *is_gfx950 = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return SPH_E_DEVICE; }
    if (n > 0 && is_gfx950) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, 0) == hipSuccess) *is_gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }
    return n;
}

int sph_select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): libsph_hip has no CPU fallback", hipGetErrorString(e));
        return SPH_E_DEVICE;
    }
    SPH_REQUIRE(device >= 0 && device < n, SPH_E_DEVICE, "device %d out of range (%d devices)", device, n);
    hipDeviceProp_t prop;
    SPH_HIP(hipGetDeviceProperties(&prop, device));
    SPH_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, SPH_E_DEVICE,
                "device %d is %s; libsph_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    SPH_HIP(hipSetDevice(device));
    g_device = device;
    return SPH_OK;
}

int sph_selected_device(void) { return g_device; }

void sph_default_params(sph_params* p, const float box_dims[3], const uint32_t grid[3]) {
    memset(p, 0, sizeof(*p));
    for (int a = 0; a < 3; a++) {
        p->box_min[a] = -box_dims[a] / 2;      // particleSystem.cpp:55-60
        p->box_max[a] = box_dims[a] / 2;
        p->grid[a] = grid[a];
    }
    p->h = 0.1f;                                // particles_kernel.cuh:20-33
    p->mass = 65.f;
    p->rest_density = 1000.f;
    p->gas_constant = 2000.f;
    p->viscosity = 250.f;
    p->gravity_y = -9.81f * 11000;
    p->wall_eps = 0.00001f;
    p->wall_damping = -.75f;                    // particleSystem.cu:376
    p->restitution = 0.f;
    p->collision_param = 1.0f;
    p->particle_radius = 1.0f / 64.0f;          // particleSystem.cpp:51
}

uint32_t sph_grid_dim_for_edge(float edge, float h) { return next_pow2((uint32_t)(edge / (0.66666f * h))); }

int sph_create(sph_ctx** out, int device, uint32_t capacity, const sph_params* p) {
    if (!p) { set_error("null params"); return SPH_E_INVALID; }
    return create_impl(out, device, capacity, p, 0, p->grid[2], 0, false);
}

int sph_create_slab(sph_ctx** out, int device, uint32_t capacity, const sph_params* p, uint32_t z_lo, uint32_t z_hi,
                    uint32_t ghost_capacity) {
    if (!p) { set_error("null params"); return SPH_E_INVALID; }
    return create_impl(out, device, capacity, p, z_lo, z_hi, ghost_capacity, true);
}

int sph_create_slab_layers(sph_ctx** out, int device, uint32_t capacity, const sph_params* p, uint32_t z_lo, uint32_t z_hi,
                           uint32_t ghost_capacity, uint32_t ghost_layers) {
    if (!p) { set_error("null params"); return SPH_E_INVALID; }
    if (ghost_layers != 1u && ghost_layers != 2u) { set_error("a slab keeps 1 or 2 ghost layers per side, not %u", ghost_layers); return SPH_E_INVALID; }
    return create_impl(out, device, capacity, p, z_lo, z_hi, ghost_capacity, true, ghost_layers);
}

uint32_t sph_ghost_layers(const sph_ctx* c) { return c ? (c->slab ? c->ghost_layers : 0u) : 0u; }

void sph_destroy(sph_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    timing_collect(c);
    free_all(c);
    delete c;
}

int sph_set_stream(sph_ctx* c, void* s) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    c->stream = (hipStream_t)s;
    return SPH_OK;
}

int sph_set_params(sph_ctx* c, const sph_params* p) {
    SPH_REQUIRE(c && p, SPH_E_INVALID, "null argument");
    for (int a = 0; a < 3; a++)
        SPH_REQUIRE(p->grid[a] == c->params.grid[a], SPH_E_INVALID, "the grid cannot change after sph_create");
    sph_ctx tmp;   // validate first
    tmp.params = c->params;
    tmp.ghost_layers = c->ghost_layers;
    int rc = derive(&tmp, p, c->z_lo, c->z_hi, c->slab);
    if (rc) return rc;
    c->params = tmp.params; c->grid = tmp.grid; c->phys = tmp.phys;
    c->keys_fresh = false;      // the box may have moved: keys of the old box are stale
    c->order_valid = false;
    c->sort_form_both_until = c->sort_calls + 5;
    return SPH_OK;
}

int sph_get_params(const sph_ctx* c, sph_params* p) {
    SPH_REQUIRE(c && p, SPH_E_INVALID, "null argument");
    *p = c->params;
    return SPH_OK;
}

int sph_sync(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_HIP(hipStreamSynchronize(c->stream));
    SPH_REQUIRE(*c->os_err_host == 0u, SPH_E_DEVICE,
                "radix sort: a look-back on another tile's digit counts timed out (results of the last sort are invalid)");
    return SPH_OK;
}

uint32_t sph_num_particles(const sph_ctx* c) { return c ? c->n : 0; }
uint32_t sph_capacity(const sph_ctx* c) { return c ? c->cap : 0; }

int sph_upload(sph_ctx* c, uint32_t n, const float* pos, const float* vel, const uint32_t* index) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(n <= c->cap, SPH_E_CAPACITY, "%u particles > capacity %u", n, c->cap);
    SPH_REQUIRE(n == 0 || pos, SPH_E_INVALID, "null positions");
    SPH_HIP(hipSetDevice(c->device));
    std::vector<float4> hp(n), hv(n);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t idx = index ? index[i] : i;
        if (!c->slab) SPH_REQUIRE(idx < c->pos_out_cap, SPH_E_INVALID, "creation index %u >= capacity %u", idx, c->pos_out_cap);
        float w;
        memcpy(&w, &idx, 4);
        hp[i] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], w);
        hv[i] = vel ? make_float4(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    SPH_HIP(hipStreamSynchronize(c->stream));
    int rc = launch_cells_clear(c);
    if (rc) return rc;
    c->own_off = c->gcap;
    c->n = n; c->n_glo = c->n_ghi = 0;
    if (n) {
        SPH_HIP(hipMemcpyAsync(c->posi + c->own_off, hp.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        SPH_HIP(hipMemcpyAsync(c->velr + c->own_off, hv.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        SPH_HIP(hipMemsetAsync(c->dp + c->own_off, 0, n * sizeof(float2), c->stream));
        SPH_HIP(hipMemsetAsync(c->cw + c->own_off, 0, n * sizeof(float2), c->stream));
        if (!c->slab) {
            // gl_pos starts as the initial positions (initGrid writes m_hPos, particleSystem.cpp:865-868)
            std::vector<float4> ho(c->pos_out_cap, make_float4(0.f, 0.f, 0.f, 0.f));
            for (uint32_t i = 0; i < n; i++) {
                uint32_t idx = index ? index[i] : i;
                ho[idx] = make_float4(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], 1.0f);
            }
            SPH_HIP(hipMemcpyAsync(c->pos_out, ho.data(), ho.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
            SPH_HIP(hipStreamSynchronize(c->stream));
        }
    }
    SPH_HIP(hipStreamSynchronize(c->stream));
    c->stage = sph_ctx::ST_LOADED;
    c->keys_fresh = false;
    c->order_valid = false;     // the slots no longer follow the last sort
    c->sort_form_both_until = c->sort_calls + 5;
    c->have_dens = c->have_force = c->have_coll = false;
    return SPH_OK;
}

int sph_set_by_index(sph_ctx* c, uint32_t first_index, uint32_t count, const float* pos_xyz, const float* vel_xyz) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->slab || (uint64_t)first_index + count <= c->pos_out_cap, SPH_E_INVALID,
                "creation indices [%u, +%u) exceed the capacity %u", first_index, count, c->pos_out_cap);
    if (count == 0 || (!pos_xyz && !vel_xyz)) return SPH_OK;
    SPH_HIP(hipSetDevice(c->device));
    float* d_pos = nullptr; float* d_vel = nullptr;
    const size_t bytes = (size_t)count * 3 * sizeof(float);
    int rc = SPH_OK;
    if (pos_xyz) rc = dev_alloc(&d_pos, (size_t)count * 3);
    if (!rc && vel_xyz) rc = dev_alloc(&d_vel, (size_t)count * 3);
    if (!rc) {
        hipError_t e = hipSuccess;
        if (pos_xyz) e = hipMemcpyAsync(d_pos, pos_xyz, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && vel_xyz) e = hipMemcpyAsync(d_vel, vel_xyz, bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && c->n) {
            hipLaunchKernelGGL(k_set_by_index, dim3(ceil_div(c->n, 256)), dim3(256), 0, c->stream, c->posi + c->own_off,
                               c->velr + c->own_off, c->n, first_index, count, d_pos, d_vel,
                               c->slab ? nullptr : c->pos_out);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);        // the host buffers may go away now
        if (e != hipSuccess) { set_error("sph_set_by_index: %s", hipGetErrorString(e)); rc = SPH_E_DEVICE; }
    }
    hipFree(d_pos); hipFree(d_vel);
    if (rc) return rc;
    // positions moved under the keys: hash again; the slots still follow the last sort (order_valid stays)
    mm_drop_marks(c);
    c->keys_fresh = false;
    c->sort_form_both_until = c->sort_calls + 5;      // the last reported mover count says nothing about these particles
    c->stage = sph_ctx::ST_LOADED;
    c->have_dens = c->have_force = c->have_coll = false;
    return SPH_OK;
}

int sph_reset_lattice(sph_ctx* c, const uint32_t lattice[3], int jitter, const float jitter_dims[3], uint64_t start,
                      uint32_t count) {
    SPH_REQUIRE(c && lattice, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(count <= c->cap, SPH_E_CAPACITY, "%u particles > capacity %u", count, c->cap);
    const uint64_t total = (uint64_t)lattice[0] * lattice[1] * lattice[2];
    SPH_REQUIRE(lattice[0] && lattice[1] && lattice[2] && start + count <= total, SPH_E_INVALID,
                "index range [%llu, +%u) outside the %llu-particle lattice", (unsigned long long)start, count,
                (unsigned long long)total);
    SPH_REQUIRE(c->slab || start + count <= c->pos_out_cap, SPH_E_INVALID, "creation indices exceed the capacity %u",
                c->pos_out_cap);
    SPH_REQUIRE(total <= 0xFFFFFFFFull, SPH_E_INVALID, "creation indices are 32-bit");
    SPH_HIP(hipSetDevice(c->device));
    int rc = launch_cells_clear(c);
    if (rc) return rc;
    c->own_off = c->gcap;
    c->n = count; c->n_glo = c->n_ghi = 0;
    rc = launch_reset_lattice(c, lattice, jitter, jitter_dims, start, count);
    if (rc) return rc;
    c->stage = sph_ctx::ST_LOADED;
    c->keys_fresh = false;
    c->order_valid = false;     // the slots no longer follow the last sort
    c->sort_form_both_until = c->sort_calls + 5;
    c->have_dens = c->have_force = c->have_coll = false;
    return SPH_OK;
}

static int fetch_sorted(sph_ctx* c, std::vector<float4>* hp, std::vector<float4>* hv, std::vector<float2>* hd) {
    SPH_HIP(hipSetDevice(c->device));
    const uint32_t n = c->n;
    if (hp) { hp->resize(n); if (n) SPH_HIP(hipMemcpyAsync(hp->data(), c->posi + c->own_off, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream)); }
    if (hv) { hv->resize(n); if (n) SPH_HIP(hipMemcpyAsync(hv->data(), c->velr + c->own_off, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream)); }
    if (hd) { hd->resize(n); if (n) SPH_HIP(hipMemcpyAsync(hd->data(), c->dp + c->own_off, n * sizeof(float2), hipMemcpyDeviceToHost, c->stream)); }
    SPH_HIP(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

static inline uint32_t idx_of(const float4& p) { uint32_t u; memcpy(&u, &p.w, 4); return u; }

// Particles whose creation index lies outside [base, base + count) are not written: the caller's arrays hold
// `count` entries and nothing beyond them is touched (an index below `base` wraps to a huge offset and fails
// the same test).
int sph_download(sph_ctx* c, uint32_t base, uint32_t count, float* pos, float* vel, float* density, float* pressure) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    std::vector<float4> hp, hv;
    std::vector<float2> hd;
    int rc = fetch_sorted(c, &hp, vel ? &hv : nullptr, (density || pressure) ? &hd : nullptr);
    if (rc) return rc;
    for (uint32_t s = 0; s < c->n; s++) {
        const uint32_t rel = idx_of(hp[s]) - base;
        if (rel >= count) continue;
        const size_t i = rel;
        if (pos) { pos[3 * i] = hp[s].x; pos[3 * i + 1] = hp[s].y; pos[3 * i + 2] = hp[s].z; }
        if (vel) { vel[3 * i] = hv[s].x; vel[3 * i + 1] = hv[s].y; vel[3 * i + 2] = hv[s].z; }
        if (density) density[i] = hd[s].x;
        if (pressure) pressure[i] = hd[s].y;
    }
    return SPH_OK;
}

int sph_download_owned(sph_ctx* c, float* pos, float* vel, uint32_t* index) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    std::vector<float4> hp, hv;
    int rc = fetch_sorted(c, &hp, vel ? &hv : nullptr, nullptr);
    if (rc) return rc;
    for (uint32_t s = 0; s < c->n; s++) {
        if (pos) { pos[3 * s] = hp[s].x; pos[3 * s + 1] = hp[s].y; pos[3 * s + 2] = hp[s].z; }
        if (vel) { vel[3 * s] = hv[s].x; vel[3 * s + 1] = hv[s].y; vel[3 * s + 2] = hv[s].z; }
        if (index) index[s] = idx_of(hp[s]);
    }
    return SPH_OK;
}

int sph_download_forces(sph_ctx* c, uint32_t base, uint32_t n_out, float* fp, float* fv, float* dv, int32_t* count) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE((!fp && !fv) || c->have_force, SPH_E_STATE, "sph_force has not run for this particle order");
    SPH_REQUIRE((!dv && !count) || c->have_coll, SPH_E_STATE, "sph_collide has not run for this particle order");
    std::vector<float4> hp;
    int rc = fetch_sorted(c, &hp, nullptr, nullptr);
    if (rc) return rc;
    const uint32_t n = c->n;
    std::vector<float4> a(n), b(n), d(n);
    if (n && (fp || fv)) {
        SPH_HIP(hipMemcpy(a.data(), c->fpress + c->own_off, n * sizeof(float4), hipMemcpyDeviceToHost));
        SPH_HIP(hipMemcpy(b.data(), c->fvisc + c->own_off, n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    if (n && (dv || count)) SPH_HIP(hipMemcpy(d.data(), c->dvel + c->own_off, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t rel = idx_of(hp[s]) - base;
        if (rel >= n_out) continue;
        const size_t i = rel;
        if (fp) { fp[3 * i] = a[s].x; fp[3 * i + 1] = a[s].y; fp[3 * i + 2] = a[s].z; }
        if (fv) { fv[3 * i] = b[s].x; fv[3 * i + 1] = b[s].y; fv[3 * i + 2] = b[s].z; }
        if (dv) { dv[3 * i] = d[s].x; dv[3 * i + 1] = d[s].y; dv[3 * i + 2] = d[s].z; }
        if (count) { uint32_t u; memcpy(&u, &d[s].w, 4); count[i] = (int32_t)u; }
    }
    return SPH_OK;
}

// ---- snapshots --------------------------------------------------------------------------------------------
// layout: u32 magic 'SPHS', u32 version, u32 n, u32 sizeof(sph_params), sph_params, then n float4 posi
// (x, y, z, creation index bits) and n float4 velr, both in slot order.
static const uint32_t kSnapMagic = 0x53485053u, kSnapVersion = 1u;

int sph_snapshot_save(sph_ctx* c, const char* path) {
    SPH_REQUIRE(c && path, SPH_E_INVALID, "null argument");
    std::vector<float4> hp, hv;
    int rc = fetch_sorted(c, &hp, &hv, nullptr);
    if (rc) return rc;
    FILE* f = fopen(path, "wb");
    SPH_REQUIRE(f, SPH_E_INVALID, "cannot open %s for writing", path);
    const uint32_t hdr[4] = {kSnapMagic, kSnapVersion, c->n, (uint32_t)sizeof(sph_params)};
    bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1 && fwrite(&c->params, sizeof(sph_params), 1, f) == 1;
    if (c->n) ok = ok && fwrite(hp.data(), sizeof(float4), c->n, f) == c->n && fwrite(hv.data(), sizeof(float4), c->n, f) == c->n;
    ok = (fclose(f) == 0) && ok;
    SPH_REQUIRE(ok, SPH_E_INVALID, "short write to %s", path);
    return SPH_OK;
}

static int snap_header(FILE* f, const char* path, uint32_t* n, sph_params* p) {
    uint32_t hdr[4];
    SPH_REQUIRE(fread(hdr, sizeof(hdr), 1, f) == 1 && hdr[0] == kSnapMagic, SPH_E_INVALID, "%s is not an SPH snapshot", path);
    SPH_REQUIRE(hdr[1] == kSnapVersion && hdr[3] == sizeof(sph_params), SPH_E_INVALID, "%s: unsupported snapshot version", path);
    SPH_REQUIRE(fread(p, sizeof(sph_params), 1, f) == 1, SPH_E_INVALID, "%s: truncated", path);
    *n = hdr[2];
    return SPH_OK;
}

int sph_snapshot_info(const char* path, uint32_t* n, sph_params* p) {
    SPH_REQUIRE(path && n && p, SPH_E_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    SPH_REQUIRE(f, SPH_E_INVALID, "cannot open %s", path);
    int rc = snap_header(f, path, n, p);
    fclose(f);
    return rc;
}

int sph_snapshot_load(sph_ctx* c, const char* path) {
    SPH_REQUIRE(c && path, SPH_E_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    SPH_REQUIRE(f, SPH_E_INVALID, "cannot open %s", path);
    uint32_t n = 0;
    sph_params p;
    int rc = snap_header(f, path, &n, &p);
    if (rc) { fclose(f); return rc; }
    if (n > c->cap) { fclose(f); set_error("snapshot holds %u particles, context capacity is %u", n, c->cap); return SPH_E_CAPACITY; }
    std::vector<float4> hp(n), hv(n);
    bool ok = n == 0 || (fread(hp.data(), sizeof(float4), n, f) == n && fread(hv.data(), sizeof(float4), n, f) == n);
    fclose(f);
    SPH_REQUIRE(ok, SPH_E_INVALID, "%s: truncated", path);
    rc = sph_set_params(c, &p);            // rejects a different grid
    if (rc) return rc;
    std::vector<float> pos((size_t)n * 3), vel((size_t)n * 3);
    std::vector<uint32_t> idx(n);
    if (!c->slab) {
        // a whole-domain snapshot numbers its particles 0..n-1, each once: a file that says otherwise is
        // corrupt (external input -- every by-index buffer of the host class is sized from n)
        std::vector<bool> seen(n, false);
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t k = idx_of(hp[i]);
            SPH_REQUIRE(k < n && !seen[k], SPH_E_INVALID, "%s: creation index %u of record %u is out of range or repeated", path, k, i);
            seen[k] = true;
        }
    }
    for (uint32_t i = 0; i < n; i++) {
        pos[3 * i] = hp[i].x; pos[3 * i + 1] = hp[i].y; pos[3 * i + 2] = hp[i].z;
        vel[3 * i] = hv[i].x; vel[3 * i + 1] = hv[i].y; vel[3 * i + 2] = hv[i].z;
        idx[i] = idx_of(hp[i]);
    }
    return sph_upload(c, n, pos.data(), vel.data(), idx.data());
}

int sph_positions_dev(sph_ctx* c, void** out) {
    SPH_REQUIRE(c && out, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(!c->slab, SPH_E_STATE, "slab contexts keep no by-index position buffer");
    *out = c->pos_out;
    return SPH_OK;
}

int sph_download_positions4(sph_ctx* c, float* out) {
    SPH_REQUIRE(c && out, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(!c->slab, SPH_E_STATE, "slab contexts keep no by-index position buffer");
    SPH_HIP(hipSetDevice(c->device));
    SPH_HIP(hipMemcpyAsync(out, c->pos_out, (size_t)c->pos_out_cap * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_get_keys(sph_ctx* c, uint32_t* keys) {
    SPH_REQUIRE(c && keys, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_HASHED, SPH_E_STATE, "sph_hash has not run");
    SPH_HIP(hipSetDevice(c->device));
    const uint32_t* src = c->stage == sph_ctx::ST_HASHED ? c->k0 : c->keyS + c->own_off;
    if (c->n) SPH_HIP(hipMemcpyAsync(keys, src, c->n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(hipStreamSynchronize(c->stream));
    return SPH_OK;
}

int sph_get_order(sph_ctx* c, uint32_t* index) {
    SPH_REQUIRE(c && index, SPH_E_INVALID, "null argument");
    std::vector<float4> hp;
    int rc = fetch_sorted(c, &hp, nullptr, nullptr);
    if (rc) return rc;
    for (uint32_t s = 0; s < c->n; s++) index[s] = idx_of(hp[s]);
    return SPH_OK;
}

int sph_get_cell_range(sph_ctx* c, uint32_t cell, uint32_t* start, uint32_t* end) {
    SPH_REQUIRE(c && start && end, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->cells_valid, SPH_E_STATE, "sph_build_cells has not run");
    SPH_REQUIRE(cell < c->grid.ncells, SPH_E_INVALID, "cell %u out of range", cell);
    uint2 v;
    SPH_HIP(hipSetDevice(c->device));
    SPH_HIP(hipStreamSynchronize(c->stream));
    SPH_HIP(hipMemcpy(&v, c->cells + cell, sizeof(v), hipMemcpyDeviceToHost));
    *start = v.x - (v.y > v.x ? c->own_off - c->n_glo : 0);
    *end = v.y - (v.y > v.x ? c->own_off - c->n_glo : 0);
    return SPH_OK;
}

int sph_get_cells(sph_ctx* c, uint32_t max_cells, uint32_t* key, uint32_t* start, uint32_t* count) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->cells_valid, SPH_E_STATE, "sph_build_cells has not run");
    SPH_HIP(hipSetDevice(c->device));
    SPH_HIP(hipStreamSynchronize(c->stream));
    const uint32_t lo = c->cells_lo, hi = c->cells_hi;
    std::vector<uint32_t> ks(hi - lo);
    if (hi > lo) SPH_HIP(hipMemcpy(ks.data(), c->keyS + lo, (hi - lo) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> uniq;                         // occupied cells, in slot order
    for (uint32_t s = 0; s < hi - lo; s++)
        if (s == 0 || ks[s] != ks[s - 1]) uniq.push_back(ks[s]);
    const uint32_t m = (uint32_t)uniq.size(), take = m < max_cells ? m : max_cells;
    if (take) {                                         // one gather on the device, one copy back
        uint32_t* d_keys = nullptr; uint2* d_out = nullptr;
        int rc = dev_alloc(&d_keys, (size_t)take);
        if (!rc) rc = dev_alloc(&d_out, (size_t)take);
        std::vector<uint2> got(take);
        if (!rc) {
            hipError_t e = hipMemcpy(d_keys, uniq.data(), take * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_gather_cells, dim3(ceil_div(take, 256)), dim3(256), 0, c->stream, d_keys, take, c->cells,
                                   d_out);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipMemcpy(got.data(), d_out, take * sizeof(uint2), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { set_error("sph_get_cells: %s", hipGetErrorString(e)); rc = SPH_E_DEVICE; }
        }
        hipFree(d_keys); hipFree(d_out);
        if (rc) return rc;
        for (uint32_t k = 0; k < take; k++) {
            if (key) key[k] = uniq[k];
            if (start) start[k] = got[k].x - lo;
            if (count) count[k] = got[k].y - got[k].x;
        }
    }
    return (int)m;
}

uint32_t sph_cell_key(const sph_ctx* c, uint32_t x, uint32_t y, uint32_t z) {
    if (!c) return 0;
    return ((uint32_t)((int32_t)z - c->grid.z_off) * c->grid.g[1] + y) * c->grid.g[0] + x;
}

// ---- phases -------------------------------------------------------------------------------------------------------
int sph_hash(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_HIP(hipSetDevice(c->device));
    return do_hash(c);
}

int sph_sort(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->stage == sph_ctx::ST_HASHED, SPH_E_STATE, "sph_sort needs sph_hash first");
    SPH_HIP(hipSetDevice(c->device));
    return do_sort(c);
}

int sph_build_cells(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_SORTED, SPH_E_STATE, "sph_build_cells needs sph_sort first");
    SPH_HIP(hipSetDevice(c->device));
    return do_cells(c);
}

int sph_density(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_CELLS && c->cells_valid, SPH_E_STATE, "sph_density needs sph_build_cells first");
    SPH_HIP(hipSetDevice(c->device));
    return do_density(c);
}

int sph_force(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_CELLS && c->have_dens, SPH_E_STATE, "sph_force needs sph_density first");
    SPH_HIP(hipSetDevice(c->device));
    PhaseTimer t(c, SPH_PH_FORCE);
    int rc = launch_force(c, true, false, false, 0.f);
    if (rc) return rc;
    c->have_force = true;
    return SPH_OK;
}

int sph_collide(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_CELLS && c->cells_valid, SPH_E_STATE, "sph_collide needs sph_build_cells first");
    SPH_HIP(hipSetDevice(c->device));
    PhaseTimer t(c, SPH_PH_COLLISION);
    int rc = launch_force(c, false, true, false, 0.f);
    if (rc) return rc;
    c->have_coll = true;
    return SPH_OK;
}

int sph_integrate(sph_ctx* c, float dt) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->have_dens && c->have_force && c->have_coll, SPH_E_STATE,
                "sph_integrate needs sph_density, sph_force and sph_collide first");
    SPH_HIP(hipSetDevice(c->device));
    PhaseTimer t(c, SPH_PH_INTEGRATE);
    int rc = launch_integrate(c, dt);
    if (rc) return rc;
    c->have_force = c->have_coll = false;   // consumed; positions moved
    return SPH_OK;
}

int sph_step(sph_ctx* c, float dt, uint32_t n_steps) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_HIP(hipSetDevice(c->device));
    for (uint32_t s = 0; s < n_steps; s++) {
        int rc = do_hash(c);
        if (!rc) rc = do_sort(c);
        if (!rc) rc = do_cells(c);
        if (!rc) rc = do_density(c);
        if (rc) return rc;
        {
            PhaseTimer t(c, SPH_PH_FORCE);
            rc = launch_force(c, true, true, true, dt);
            if (rc) return rc;
        }
        c->have_force = c->have_coll = false;
        if (c->timing) { c->timed_steps++; if (c->events.size() > 3 * 4096) timing_collect(c); }
    }
    return SPH_OK;
}

int sph_force_collide_integrate(sph_ctx* c, float dt) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_CELLS && c->have_dens, SPH_E_STATE,
                "sph_force_collide_integrate needs sph_density first");
    SPH_HIP(hipSetDevice(c->device));
    PhaseTimer t(c, SPH_PH_FORCE);
    int rc = launch_force(c, true, true, true, dt);
    if (rc) return rc;
    c->have_force = c->have_coll = false;
    return SPH_OK;
}

int sph_step_phased(sph_ctx* c, float dt, uint32_t n_steps) {
    for (uint32_t s = 0; s < n_steps; s++) {
        int rc = sph_hash(c);
        if (!rc) rc = sph_sort(c);
        if (!rc) rc = sph_build_cells(c);
        if (!rc) rc = sph_density(c);
        if (!rc) rc = sph_force(c);
        if (!rc) rc = sph_collide(c);
        if (!rc) rc = sph_integrate(c, dt);
        if (rc) return rc;
        if (c->timing) { c->timed_steps++; if (c->events.size() > 3 * 4096) timing_collect(c); }
    }
    return SPH_OK;
}

int sph_timing_enable(sph_ctx* c, int on) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    c->timing = on != 0;
    return SPH_OK;
}

int sph_timing_get(sph_ctx* c, float ms[SPH_PH_COUNT], uint32_t* n_steps) {
    SPH_REQUIRE(c && ms, SPH_E_INVALID, "null argument");
    SPH_HIP(hipSetDevice(c->device));
    timing_collect(c);
    for (int k = 0; k < SPH_PH_COUNT; k++) ms[k] = (float)c->ph_ms[k];
    if (n_steps) *n_steps = c->timed_steps;
    return SPH_OK;
}

int sph_timing_reset(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_HIP(hipSetDevice(c->device));
    timing_collect(c);
    for (int k = 0; k < SPH_PH_COUNT; k++) c->ph_ms[k] = 0;
    c->timed_steps = 0;
    return SPH_OK;
}

int sph_last_sort_skipped(const sph_ctx* c) { return c && c->last_sort_skipped ? 1 : 0; }

int sph_set_precision(sph_ctx* c, int precision) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(precision == SPH_PRECISION_F32 || precision == SPH_PRECISION_MIXED_F16, SPH_E_INVALID,
                "unknown precision %d", precision);
    c->precision = precision;
    return SPH_OK;
}

int sph_get_precision(const sph_ctx* c) { return c ? c->precision : SPH_PRECISION_F32; }

int sph_set_sort_mode(sph_ctx* c, int merge) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    c->sort_merge = merge != 0;
    c->sort_merge_always = merge == 2;
    return SPH_OK;
}

int sph_test_trust_mover_hint(sph_ctx* c) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    c->sort_form_both_until = 0;
    return SPH_OK;
}

int sph_sort_forms(const sph_ctx* c, uint64_t out[3]) {
    SPH_REQUIRE(c && out, SPH_E_INVALID, "null argument");
    for (int k = 0; k < 3; k++) out[k] = c->sort_forms[k];
    return SPH_OK;
}

int sph_set_block_order(sph_ctx* c, int xcd, int ztile, uint32_t strip_blocks_log2) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(strip_blocks_log2 >= 1u && strip_blocks_log2 <= 10u, SPH_E_INVALID, "strip of 2^%u blocks", strip_blocks_log2);
    c->order_xcd = xcd != 0;
    c->order_ztile = c->order_ztile_dens = ztile != 0;
    c->order_strip_sh = strip_blocks_log2;
    return SPH_OK;
}

int sph_set_direct_hull(sph_ctx* c, uint32_t slots) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    c->direct_hull = slots;
    return SPH_OK;
}

int sph_set_pair_small_launch(sph_ctx* c, uint32_t slots) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    c->pair_small_slots = slots;
    return SPH_OK;
}

int sph_sort_stats(sph_ctx* c, uint64_t* sorts, uint64_t* merges, uint64_t* skips, uint32_t* last_movers,
                   uint64_t* movers_total) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_HIP(hipSetDevice(c->device));
    SPH_HIP(hipStreamSynchronize(c->stream));
    if (movers_total) {
        unsigned long long t = 0;
        SPH_HIP(hipMemcpy(&t, c->mm_total, sizeof(t), hipMemcpyDeviceToHost));
        *movers_total = t;
    }
    if (sorts) *sorts = c->sort_calls;
    if (merges) *merges = c->sort_merges;
    if (skips) *skips = c->sort_skips;
    if (last_movers) *last_movers = *c->mm_count_host;
    return SPH_OK;
}

}  // extern "C"
