// sph_halo.hip -- z-slab support: migrants, ghost layers, layer histogram (multi-GPU).
//
// No counterpart in the reference (single GPU).  A rank owns the cell layers [z_lo, z_hi) of the
// global grid; cell keys are slab-local with z slowest, so after the sort
//   * particles that left the slab sit at the two ENDS of the owned range (local layer 0 / zl-1),
//   * the boundary layers a neighbour needs as ghosts are the first / last layer of what is left,
// i.e. both are contiguous slices of the sorted SoA arrays.  The library packs those slices into
// caller-provided device buffers (the caller moves them with RCCL send/recv via torch.distributed)
// and installs received records: ghosts go directly in front of / behind the owned range, already
// in key order, so no re-sort is needed for ghosts.
#include "sph_device.hpp"

namespace sph {

__global__ __launch_bounds__(256) void k_unpack_dp(const float2* __restrict__ src, float2* __restrict__ dp, float2* __restrict__ cw,
                                                   uint32_t n, Phys ph) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n) return;
    const float2 v = src[t];
    dp[t] = v;
    cw[t] = neighbour_terms(ph, v.x, v.y);
}

__global__ void k_lower_bounds(const uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ targets,
                               uint32_t m, uint32_t* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    uint32_t v = targets[t], lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    out[t] = lo;
}

__global__ __launch_bounds__(256) void k_pack(const float4* __restrict__ posi, const float4* __restrict__ velr,
                                              uint32_t n, float4* __restrict__ rec) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    rec[2 * i] = posi[i];
    rec[2 * i + 1] = velr[i];
}

__global__ __launch_bounds__(256) void k_unpack(const float4* __restrict__ rec, uint32_t n, float4* __restrict__ posi,
                                                float4* __restrict__ velr, uint32_t* __restrict__ key, GridDesc g,
                                                int write_key) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = rec[2 * i];
    posi[i] = p;
    velr[i] = rec[2 * i + 1];
    if (write_key) key[i] = cell_key(g, p.x, p.y, p.z);
}

// lower bounds of up to 62 key targets in the owned sorted keys, returned through pinned memory
static int lower_bounds(sph_ctx* c, const uint32_t* targets, uint32_t m, uint32_t* out) {
    SPH_REQUIRE(m <= 31, SPH_E_INVALID, "too many targets");
    SPH_HIP(hipSetDevice(c->device));
    for (uint32_t k = 0; k < m; k++) c->h_scratch[k] = targets[k];
    SPH_HIP(hipMemcpyAsync(c->d_scratch, c->h_scratch, m * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_lower_bounds, dim3(1), dim3(64), 0, c->stream, c->keyS + c->own_off, c->n, c->d_scratch, m,
                       c->d_scratch + 32);
    SPH_HIP(hipGetLastError());
    SPH_HIP(hipMemcpyAsync(c->h_scratch + 32, c->d_scratch + 32, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(hipStreamSynchronize(c->stream));
    for (uint32_t k = 0; k < m; k++) out[k] = c->h_scratch[32 + k];
    return SPH_OK;
}

static int pack_slice(sph_ctx* c, uint32_t first, uint32_t count, void* buf, uint32_t capacity) {
    SPH_REQUIRE(count <= capacity, SPH_E_CAPACITY, "halo slice of %u records > buffer capacity %u", count, capacity);
    if (count == 0) return SPH_OK;
    SPH_REQUIRE(buf, SPH_E_INVALID, "null halo buffer");
    hipLaunchKernelGGL(k_pack, dim3(ceil_div(count, 256)), dim3(256), 0, c->stream, c->posi + first, c->velr + first, count,
                       (float4*)buf);
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

}  // namespace sph

using namespace sph;

extern "C" {

int sph_migrants_count(sph_ctx* c, uint32_t count[2]) {
    SPH_REQUIRE(c && count, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->stage == sph_ctx::ST_SORTED, SPH_E_STATE, "sph_migrants_count needs sph_sort first");
    const uint32_t layer = c->grid.g[0] * c->grid.g[1];
    const uint32_t G = c->ghost_layers;
    uint32_t tg[2] = {G * layer, (c->grid.zl - G) * layer}, lb[2];
    int rc = lower_bounds(c, tg, 2, lb);
    if (rc) return rc;
    count[0] = lb[0];
    count[1] = c->n - lb[1];
    return SPH_OK;
}

int sph_slab_counts(sph_ctx* c, uint32_t count[4]) {
    SPH_REQUIRE(c && count, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->stage == sph_ctx::ST_SORTED, SPH_E_STATE, "sph_slab_counts needs sph_sort first");
    const uint32_t layer = c->grid.g[0] * c->grid.g[1];
    const uint32_t G = c->ghost_layers;
    uint32_t tg[4] = {G * layer, (G + 1) * layer, (c->grid.zl - G - 1) * layer, (c->grid.zl - G) * layer}, lb[4];
    int rc = lower_bounds(c, tg, 4, lb);
    if (rc) return rc;
    count[0] = lb[0];
    count[1] = lb[1] - lb[0];
    count[2] = lb[3] - lb[2];
    count[3] = c->n - lb[3];
    return SPH_OK;
}

int sph_migrants_pack(sph_ctx* c, void* buf_dev[2], uint32_t capacity) {
    SPH_REQUIRE(c && buf_dev, SPH_E_INVALID, "null argument");
    uint32_t m[2];
    int rc = sph_migrants_count(c, m);
    if (rc) return rc;
    rc = pack_slice(c, c->own_off, m[0], buf_dev[0], capacity);
    if (!rc) rc = pack_slice(c, c->own_off + c->n - m[1], m[1], buf_dev[1], capacity);
    if (rc) return rc;
    if (c->cells_valid && c->cells_lo == c->own_off && c->cells_hi == c->own_off + c->n) {
        // the sort built the table over all owned slots: drop the cells of the particles that leave (they sit in
        // the two ghost layers, which hold nothing else until the ghosts are installed)
        rc = launch_cells_clear_range(c, c->own_off, c->own_off + m[0]);
        if (!rc) rc = launch_cells_clear_range(c, c->own_off + c->n - m[1], c->own_off + c->n);
        if (rc) return rc;
        c->cells_lo += m[0];
        c->cells_hi -= m[1];
    }
    c->own_off += m[0];
    c->n -= m[0] + m[1];
    return SPH_OK;
}

int sph_migrants_append(sph_ctx* c, const void* buf_dev, uint32_t n_in) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    if (n_in == 0) return SPH_OK;
    SPH_REQUIRE(buf_dev, SPH_E_INVALID, "null buffer");
    SPH_REQUIRE(c->n + n_in <= c->cap && c->own_off + c->n + n_in <= c->tot, SPH_E_CAPACITY,
                "%u + %u particles exceed the capacity %u", c->n, n_in, c->cap);
    SPH_HIP(hipSetDevice(c->device));
    const uint32_t at = c->own_off + c->n;
    hipLaunchKernelGGL(k_unpack, dim3(ceil_div(n_in, 256)), dim3(256), 0, c->stream, (const float4*)buf_dev, n_in,
                       c->posi + at, c->velr + at, (uint32_t*)nullptr, c->grid, 0);
    SPH_HIP(hipGetLastError());
    c->n += n_in;
    c->keys_fresh = false;
    c->order_valid = false;     // the slots no longer follow the last sort
    c->stage = sph_ctx::ST_LOADED;   // order destroyed: hash + sort again
    return SPH_OK;
}

int sph_halo_count(sph_ctx* c, uint32_t count[2]) {
    SPH_REQUIRE(c && count, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_SORTED, SPH_E_STATE, "sph_halo_count needs sph_sort first");
    const uint32_t layer = c->grid.g[0] * c->grid.g[1];
    const uint32_t G = c->ghost_layers;
    uint32_t tg[2] = {(G + 1) * layer, (c->grid.zl - G - 1) * layer}, lb[2];
    int rc = lower_bounds(c, tg, 2, lb);
    if (rc) return rc;
    count[0] = lb[0];
    count[1] = c->n - lb[1];
    return SPH_OK;
}

int sph_halo_pack_counts(sph_ctx* c, void* buf_dev[2], uint32_t capacity, const uint32_t count[2]) {
    SPH_REQUIRE(c && buf_dev, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_SORTED, SPH_E_STATE, "sph_halo_pack needs sph_sort first");
    uint32_t m[2];
    if (count) {
        m[0] = count[0]; m[1] = count[1];
        SPH_REQUIRE(m[0] <= c->n && m[1] <= c->n, SPH_E_INVALID, "halo counts exceed the owned particles");
    } else {
        int rc = sph_halo_count(c, m);
        if (rc) return rc;
    }
    SPH_HIP(hipSetDevice(c->device));
    int rc = pack_slice(c, c->own_off, m[0], buf_dev[0], capacity);
    if (!rc) rc = pack_slice(c, c->own_off + c->n - m[1], m[1], buf_dev[1], capacity);
    if (rc) return rc;
    c->halo_n[0] = m[0]; c->halo_n[1] = m[1]; c->halo_n_valid = true;
    return SPH_OK;
}

int sph_halo_pack(sph_ctx* c, void* buf_dev[2], uint32_t capacity) { return sph_halo_pack_counts(c, buf_dev, capacity, nullptr); }

int sph_halo_unpack(sph_ctx* c, const void* lo_dev, uint32_t n_lo, const void* hi_dev, uint32_t n_hi) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_SORTED, SPH_E_STATE, "sph_halo_unpack needs sph_sort first");
    SPH_REQUIRE(n_lo <= c->own_off && c->own_off + c->n + n_hi <= c->tot && n_lo <= c->gcap && n_hi <= c->gcap,
                SPH_E_CAPACITY, "ghost layers of %u / %u records exceed the ghost capacity %u", n_lo, n_hi, c->gcap);
    SPH_HIP(hipSetDevice(c->device));
    if (!(c->cells_valid && c->cells_lo == c->own_off && c->cells_hi == c->own_off + c->n)) {
        int rc = launch_cells_clear(c);   // a table that includes old ghosts; one over just the owned slots is kept
        if (rc) return rc;
    }
    if (n_lo) {
        SPH_REQUIRE(lo_dev, SPH_E_INVALID, "null buffer");
        const uint32_t at = c->own_off - n_lo;
        hipLaunchKernelGGL(k_unpack, dim3(ceil_div(n_lo, 256)), dim3(256), 0, c->stream, (const float4*)lo_dev, n_lo,
                           c->posi + at, c->velr + at, c->keyS + at, c->grid, 1);
    }
    if (n_hi) {
        SPH_REQUIRE(hi_dev, SPH_E_INVALID, "null buffer");
        const uint32_t at = c->own_off + c->n;
        hipLaunchKernelGGL(k_unpack, dim3(ceil_div(n_hi, 256)), dim3(256), 0, c->stream, (const float4*)hi_dev, n_hi,
                           c->posi + at, c->velr + at, c->keyS + at, c->grid, 1);
    }
    SPH_HIP(hipGetLastError());
    c->n_glo = n_lo;
    c->n_ghi = n_hi;
    c->stage = sph_ctx::ST_SORTED;
    return SPH_OK;
}

int sph_halo_pack_density(sph_ctx* c, void* buf_dev[2], uint32_t capacity) {
    SPH_REQUIRE(c && buf_dev, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->have_dens, SPH_E_STATE, "sph_halo_pack_density needs sph_density first");
    uint32_t m[2];
    if (c->halo_n_valid) {          // same particles, same order as the position halo of this step
        m[0] = c->halo_n[0]; m[1] = c->halo_n[1];
    } else {
        int rc = sph_halo_count(c, m);
        if (rc) return rc;
    }
    SPH_HIP(hipSetDevice(c->device));
    SPH_REQUIRE(m[0] <= capacity && m[1] <= capacity, SPH_E_CAPACITY, "halo slice > buffer capacity %u", capacity);
    if (m[0]) SPH_HIP(hipMemcpyAsync(buf_dev[0], c->dp + c->own_off, m[0] * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
    if (m[1]) SPH_HIP(hipMemcpyAsync(buf_dev[1], c->dp + c->own_off + c->n - m[1], m[1] * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
    return SPH_OK;
}

int sph_halo_unpack_density(sph_ctx* c, const void* lo_dev, const void* hi_dev) {
    SPH_REQUIRE(c, SPH_E_INVALID, "null context");
    SPH_HIP(hipSetDevice(c->device));
    SPH_REQUIRE((!c->n_glo || lo_dev) && (!c->n_ghi || hi_dev), SPH_E_INVALID, "null buffer");
    for (int side = 0; side < 2; side++) {       // (rho, p) and the neighbour terms the force pass reads of a ghost
        const uint32_t cnt = side == 0 ? c->n_glo : c->n_ghi, at = side == 0 ? c->own_off - c->n_glo : c->own_off + c->n;
        if (cnt)
            hipLaunchKernelGGL(k_unpack_dp, dim3(ceil_div(cnt, 256u)), dim3(256), 0, c->stream,
                               (const float2*)(side == 0 ? lo_dev : hi_dev), c->dp + at, c->cw + at, cnt, c->phys);
    }
    SPH_HIP(hipGetLastError());
    return SPH_OK;
}

int sph_layer_histogram(sph_ctx* c, uint32_t* hist, uint32_t n_layers) {
    SPH_REQUIRE(c && hist, SPH_E_INVALID, "null argument");
    SPH_REQUIRE(c->stage >= sph_ctx::ST_SORTED, SPH_E_STATE, "sph_layer_histogram needs sph_sort first");
    SPH_REQUIRE(n_layers == c->grid.g[2], SPH_E_INVALID, "n_layers must be the global z grid size %u", c->grid.g[2]);
    const uint32_t layer = c->grid.g[0] * c->grid.g[1];
    for (uint32_t l = 0; l < n_layers; l++) hist[l] = 0;
    // local layer l holds global layer l + z_off
    uint32_t prev = 0;
    for (uint32_t ll0 = 0; ll0 < c->grid.zl; ll0 += 30) {
        uint32_t m = c->grid.zl - ll0 < 30 ? c->grid.zl - ll0 : 30, tg[31], lb[31];
        for (uint32_t k = 0; k < m; k++) tg[k] = (ll0 + k + 1) * layer;
        int rc = lower_bounds(c, tg, m, lb);
        if (rc) return rc;
        for (uint32_t k = 0; k < m; k++) {
            int64_t gl = (int64_t)c->grid.z_off + ll0 + k;
            if (gl >= 0 && gl < (int64_t)n_layers) hist[gl] = lb[k] - prev;
            prev = lb[k];
        }
    }
    return SPH_OK;
}

}  // extern "C"
