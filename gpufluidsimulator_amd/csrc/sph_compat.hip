// sph_compat.hip -- the reference's extern "C" seam (include/sph_compat_seam.h) on top of the
// native phases.  AoS <-> SoA conversion kernels + a registry from particle-array pointers to
// contexts.  Abort-on-error like checkCudaErrors (common/inc/helper_cuda.h:566-579).
//
// NUMBERING.  Everything a caller of the seam can see carries the REFERENCE's integers: Particle::zindex is the
// Morton code coord2zIndex(cell) (particleSystem.cu:68-91), cudaSortParticles leaves the caller's array sorted by
// it, dev_B is indexed by it and dev_B_prime lists the 32-particle chunks in that order (:311-373).  The native
// context underneath keeps its row-major cell key (the pair kernels want a cell's three x-neighbours contiguous);
// the seam holds the permutation between the two orders -- `m2n[a]` = native slot of the caller's a-th particle --
// and every write-back goes through it.  Only the order INSIDE a cell is this library's (stable); the reference's
// is whatever thrust::sort / std::sort leave, i.e. unspecified.
#include "sph_common.hpp"
#include "../../include/sph_compat_seam.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {

using namespace sph;

struct Compat {
    sph_ctx* ctx = nullptr;
    uint32_t n = 0;
    sph_compat_particle* tmp = nullptr;     // scratch for the AoS permutation
    const uint32_t* perm = nullptr;
    sph_compat_simparams prm{};             // the SimParams the context was last configured from
    uint32_t* bp_cnt = nullptr;             // B' construction: chunk starts per 256-slot tile, their scan
    uint32_t* bp_off = nullptr;
    uint32_t* bp_total = nullptr;           // pinned: number of B' entries (the seam hands it to the host)
    uint32_t* m2n = nullptr;                // caller's (Morton-sorted) position -> native slot, relative to own_off
    uint32_t* scr0 = nullptr;               // n words each: Morton keys of the native slots / the composed struct move,
    uint32_t* scr1 = nullptr;               // the permutation of the native sort
    bool sorted = false;                    // m2n is valid (between cudaSortParticles and the next cudaMapZIndex)
};

struct Vbo {
    unsigned int id = 0;
    float4* dev = nullptr;
    size_t bytes = 0;
};

std::mutex g_mu;
std::map<const void*, Compat> g_reg;

[[noreturn]] void die(const char* what) {
    fprintf(stderr, "libsph_hip compat seam: %s: %s\n", what, sph_last_error());
    exit(EXIT_FAILURE);
}
#define CK(call) do { if ((call) < 0) die(#call); } while (0)
#define CKH(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { fprintf(stderr, "libsph_hip compat seam: %s: %s\n", #call, hipGetErrorString(e__)); exit(EXIT_FAILURE); } } while (0)

enum Field { F_ZINDEX = 1, F_DENS = 2, F_FORCE = 4, F_COLL = 8, F_STATE = 16 };

// AoS -> SoA: position, velocity, creation index (the state the native phases need)
__global__ __launch_bounds__(256) void k_aos_unpack(const sph_compat_particle* __restrict__ p, uint32_t n,
                                                    float4* __restrict__ posi, float4* __restrict__ velr) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const sph_compat_particle q = p[i];
    posi[i] = make_float4(q.position[0], q.position[1], q.position[2], __uint_as_float(q.index));
    velr[i] = make_float4(q.velocity[0], q.velocity[1], q.velocity[2], 0.f);
}

// coord2zIndex of the reference (particleSystem.cu:68-91): the bits of x, y, z (10 each) interleaved, x lowest
__host__ __device__ inline uint32_t spread3(uint32_t v) {
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// native whole-domain key (z * gy + y) * gx + x  ->  Morton code of the same cell
__device__ __forceinline__ uint32_t morton_of_key(uint32_t key, uint32_t gx, uint32_t gy) {
    const uint32_t x = key % gx, r = key / gx;
    return spread3(x) | (spread3(r % gy) << 1) | (spread3(r / gy) << 2);
}

__global__ __launch_bounds__(256) void k_compat_morton(const uint32_t* __restrict__ key, uint32_t n, uint32_t gx, uint32_t gy,
                                                       uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = morton_of_key(key[i], gx, gy);
}

// the struct move of cudaSortParticles in one gather: q[a] = perm[m2n[a]] (perm: sorted native slot -> slot before the
// native sort = the caller's index on entry; null = identity)
__global__ __launch_bounds__(256) void k_compat_compose(const uint32_t* __restrict__ m2n, const uint32_t* __restrict__ perm,
                                                        uint32_t n, uint32_t* __restrict__ q) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a < n) q[a] = perm ? perm[m2n[a]] : m2n[a];
}

// SoA -> AoS: the fields one reference kernel would have written.  The caller's particle a sits in native slot
// m2n[a] (null before the sort: the native arrays still are in the caller's order).
__global__ __launch_bounds__(256) void k_aos_writeback(sph_compat_particle* __restrict__ p, uint32_t n, int fields,
                                                       const float4* __restrict__ posi, const float4* __restrict__ velr,
                                                       const float2* __restrict__ dp, const float4* __restrict__ fp,
                                                       const float4* __restrict__ fv, const float4* __restrict__ dv,
                                                       const uint32_t* __restrict__ key, const uint32_t* __restrict__ m2n,
                                                       uint32_t gx, uint32_t gy) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= n) return;
    sph_compat_particle* q = &p[a];
    const uint32_t i = m2n ? m2n[a] : a;
    if (fields & F_ZINDEX) q->zindex = morton_of_key(key[i], gx, gy);
    if (fields & F_DENS) { float2 d = dp[i]; q->density = d.x; q->pressure = d.y; }
    if (fields & F_FORCE) {
        float4 a = fp[i], b = fv[i];
        q->force_press[0] = a.x; q->force_press[1] = a.y; q->force_press[2] = a.z;
        q->force_visc[0] = b.x; q->force_visc[1] = b.y; q->force_visc[2] = b.z;
    }
    if (fields & F_COLL) {
        float4 d = dv[i];
        q->delta_velocity[0] = d.x; q->delta_velocity[1] = d.y; q->delta_velocity[2] = d.z;
        q->collision_count = (int32_t)__float_as_uint(d.w);
    }
    if (fields & F_STATE) {
        float4 a = posi[i], b = velr[i];
        q->position[0] = a.x; q->position[1] = a.y; q->position[2] = a.z;
        q->velocity[0] = b.x; q->velocity[1] = b.y; q->velocity[2] = b.z;
    }
}

// the struct move of thrust::sort: out[i] = in[perm[i]], 22 dwords per particle
__global__ __launch_bounds__(256) void k_aos_permute(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     const uint32_t* __restrict__ perm, uint32_t n) {
    uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t i = t / 22u, w = t % 22u;
    if (i >= n) return;
    out[(size_t)i * 22u + w] = in[(size_t)perm[i] * 22u + w];
}

// ---- B and B' of the reference (particleSystem.cu:311-373, 503-528), derived from the native {start, end}
// cell table.  B[zindex] = {nParticles, start} per occupied cell (the caller's array is zeroed first, like the
// cudaMemset at :506); B' = one {nParticles <= 32, start} entry per GRID_COMPACT_WIDTH = 32 chunk of every
// occupied cell, in the order of the sorted particle array.  `start` indexes the caller's (Morton-sorted) Particle
// array and cells carry the reference's Morton number.  A cell's particles are contiguous in both orders and the
// Morton sort of the native slots is stable, so a particle's rank inside its cell is the same in both.
__global__ __launch_bounds__(256) void k_compat_B(const uint32_t* __restrict__ key, const uint2* __restrict__ cells,
                                                  const uint32_t* __restrict__ m2n, uint32_t n, uint32_t b_size,
                                                  uint32_t gx, uint32_t gy, sph_compat_grid_item* __restrict__ B) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= n) return;
    const uint32_t i = m2n ? m2n[a] : a, k = key[i];
    if (i == 0 || key[i - 1] != k) {                 // the first particle of its cell, in both orders
        const uint32_t z = morton_of_key(k, gx, gy);
        if (z < b_size) {
            const uint2 c = cells[k];
            B[z].nParticles = c.y - c.x;
            B[z].start = a;
        }
    }
}

// chunk starts of one 256-slot tile: pass 0 counts them, pass 1 writes the entries at their scanned offsets
template <int WRITE>
__global__ __launch_bounds__(256) void k_compat_Bprime(const uint32_t* __restrict__ key, const uint2* __restrict__ cells,
                                                       const uint32_t* __restrict__ m2n,
                                                       uint32_t n, uint32_t slot0, uint32_t* __restrict__ tile_cnt,
                                                       const uint32_t* __restrict__ tile_off,
                                                       sph_compat_grid_item* __restrict__ Bp, uint32_t bp_cap) {
    __shared__ uint32_t wcnt[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;              // the caller's (Morton-sorted) position
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    bool begins = false;
    uint32_t cnt = 0;
    if (i < n) {
        const uint32_t s = m2n ? m2n[i] : i;
        const uint2 c = cells[key[s]];
        const uint32_t local = s - (c.x - slot0);
        begins = (local & 31u) == 0u;
        cnt = min(32u, (c.y - c.x) - local);
    }
    const uint64_t m = __ballot(begins);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (!WRITE) {
        if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        return;
    }
    uint32_t at = tile_off[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; w++) at += wcnt[w];
    if (begins && at < bp_cap) { Bp[at].nParticles = cnt; Bp[at].start = i; }
}

__global__ __launch_bounds__(1024) void k_compat_scan(const uint32_t* __restrict__ cnt, uint32_t nt,
                                                      uint32_t* __restrict__ off, volatile uint32_t* __restrict__ total) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nt + 1023u) / 1024u;
    const uint32_t lo = min(threadIdx.x * per, nt), hi = min(lo + per, nt);
    uint32_t s = 0;
    for (uint32_t t = lo; t < hi; t++) s += cnt[t];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = threadIdx.x >= (uint32_t)o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t t = lo; t < hi; t++) { off[t] = run; run += cnt[t]; }
    if (threadIdx.x == 1023) *total = part[1023];
}

// (re)configure the context of a particle array from the caller's device-resident SimParams; the reference
// uploads them before every update() (particleSystem.cpp:723), so they are looked at on every cudaMapZIndex
void configure(Compat& c, unsigned int n, const sph_compat_simparams* params_dev) {
    sph_compat_simparams h;
    CKH(hipMemcpy(&h, params_dev, sizeof(h), hipMemcpyDeviceToHost));   // SimParams* is a DEVICE pointer in the seam
    const bool same_grid = c.ctx && c.n == n && c.prm.gridDim == h.gridDim;
    if (same_grid && memcmp(&c.prm, &h, sizeof(h)) == 0) return;
    const uint32_t grid[3] = {h.gridDim, h.gridDim, h.gridDim};
    sph_params q;
    sph_default_params(&q, h.boxDims, grid);
    for (int a = 0; a < 3; a++) { q.box_min[a] = h.boxMin[a]; q.box_max[a] = h.boxMax[a]; }
    q.particle_radius = h.particleRadius;
    if (same_grid) {
        CK(sph_set_params(c.ctx, &q));       // box or radius changed between two updates
    } else {
        if (c.ctx) { sph_destroy(c.ctx); c.ctx = nullptr; }
        if (c.tmp) { hipFree(c.tmp); c.tmp = nullptr; }
        if (c.bp_cnt) { hipFree(c.bp_cnt); c.bp_cnt = nullptr; }
        if (c.bp_off) { hipFree(c.bp_off); c.bp_off = nullptr; }
        if (c.m2n) { hipFree(c.m2n); c.m2n = nullptr; }
        if (c.scr0) { hipFree(c.scr0); c.scr0 = nullptr; }
        if (c.scr1) { hipFree(c.scr1); c.scr1 = nullptr; }
        if (h.gridDim > 1024u) {
            fprintf(stderr, "libsph_hip compat seam: gridDim %u > 1024: the reference's z-index holds 10 bits per axis "
                    "(particleSystem.cu:67)\n", h.gridDim);
            exit(EXIT_FAILURE);
        }
        CK(sph_create(&c.ctx, -1, n ? n : 1, &q));      // -1: the device cudaInit / sph_select_device chose
        c.ctx->keep_perm = true;                        // cudaSortParticles moves the caller's structs by it
        CKH(hipMalloc((void**)&c.tmp, (size_t)(n ? n : 1) * sizeof(sph_compat_particle)));
        const size_t nt = (size_t)ceil_div(n ? n : 1, 256u);
        CKH(hipMalloc((void**)&c.bp_cnt, nt * sizeof(uint32_t)));
        CKH(hipMalloc((void**)&c.bp_off, nt * sizeof(uint32_t)));
        CKH(hipMalloc((void**)&c.m2n, (size_t)(n ? n : 1) * sizeof(uint32_t)));
        CKH(hipMalloc((void**)&c.scr0, (size_t)(n ? n : 1) * sizeof(uint32_t)));
        CKH(hipMalloc((void**)&c.scr1, (size_t)(n ? n : 1) * sizeof(uint32_t)));
        if (!c.bp_total) CKH(hipHostMalloc((void**)&c.bp_total, sizeof(uint32_t), hipHostMallocMapped));
        c.n = n;
    }
    c.prm = h;
}

void release(Compat& c) {
    if (c.ctx) sph_destroy(c.ctx);
    if (c.tmp) hipFree(c.tmp);
    if (c.bp_cnt) hipFree(c.bp_cnt);
    if (c.bp_off) hipFree(c.bp_off);
    if (c.m2n) hipFree(c.m2n);
    if (c.scr0) hipFree(c.scr0);
    if (c.scr1) hipFree(c.scr1);
    if (c.bp_total) hipHostFree(c.bp_total);
    c = Compat();
}

Compat& lookup(const void* p, const char* who) {
    auto it = g_reg.find(p);
    if (it == g_reg.end()) {
        fprintf(stderr, "libsph_hip compat seam: %s called before cudaMapZIndex on this particle array\n", who);
        exit(EXIT_FAILURE);
    }
    return it->second;
}

void writeback(Compat& c, sph_compat_particle* p, int fields) {
    sph_ctx* x = c.ctx;
    const uint32_t o = x->own_off;
    const bool hashed = x->stage == sph_ctx::ST_HASHED;      // before the sort: native slot = the caller's index
    hipLaunchKernelGGL(k_aos_writeback, dim3(ceil_div(c.n, 256)), dim3(256), 0, x->stream, p, c.n, fields, x->posi + o,
                       x->velr + o, x->dp + o, x->fpress + o, x->fvisc + o, x->dvel + o,
                       hashed ? x->k0 : x->keyS + o, (hashed || !c.sorted) ? (const uint32_t*)nullptr : c.m2n, x->grid.g[0],
                       x->grid.g[1]);
    CKH(hipGetLastError());
}

}  // namespace

namespace sph {
const uint32_t* last_sort_permutation(sph_ctx* c);
int sort_indices_by_key(sph_ctx* c, const uint32_t* keys_dev, uint32_t n, uint32_t bits, const uint32_t** perm_out);
}

extern "C" {

unsigned int iceildiv(unsigned int num, unsigned int denom) { return (num % denom == 0) ? num / denom : 1 + (num / denom); }

void cudaInit(int argc, char** argv) {
    int is950 = 0;
    if (sph_device_count(&is950) <= 0) {
        printf("No gfx950 (MI355X) devices found, exiting\n");
        exit(EXIT_SUCCESS);     // as particleSystem.cu:432-435
    }
    // findCudaDevice (common/inc/helper_cuda.h:845): `-device=N` (any number of leading dashes) picks the device
    int device = 0;
    for (int i = 1; i < argc && argv; i++) {
        const char* a = argv[i];
        if (!a) continue;
        while (*a == '-') a++;
        if (!strncmp(a, "device=", 7)) device = atoi(a + 7);
    }
    if (sph_select_device(device) < 0) {
        fprintf(stderr, "cudaInit: %s\n", sph_last_error());
        exit(EXIT_FAILURE);
    }
}

void allocateArray(void** devPtr, size_t size) { CKH(hipMalloc(devPtr, size ? size : 1)); }

void freeArray(void* devPtr) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_reg.find(devPtr);
        if (it != g_reg.end()) {
            release(it->second);
            g_reg.erase(it);
        }
    }
    CKH(hipFree(devPtr));
}

void registerGLBufferObject(unsigned int vbo, struct cudaGraphicsResource** res) {
    Vbo* v = new Vbo();
    v->id = vbo;
    *res = reinterpret_cast<cudaGraphicsResource*>(v);
}

void unregisterGLBufferObject(struct cudaGraphicsResource* res) {
    Vbo* v = reinterpret_cast<Vbo*>(res);
    if (!v) return;
    if (v->dev) hipFree(v->dev);
    delete v;
}

// headless: the mapped pointer is the resource handle itself; cudaIntegrate recognises it and
// (re)allocates the float4-per-particle buffer behind it
void* mapGLBufferObject(struct cudaGraphicsResource** res) { return reinterpret_cast<void*>(*res); }
void unmapGLBufferObject(struct cudaGraphicsResource*) {}

void threadSync(void) { CKH(hipDeviceSynchronize()); }
void copyArrayFromDevice(void* host, const void* device, size_t size) { CKH(hipMemcpy(host, device, size, hipMemcpyDeviceToHost)); }
void copyArrayToDevice(void* device, const void* host, size_t size) { CKH(hipMemcpy(device, host, size, hipMemcpyHostToDevice)); }

void cudaMapZIndex(sph_compat_particle* p, unsigned int n, sph_compat_simparams* params) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = g_reg[p];
    configure(c, n, params);
    sph_ctx* x = c.ctx;
    // the AoS is authoritative on entry of a step: take position, velocity and index from it
    CK(launch_cells_clear(x));
    x->own_off = x->gcap; x->n = n; x->n_glo = x->n_ghi = 0;
    if (n) {
        hipLaunchKernelGGL(k_aos_unpack, dim3(ceil_div(n, 256)), dim3(256), 0, x->stream, p, n, x->posi + x->own_off,
                           x->velr + x->own_off);
        CKH(hipGetLastError());
    }
    x->stage = sph_ctx::ST_LOADED;
    x->keys_fresh = false;
    x->order_valid = false;     // the slots no longer follow the last sort
    x->have_dens = x->have_force = x->have_coll = false;
    c.sorted = false;
    CK(sph_hash(x));
    writeback(c, p, F_ZINDEX);
}

void cudaSortParticles(sph_compat_particle* p, unsigned int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaSortParticles");
    sph_ctx* x = c.ctx;
    CK(sph_sort(x));                 // the native order: row-major cell key, stable
    if (n) {
        // the caller's order: the reference's Morton code of the same cells.  Morton keys of the native slots, a stable
        // sort of the slot numbers by them (m2n), and ONE move of the structs: caller's a <- what stood at perm[m2n[a]]
        const uint32_t* perm = sph::last_sort_permutation(x);      // sorted native slot -> the caller's index on entry
        if (perm) CKH(hipMemcpyAsync(c.scr1, perm, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, x->stream));
        hipLaunchKernelGGL(k_compat_morton, dim3(ceil_div(n, 256)), dim3(256), 0, x->stream, x->keyS + x->own_off, n, x->grid.g[0],
                           x->grid.g[1], c.scr0);
        CKH(hipGetLastError());
        const uint32_t* order = nullptr;
        CK(sph::sort_indices_by_key(x, c.scr0, n, 30u, &order));
        CKH(hipMemcpyAsync(c.m2n, order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, x->stream));
        hipLaunchKernelGGL(k_compat_compose, dim3(ceil_div(n, 256)), dim3(256), 0, x->stream, c.m2n, perm ? c.scr1 : nullptr, n, c.scr0);
        hipLaunchKernelGGL(k_aos_permute, dim3(ceil_div(n * 22u, 256)), dim3(256), 0, x->stream, (const uint32_t*)p,
                           (uint32_t*)c.tmp, c.scr0, n);
        CKH(hipGetLastError());
        CKH(hipMemcpyAsync(p, c.tmp, (size_t)n * sizeof(sph_compat_particle), hipMemcpyDeviceToDevice, x->stream));
        c.sorted = true;
    }
    CKH(hipStreamSynchronize(x->stream));    // thrust::sort blocks the host; keep that
}

void cudaConstructBGrid(sph_compat_particle* p, unsigned int n, sph_compat_grid_item* B, unsigned int B_size,
                        sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaConstructBGrid");
    sph_ctx* x = c.ctx;
    CK(sph_build_cells(x));
    if (B && B_size) {     // the caller's table, as kernelConstructBGrid leaves it (particleSystem.cu:503-509)
        CKH(hipMemsetAsync(B, 0, (size_t)B_size * sizeof(sph_compat_grid_item), x->stream));
        if (n) {
            hipLaunchKernelGGL(k_compat_B, dim3(ceil_div(n, 256)), dim3(256), 0, x->stream, x->keyS + x->own_off, x->cells,
                               c.sorted ? c.m2n : (const uint32_t*)nullptr, n, B_size, x->grid.g[0], x->grid.g[1], B);
            CKH(hipGetLastError());
        }
    }
}

void cudaConstructGridArray(sph_compat_particle* p, unsigned int n, sph_compat_grid_item*, unsigned int,
                            sph_compat_grid_item** Bp, unsigned int* Bp_size, sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaConstructGridArray");
    sph_ctx* x = c.ctx;
    // The native kernels schedule by waves of 64 sorted particles and do not read B'; it is produced for the
    // caller, who owns the array (capacity: one entry per particle, particleSystem.cpp:124) and gets its size
    // back through a host pointer (a blocking copy in the reference, particleSystem.cu:520-524).
    uint32_t total = 0;
    if (n) {
        const uint32_t nt = ceil_div(n, 256u);
        sph_compat_grid_item* out = Bp ? *Bp : nullptr;
        hipLaunchKernelGGL(k_compat_Bprime<0>, dim3(nt), dim3(256), 0, x->stream, x->keyS + x->own_off, x->cells,
                           c.sorted ? c.m2n : (const uint32_t*)nullptr, n, x->own_off, c.bp_cnt, c.bp_off, out, n);
        hipLaunchKernelGGL(k_compat_scan, dim3(1), dim3(1024), 0, x->stream, c.bp_cnt, nt, c.bp_off, c.bp_total);
        if (out)
            hipLaunchKernelGGL(k_compat_Bprime<1>, dim3(nt), dim3(256), 0, x->stream, x->keyS + x->own_off, x->cells,
                               c.sorted ? c.m2n : (const uint32_t*)nullptr, n, x->own_off, c.bp_cnt, c.bp_off, out, n);
        CKH(hipGetLastError());
        CKH(hipStreamSynchronize(x->stream));
        total = *c.bp_total;
    }
    if (Bp_size) *Bp_size = total;
}

void cudaComputeDensities(sph_compat_particle* p, unsigned int, sph_compat_grid_item*, unsigned int, sph_compat_grid_item*,
                          unsigned int, sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaComputeDensities");
    CK(sph_density(c.ctx));
    writeback(c, p, F_DENS);
}

void cudaComputeForces(sph_compat_particle* p, unsigned int, sph_compat_grid_item*, unsigned int, sph_compat_grid_item*,
                       unsigned int, sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaComputeForces");
    CK(sph_force(c.ctx));
    writeback(c, p, F_FORCE);
}

void cudaParticleCollisions(sph_compat_particle* p, unsigned int, sph_compat_grid_item*, unsigned int, sph_compat_grid_item*,
                            unsigned int, sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaParticleCollisions");
    CK(sph_collide(c.ctx));
    writeback(c, p, F_COLL);
}

void cudaIntegrate(float* gl_pos, float deltaTime, sph_compat_particle* p, unsigned int n, sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaIntegrate");
    CK(sph_integrate(c.ctx, deltaTime));
    writeback(c, p, F_STATE);
    // gl_pos: (x,y,z,1) per creation index (particleSystem.cu:416-419) = the context's pos_out
    Vbo* v = reinterpret_cast<Vbo*>(gl_pos);
    const size_t bytes = (size_t)n * sizeof(float4);
    if (v->bytes < bytes) {
        if (v->dev) hipFree(v->dev);
        CKH(hipMalloc((void**)&v->dev, bytes ? bytes : 16));
        v->bytes = bytes;
    }
    if (n) CKH(hipMemcpyAsync(v->dev, c.ctx->pos_out, bytes, hipMemcpyDeviceToDevice, c.ctx->stream));
}

struct sph_ctx* sph_compat_context(const void* dev_particles) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_reg.find(dev_particles);
    return it == g_reg.end() ? nullptr : it->second.ctx;
}

// drop the context behind a particle array WITHOUT freeing the array (freeArray does both: the reference's own pairing
// with allocateArray); for callers whose array belongs to another allocator
void sph_compat_release(const void* dev_particles) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_reg.find(dev_particles);
    if (it == g_reg.end()) return;
    release(it->second);
    g_reg.erase(it);
}

void* sph_compat_vbo_dev(struct cudaGraphicsResource* res, size_t* bytes) {
    Vbo* v = reinterpret_cast<Vbo*>(res);
    if (bytes) *bytes = v ? v->bytes : 0;
    return v ? v->dev : nullptr;
}

}  // extern "C"
