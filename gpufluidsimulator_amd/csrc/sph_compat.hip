// sph_compat.hip -- the reference's extern "C" seam (include/sph_compat_seam.h) on top of the
// native phases.  AoS <-> SoA conversion kernels + a registry from particle-array pointers to
// contexts.  Abort-on-error like checkCudaErrors (common/inc/helper_cuda.h:566-579).
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

// SoA -> AoS: the fields one reference kernel would have written
__global__ __launch_bounds__(256) void k_aos_writeback(sph_compat_particle* __restrict__ p, uint32_t n, int fields,
                                                       const float4* __restrict__ posi, const float4* __restrict__ velr,
                                                       const float2* __restrict__ dp, const float4* __restrict__ fp,
                                                       const float4* __restrict__ fv, const float4* __restrict__ dv,
                                                       const uint32_t* __restrict__ key) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    sph_compat_particle* q = &p[i];
    if (fields & F_ZINDEX) q->zindex = key[i];
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
// cell table.  B[key] = {nParticles, start} per occupied cell (the caller's array is zeroed first, like the
// cudaMemset at :506); B' = one {nParticles <= 32, start} entry per GRID_COMPACT_WIDTH = 32 chunk of every
// occupied cell, in the order of the sorted particle array.  `start` indexes the caller's (sorted) Particle
// array.  Cells are numbered with this library's row-major key -- the same numbers cudaMapZIndex wrote into
// Particle::zindex -- not with the reference's Morton code ("numbering").
__global__ __launch_bounds__(256) void k_compat_B(const uint32_t* __restrict__ key, const uint2* __restrict__ cells,
                                                  uint32_t n, uint32_t slot0, uint32_t b_size,
                                                  sph_compat_grid_item* __restrict__ B) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = key[i];
    if ((i == 0 || key[i - 1] != k) && k < b_size) {
        const uint2 c = cells[k];
        B[k].nParticles = c.y - c.x;
        B[k].start = c.x - slot0;
    }
}

// chunk starts of one 256-slot tile: pass 0 counts them, pass 1 writes the entries at their scanned offsets
template <int WRITE>
__global__ __launch_bounds__(256) void k_compat_Bprime(const uint32_t* __restrict__ key, const uint2* __restrict__ cells,
                                                       uint32_t n, uint32_t slot0, uint32_t* __restrict__ tile_cnt,
                                                       const uint32_t* __restrict__ tile_off,
                                                       sph_compat_grid_item* __restrict__ Bp, uint32_t bp_cap) {
    __shared__ uint32_t wcnt[4];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    bool begins = false;
    uint32_t cnt = 0;
    if (i < n) {
        const uint2 c = cells[key[i]];
        const uint32_t local = i - (c.x - slot0);
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
        CK(sph_create(&c.ctx, -1, n ? n : 1, &q));      // -1: the device cudaInit / sph_select_device chose
        c.ctx->keep_perm = true;                        // cudaSortParticles moves the caller's structs by it
        CKH(hipMalloc((void**)&c.tmp, (size_t)(n ? n : 1) * sizeof(sph_compat_particle)));
        const size_t nt = (size_t)ceil_div(n ? n : 1, 256u);
        CKH(hipMalloc((void**)&c.bp_cnt, nt * sizeof(uint32_t)));
        CKH(hipMalloc((void**)&c.bp_off, nt * sizeof(uint32_t)));
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
    hipLaunchKernelGGL(k_aos_writeback, dim3(ceil_div(c.n, 256)), dim3(256), 0, x->stream, p, c.n, fields, x->posi + o,
                       x->velr + o, x->dp + o, x->fpress + o, x->fvisc + o, x->dvel + o,
                       x->stage == sph_ctx::ST_HASHED ? x->k0 : x->keyS + o);
    CKH(hipGetLastError());
}

}  // namespace

namespace sph { const uint32_t* last_sort_permutation(sph_ctx* c); }

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
    CK(sph_hash(x));
    writeback(c, p, F_ZINDEX);
}

void cudaSortParticles(sph_compat_particle* p, unsigned int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    Compat& c = lookup(p, "cudaSortParticles");
    CK(sph_sort(c.ctx));
    if (n) {   // move the structs like thrust::sort does
        const uint32_t* perm = sph::last_sort_permutation(c.ctx);
        if (perm)                                   // null: the order did not change
        hipLaunchKernelGGL(k_aos_permute, dim3(ceil_div(n * 22u, 256)), dim3(256), 0, c.ctx->stream, (const uint32_t*)p,
                           (uint32_t*)c.tmp, perm, n);
        CKH(hipGetLastError());
        if (perm) CKH(hipMemcpyAsync(p, c.tmp, (size_t)n * sizeof(sph_compat_particle), hipMemcpyDeviceToDevice, c.ctx->stream));
    }
    CKH(hipStreamSynchronize(c.ctx->stream));    // thrust::sort blocks the host; keep that
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
            hipLaunchKernelGGL(k_compat_B, dim3(ceil_div(n, 256)), dim3(256), 0, x->stream, x->keyS + x->own_off, x->cells, n,
                               x->own_off, B_size, B);
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
        hipLaunchKernelGGL(k_compat_Bprime<0>, dim3(nt), dim3(256), 0, x->stream, x->keyS + x->own_off, x->cells, n, x->own_off,
                           c.bp_cnt, c.bp_off, out, n);
        hipLaunchKernelGGL(k_compat_scan, dim3(1), dim3(1024), 0, x->stream, c.bp_cnt, nt, c.bp_off, c.bp_total);
        if (out)
            hipLaunchKernelGGL(k_compat_Bprime<1>, dim3(nt), dim3(256), 0, x->stream, x->keyS + x->own_off, x->cells, n,
                               x->own_off, c.bp_cnt, c.bp_off, out, n);
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

void* sph_compat_vbo_dev(struct cudaGraphicsResource* res, size_t* bytes) {
    Vbo* v = reinterpret_cast<Vbo*>(res);
    if (bytes) *bytes = v ? v->bytes : 0;
    return v ? v->dev : nullptr;
}

}  // extern "C"
