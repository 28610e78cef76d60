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

void cudaInit(int, char**) {
    int is950 = 0;
    if (sph_device_count(&is950) <= 0 || !is950) {
        printf("No gfx950 (MI355X) devices found, exiting\n");
        exit(EXIT_SUCCESS);     // as particleSystem.cu:432-435
    }
}

void allocateArray(void** devPtr, size_t size) { CKH(hipMalloc(devPtr, size ? size : 1)); }

void freeArray(void* devPtr) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_reg.find(devPtr);
        if (it != g_reg.end()) {
            sph_destroy(it->second.ctx);
            if (it->second.tmp) hipFree(it->second.tmp);
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
    if (!c.ctx || c.n != n) {
        if (c.ctx) { sph_destroy(c.ctx); c.ctx = nullptr; }
        if (c.tmp) { hipFree(c.tmp); c.tmp = nullptr; }
        sph_compat_simparams h;
        CKH(hipMemcpy(&h, params, sizeof(h), hipMemcpyDeviceToHost));   // SimParams* is a DEVICE pointer in the seam
        const uint32_t grid[3] = {h.gridDim, h.gridDim, h.gridDim};
        sph_params q;
        sph_default_params(&q, h.boxDims, grid);
        for (int a = 0; a < 3; a++) { q.box_min[a] = h.boxMin[a]; q.box_max[a] = h.boxMax[a]; }
        q.particle_radius = h.particleRadius;
        CK(sph_create(&c.ctx, 0, n ? n : 1, &q));
        CKH(hipMalloc((void**)&c.tmp, (size_t)(n ? n : 1) * sizeof(sph_compat_particle)));
        c.n = n;
    }
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

void cudaConstructBGrid(sph_compat_particle* p, unsigned int, sph_compat_grid_item*, unsigned int, sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    CK(sph_build_cells(lookup(p, "cudaConstructBGrid").ctx));
}

void cudaConstructGridArray(sph_compat_particle* p, unsigned int n, sph_compat_grid_item*, unsigned int,
                            sph_compat_grid_item**, unsigned int* Bp_size, sph_compat_simparams*) {
    std::lock_guard<std::mutex> lk(g_mu);
    lookup(p, "cudaConstructGridArray");
    // B' is the reference's work list; this library schedules by waves of 64 sorted particles
    // instead.  The host only hands the value back to the seam.
    if (Bp_size) *Bp_size = iceildiv(n, 32u);
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
