/* include/sph_compat_seam.h -- the reference's OWN extern "C" seam, exported by libsph_hip.so.
 *
 * These are the 19 symbols of /root/reference/SPH/particleSystem.cuh:3-30 (definitions
 * SPH/particleSystem.cu:422-535) with the reference's signatures, so that the reference's host
 * class (SPH/particleSystem.cpp, compiled unmodified) links against libsph_hip.so INSTEAD of its
 * CUDA object and runs its CUDA_PARALLEL branch (particleSystem.cpp:773-795) on an MI355X.
 *
 * How it maps onto the native library (include/sph_hip.h): the first cudaMapZIndex on a particle
 * array creates an sph_ctx from the SimParams it reads back from the device pointer; every seam
 * call runs the corresponding sph_* phase on the context's sorted SoA state and then writes the
 * fields the reference kernel would have written back into the caller's 88-byte AoS array, so the
 * array always looks the way the reference would have left it (sorted by cell, fields updated).
 * Every integer a caller can see is the reference's: Particle::zindex is the Morton code
 * coord2zIndex(cell) (particleSystem.cu:68-91), cudaSortParticles leaves the array sorted by it,
 * cudaConstructBGrid fills dev_B[zindex] = {nParticles, start} and cudaConstructGridArray fills
 * dev_B_prime with one {nParticles <= 32, start} entry per chunk and hands its size back
 * (particleSystem.cu:311-373, 503-528) -- array for array what the reference's own code leaves there
 * (tests/test_gpu_dropin.py).  The native context underneath keeps its row-major cell key and its own
 * cell table; the seam holds the permutation between the two orders.  Only the order of the particles
 * INSIDE one cell is this library's (stable); the reference's is what thrust::sort leaves, unspecified.
 * The AoS round trips make this path slower than the native one; it exists for drop-in
 * verification.  GL interop is headless: the "VBO" handed to cudaIntegrate is a device buffer.
 */
#ifndef SPH_COMPAT_SEAM_H
#define SPH_COMPAT_SEAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* byte-compatible with SPH/particles_kernel.cuh:52-65 (Vector3f = 3 packed floats), 88 bytes */
typedef struct sph_compat_particle {
    uint32_t index;
    float position[3];
    float velocity[3];
    float delta_velocity[3];
    float force_press[3];
    float force_visc[3];
    float mass;
    float density;
    float pressure;
    float radius;
    int32_t collision_count;
    uint32_t zindex;
} sph_compat_particle;

/* SPH/particles_kernel.cuh:73-76 */
typedef struct sph_compat_grid_item { uint32_t nParticles; uint32_t start; } sph_compat_grid_item;

/* SPH/particles_kernel.cuh:36-50 (float3 = 3 floats, 72 bytes) */
typedef struct sph_compat_simparams {
    float colliderPos[3]; float colliderRadius;
    float gravity[3];     float particleRadius;
    float boxMin[3];      float boxMax[3];
    float boxDims[3];     uint32_t gridDim;
} sph_compat_simparams;

struct cudaGraphicsResource;   /* opaque, as in the reference */

#ifndef SPH_COMPAT_NO_PROTOTYPES   /* the reference's own particleSystem.cuh declares the same names */
unsigned int iceildiv(unsigned int num, unsigned int denom);                          /* .cu:423 */
void cudaInit(int argc, char** argv);                                                  /* .cu:427 */
void allocateArray(void** devPtr, size_t size);                                        /* .cu:438 */
void freeArray(void* devPtr);                                                          /* .cu:442 */
void registerGLBufferObject(unsigned int vbo, struct cudaGraphicsResource** res);      /* .cu:446 */
void unregisterGLBufferObject(struct cudaGraphicsResource* res);                       /* .cu:450 */
void* mapGLBufferObject(struct cudaGraphicsResource** res);                            /* .cu:454 */
void unmapGLBufferObject(struct cudaGraphicsResource* res);                            /* .cu:463 */
void threadSync(void);                                                                 /* .cu:467 */
void copyArrayFromDevice(void* host, const void* device, size_t size);                 /* .cu:471 */
void copyArrayToDevice(void* device, const void* host, size_t size);                   /* .cu:475 */
void cudaComputeDensities(sph_compat_particle* p, unsigned int n, sph_compat_grid_item* B, unsigned int b_size,
                          sph_compat_grid_item* Bp, unsigned int Bp_size, sph_compat_simparams* params);   /* .cu:479 */
void cudaComputeForces(sph_compat_particle* p, unsigned int n, sph_compat_grid_item* B, unsigned int b_size,
                       sph_compat_grid_item* Bp, unsigned int Bp_size, sph_compat_simparams* params);      /* .cu:483 */
void cudaParticleCollisions(sph_compat_particle* p, unsigned int n, sph_compat_grid_item* B, unsigned int b_size,
                            sph_compat_grid_item* Bp, unsigned int Bp_size, sph_compat_simparams* params); /* .cu:487 */
void cudaMapZIndex(sph_compat_particle* p, unsigned int n, sph_compat_simparams* params);                   /* .cu:491 */
void cudaSortParticles(sph_compat_particle* p, unsigned int n);                                             /* .cu:497 */
void cudaConstructBGrid(sph_compat_particle* p, unsigned int n, sph_compat_grid_item* B, unsigned int b_size,
                        sph_compat_simparams* params);                                                      /* .cu:503 */
void cudaConstructGridArray(sph_compat_particle* p, unsigned int n, sph_compat_grid_item* B, unsigned int b_size,
                            sph_compat_grid_item** Bp, unsigned int* Bp_size /* host */,
                            sph_compat_simparams* params);                                                  /* .cu:511 */
void cudaIntegrate(float* gl_pos, float deltaTime, sph_compat_particle* p, unsigned int n,
                   sph_compat_simparams* params);                                                           /* .cu:530 */
#endif

/* additive: the native context behind a particle array (NULL before the first cudaMapZIndex) */
struct sph_ctx;
struct sph_ctx* sph_compat_context(const void* dev_particles);
/* additive: drop that context without freeing the array (freeArray does both) -- for arrays another allocator owns */
void sph_compat_release(const void* dev_particles);
/* the headless "VBO": device float4 per creation index behind a mapped resource */
void* sph_compat_vbo_dev(struct cudaGraphicsResource* res, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif
