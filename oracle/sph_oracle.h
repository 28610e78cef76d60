/* oracle/sph_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the SPH step of oadrian/GPUFluidSimulator (the z* / OMP
 * path of SPH/particleSystem.cpp, which is textually the same algorithm as the
 * CUDA kernels of SPH/particleSystem.cu).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this.  Parity status: PINNED --
 * checked bit for bit, phase by phase, against the reference's own code
 * compiled in the build container (oracle/_ref/sph_ref, see oracle/Makefile and
 * tests/test_oracle_vs_ref.py) and through the fixtures in tests/golden/ that
 * oracle/make_golden.py generated from that binary.
 */
#ifndef SPH_ORACLE_H
#define SPH_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same field order and size (88 B) as `struct Particle`,
 * SPH/particles_kernel.cuh:52-65 (Vector3f == 3 packed floats) */
typedef struct {
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
} orc_particle;

/* SPH/particles_kernel.cuh:73-76 */
typedef struct { uint32_t nParticles; uint32_t start; } orc_grid_item;

enum { ORC_CELL_MORTON = 0, ORC_CELL_LINEAR = 1 };

typedef struct {
    uint32_t n;              /* m_numParticles */
    orc_particle* p;         /* m_particles (owned by the caller or by orc_create) */
    uint32_t grid[3];        /* m_h_B_dim per axis (the reference: all equal) */
    uint32_t cell_mode;      /* ORC_CELL_MORTON: reference numbering (coord2zIndex);
                                ORC_CELL_LINEAR: (z*gy+y)*gx+x, for non-cubic slabs */
    uint32_t b_size;         /* m_h_B_size */
    orc_grid_item* B;        /* m_h_B */
    uint32_t bprime_size;    /* m_h_B_prime_size */
    orc_grid_item* Bprime;   /* m_h_B_prime */
    float box_min[3], box_max[3], box_dims[3];   /* m_params.boxMin/boxMax, m_boxDims */
    float* hpos;             /* m_hPos: float4 per ORIGINAL index */
    uint32_t* occ;           /* scratch: occupied cell list (ascending cell id) */
    uint32_t n_occ;
    orc_particle* tmp;       /* scratch for the stable sort */
    uint32_t* keybuf;        /* scratch */
} orc_system;

orc_system* orc_create(uint32_t n, const float box[3], const uint32_t grid[3], uint32_t cell_mode);
void orc_destroy(orc_system* s);
/* fields as initGrid() sets them (particleSystem.cpp:854-864); index = i */
void orc_load(orc_system* s, const float* pos_xyz, const float* vel_xyz);
/* reorder the particle array so that slot k holds original index order[k] */
void orc_apply_order(orc_system* s, const uint32_t* order);

uint32_t orc_coord2zindex(uint32_t x, uint32_t y, uint32_t z);      /* particleSystem.cpp:485-508 */
void orc_zindex2coord(uint32_t zidx, uint32_t out_xyz[3]);          /* particleSystem.cpp:511-525 */
uint32_t orc_cell_of(const orc_system* s, const float pos[3]);      /* get_Z_index, :527-537 */

void orc_map_zindex(orc_system* s);            /* zMapZindex        :544-549 */
void orc_sort(orc_system* s);                  /* zSortParticles    :551-554 (stable here) */
void orc_construct_bgrid(orc_system* s);       /* zConstructBGrid   :556-578 */
void orc_construct_grid_array(orc_system* s);  /* zConstructGridArray :580-596 */
void orc_compute_densities(orc_system* s);     /* zcomputeDensities :303-322 */
void orc_compute_forces(orc_system* s);        /* zcomputeForces    :334-352 */
void orc_particle_collisions(orc_system* s);   /* zparticleCollisions :368-389 */
void orc_integrate(orc_system* s, float dt);   /* zintegrate        :437-482 */
void orc_step(orc_system* s, float dt);        /* OMP branch of update(), :743-767 */

/* O(N^2) SEQUENTIAL path (:293-301, :324-332, :354-366) -- sanity cross-check
 * only; it sums over the whole support ball and is NOT the target semantics. */
void orc_compute_densities_n2(orc_system* s);

/* gather by original index: out[index] = ... */
void orc_get_state(const orc_system* s, float* pos_xyz, float* vel_xyz, float* density, float* pressure);
void orc_get_forces(const orc_system* s, float* fpress_xyz, float* fvisc_xyz, float* dv_xyz, int32_t* count);
void orc_set_num_threads(int n);
int orc_get_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
