/* oracle/sph_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the SPH step of oadrian/GPUFluidSimulator: the z* (OMP)
 * path of /root/reference/SPH/particleSystem.cpp, which states the same
 * algorithm as the CUDA kernels in SPH/particleSystem.cu.  Every function cites
 * the reference lines it follows.  Floating point: single precision, ONE
 * rounding per operation in the operation order the reference's Eigen
 * expressions evaluate to (build with -ffp-contract=off, no -march), so that
 * given the same particle order the results are bit-identical to the reference
 * binary oracle/_ref/sph_ref (tests/test_oracle_vs_ref.py).
 *
 * Parity status: PINNED (see sph_oracle.h).  Differences from the reference by
 * design: (1) the sort is stable (the reference's std::sort leaves the order
 * of particles within one cell unspecified); (2) ORC_CELL_LINEAR numbering and
 * per-axis grid sizes exist for slab tests; (3) parallel loops run over the
 * occupied-cell list instead of all cells (same per-particle arithmetic).
 */
#include "sph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* SPH/particles_kernel.cuh:20-33 */
#define REST_DENS 1000.f
#define GAS_CONSTANT 2000.f
#define m_H 0.1f
#define HSQ (m_H * m_H)
#define MASS 65.f
#define VISC 250.f
#define GRAVITY -9.81f
#define G_MODIFIER 11000
#define PI_F 3.141592654f
#define EPS_F 0.00001f
#define RESTITUTION 0.f
#define COLLISION_PARAM 1.0
#define GRID_COMPACT_WIDTH 32u

/* ---- Eigen 3.3.7 Vector3f semantics used by the reference ------------------
 * squaredNorm()/dot() reduce a fixed-size 3-vector with redux_novec_unroller
 * (Eigen/src/Core/Redux.h): Length 3 splits into (0) and (1,2), i.e.
 * a0 + (a1 + a2).  norm() = sqrt(squaredNorm()) (Dot.h:105-109).
 * normalized() (Dot.h:124-134): z = squaredNorm(); z > 0 ? v / sqrt(z) : v. */
static inline float sqn3(const float v[3]) { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }
static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

/* ---- Morton ----------------------------------------------------------------*/
/* ParticleSystem::coord2zIndex, particleSystem.cpp:485-508 (== .cu:68-91) */
static inline uint32_t spread10(uint32_t v) {
    v = (v | (v << 16)) & 0x030000FF;
    v = (v | (v << 8)) & 0x0300F00F;
    v = (v | (v << 4)) & 0x030C30C3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}
uint32_t orc_coord2zindex(uint32_t x, uint32_t y, uint32_t z) {
    return spread10(x) | (spread10(y) << 1) | (spread10(z) << 2);
}
/* collapseEvery3 + zIndex2coord, particleSystem.cpp:511-525 (== .cu:105-124) */
static inline uint32_t collapse_every3(uint32_t x) {
    uint32_t res = 0;
    for (int i = 0; i < 10; i++) res |= ((x >> (i * 3)) & 1u) << i;
    return res;
}
void orc_zindex2coord(uint32_t zidx, uint32_t out[3]) {
    out[0] = collapse_every3(zidx);
    out[1] = collapse_every3(zidx >> 1);
    out[2] = collapse_every3(zidx >> 2);
}

static inline uint32_t cell_id(const orc_system* s, uint32_t x, uint32_t y, uint32_t z) {
    if (s->cell_mode == ORC_CELL_MORTON) return orc_coord2zindex(x, y, z);
    return (z * s->grid[1] + y) * s->grid[0] + x;
}
static inline void cell_coord(const orc_system* s, uint32_t c, uint32_t out[3]) {
    if (s->cell_mode == ORC_CELL_MORTON) { orc_zindex2coord(c, out); return; }
    out[0] = c % s->grid[0];
    out[1] = (c / s->grid[0]) % s->grid[1];
    out[2] = c / (s->grid[0] * s->grid[1]);
}

/* ParticleSystem::get_Z_index, particleSystem.cpp:527-537 (== .cu:93-103):
 * subtract boxMin, DIVIDE by boxDims, THEN multiply by the grid size, floor. */
uint32_t orc_cell_of(const orc_system* s, const float pos[3]) {
    uint32_t c[3];
    for (int a = 0; a < 3; a++) {
        float rel = pos[a] - s->box_min[a];
        float q = (rel / s->box_dims[a]) * (float)s->grid[a];
        c[a] = (uint32_t)(int)floor(q);
    }
    return cell_id(s, c[0], c[1], c[2]);
}

/* ---- lifetime ---------------------------------------------------------------*/
static uint32_t next_pow2(uint32_t x) {   /* particleSystem.h:34-43 */
    x--; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; x++;
    return x;
}

orc_system* orc_create(uint32_t n, const float box[3], const uint32_t grid[3], uint32_t cell_mode) {
    orc_system* s = (orc_system*)calloc(1, sizeof(orc_system));
    if (!s) return NULL;
    s->n = n;
    s->cell_mode = cell_mode;
    for (int a = 0; a < 3; a++) {
        s->grid[a] = grid[a];
        s->box_dims[a] = box[a];
        s->box_min[a] = -box[a] / 2;      /* particleSystem.cpp:55-60 */
        s->box_max[a] = box[a] / 2;
    }
    if (cell_mode == ORC_CELL_MORTON) {
        uint32_t g = grid[0] > grid[1] ? grid[0] : grid[1];
        if (grid[2] > g) g = grid[2];
        g = next_pow2(g);
        if (g > 1024) { free(s); return NULL; }   /* 10 bits per axis, .cu:67 */
        s->b_size = g * g * g;
    } else {
        s->b_size = grid[0] * grid[1] * grid[2];
    }
    s->p = (orc_particle*)calloc(n ? n : 1, sizeof(orc_particle));
    s->tmp = (orc_particle*)calloc(n ? n : 1, sizeof(orc_particle));
    s->keybuf = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    s->occ = (uint32_t*)calloc(n ? n : 1, sizeof(uint32_t));
    s->B = (orc_grid_item*)calloc(s->b_size, sizeof(orc_grid_item));
    s->hpos = (float*)calloc((size_t)(n ? n : 1) * 4, sizeof(float));
    if (!s->p || !s->tmp || !s->keybuf || !s->occ || !s->B || !s->hpos) { orc_destroy(s); return NULL; }
    return s;
}

void orc_destroy(orc_system* s) {
    if (!s) return;
    free(s->p); free(s->tmp); free(s->keybuf); free(s->occ); free(s->B); free(s->Bprime); free(s->hpos);
    free(s);
}

/* initGrid's per-particle initialisation, particleSystem.cpp:853-868 */
void orc_load(orc_system* s, const float* pos, const float* vel) {
    for (uint32_t i = 0; i < s->n; i++) {
        orc_particle* p = &s->p[i];
        memset(p, 0, sizeof(*p));
        p->index = i;
        for (int a = 0; a < 3; a++) { p->position[a] = pos[3 * i + a]; p->velocity[a] = vel ? vel[3 * i + a] : 0.f; }
        p->mass = MASS;
        p->radius = 1.0f / 64.0f;          /* particleSystem.cpp:51,864 */
        for (int a = 0; a < 3; a++) s->hpos[4 * i + a] = p->position[a];
        s->hpos[4 * i + 3] = 1.0f;
    }
}

void orc_apply_order(orc_system* s, const uint32_t* order) {
    /* slot of each original index */
    uint32_t* slot = s->keybuf;
    for (uint32_t k = 0; k < s->n; k++) slot[s->p[k].index] = k;
    for (uint32_t k = 0; k < s->n; k++) s->tmp[k] = s->p[slot[order[k]]];
    memcpy(s->p, s->tmp, (size_t)s->n * sizeof(orc_particle));
}

/* ---- grid phases --------------------------------------------------------------*/
/* zMapZindex, particleSystem.cpp:544-549 */
void orc_map_zindex(orc_system* s) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)s->n; i++) s->p[i].zindex = orc_cell_of(s, s->p[i].position);
}

/* zSortParticles, particleSystem.cpp:551-554.  The reference uses std::sort
 * (unstable); here: stable LSD radix sort on zindex, 4 x 8 bits. */
void orc_sort(orc_system* s) {
    orc_particle* src = s->p;
    orc_particle* dst = s->tmp;
    for (int pass = 0; pass < 4; pass++) {
        size_t count[257];
        memset(count, 0, sizeof(count));
        int shift = pass * 8;
        for (uint32_t i = 0; i < s->n; i++) count[((src[i].zindex >> shift) & 0xFF) + 1]++;
        if (count[1] == s->n) continue;   /* all digits zero: pass is the identity */
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        for (uint32_t i = 0; i < s->n; i++) dst[count[(src[i].zindex >> shift) & 0xFF]++] = src[i];
        orc_particle* t = src; src = dst; dst = t;
    }
    if (src != s->p) {
        memcpy(s->p, src, (size_t)s->n * sizeof(orc_particle));
    }
}

/* zConstructBGrid, particleSystem.cpp:556-578 (memset, then run lengths) */
void orc_construct_bgrid(orc_system* s) {
    memset(s->B, 0, (size_t)s->b_size * sizeof(orc_grid_item));
    long long grid_dex = -1;
    s->n_occ = 0;
    for (uint32_t i = 0; i < s->n; i++) {
        unsigned long long zind = s->p[i].zindex;
        if ((long long)zind != grid_dex) {
            grid_dex = (long long)zind;
            s->B[grid_dex].start = i;
            s->B[grid_dex].nParticles = 1;
            s->occ[s->n_occ++] = (uint32_t)zind;
        } else {
            s->B[grid_dex].nParticles++;
        }
    }
}

/* zConstructGridArray, particleSystem.cpp:580-596: one entry per chunk of at
 * most GRID_COMPACT_WIDTH particles of every occupied cell, in cell order. */
void orc_construct_grid_array(orc_system* s) {
    free(s->Bprime);
    size_t cap = (size_t)s->n + 1, m = 0;
    s->Bprime = (orc_grid_item*)malloc(cap * sizeof(orc_grid_item));
    /* the reference walks all cells ascending; occupied cells appear in the
     * sorted particle array in ascending id order, so walk that list */
    for (uint32_t k = 0; k < s->n_occ; k++) {
        uint32_t c = s->occ[k];
        uint32_t iter = 0;
        while (iter < s->B[c].nParticles) {
            orc_grid_item gi;
            gi.start = iter + s->B[c].start;
            uint32_t left = s->B[c].nParticles - iter;
            gi.nParticles = GRID_COMPACT_WIDTH < left ? GRID_COMPACT_WIDTH : left;
            s->Bprime[m++] = gi;
            iter += GRID_COMPACT_WIDTH;
        }
    }
    s->bprime_size = (uint32_t)m;
}

/* getNeighbors, particleSystem.cpp:273-291: in-range cells of the 3x3x3
 * stencil in dx (outer), dy, dz (inner) order */
static int neighbours_of(const orc_system* s, uint32_t cell, uint32_t out[27]) {
    uint32_t c[3];
    cell_coord(s, cell, c);
    int m = 0;
    for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dz = -1; dz <= 1; dz++) {
                int x = (int)c[0] + dx, y = (int)c[1] + dy, z = (int)c[2] + dz;
                if (0 <= x && x < (int)s->grid[0] && 0 <= y && y < (int)s->grid[1] && 0 <= z && z < (int)s->grid[2])
                    out[m++] = cell_id(s, (uint32_t)x, (uint32_t)y, (uint32_t)z);
            }
    return m;
}

/* ---- pair physics ---------------------------------------------------------------*/
/* computePressureIdeal, particleSystem.cpp:226-228 (== .cu:15-17) */
static inline void pressure_ideal(orc_particle* p) {
    float v = GAS_CONSTANT * (p->density - REST_DENS);
    p->pressure = 0.f < v ? v : 0.f;    /* std::max(0.f, v) */
}

/* computeDensity, particleSystem.cpp:241-248 (== .cu:28-37) */
static inline void pair_density(orc_particle* pi, const orc_particle* pj) {
    const float POLY6 = 315.f / (65.f * PI_F * powf(m_H, 9.f));
    float rij[3] = { pi->position[0] - pj->position[0], pi->position[1] - pj->position[1], pi->position[2] - pj->position[2] };
    float r2 = sqn3(rij);
    if (r2 < HSQ) pi->density += pj->mass * POLY6 * powf(HSQ - r2, 3.f);
}

/* computeForce, particleSystem.cpp:250-259 (== .cu:39-50) */
static inline void pair_force(orc_particle* pi, const orc_particle* pj) {
    const float SPIKY_GRAD = -45.f / (PI_F * powf(m_H, 6.f));
    const float VISC_LAP = 45.f / (PI_F * powf(m_H, 6.f));
    float rij[3] = { pi->position[0] - pj->position[0], pi->position[1] - pj->position[1], pi->position[2] - pj->position[2] };
    float r = sqrtf(sqn3(rij));
    if (r < m_H) {
        /* rij.normalized(): zero vector stays zero (Eigen Dot.h:124-134) */
        float z = sqn3(rij), nrm[3];
        if (z > 0.f) { float sz = sqrtf(z); for (int a = 0; a < 3; a++) nrm[a] = rij[a] / sz; }
        else { for (int a = 0; a < 3; a++) nrm[a] = rij[a]; }
        float psum = pi->pressure + pj->pressure;
        float den = 2.f * pj->density;
        float pw = powf(m_H - r, 2.f);
        float vm = VISC * pj->mass;
        float hr = m_H - r;
        for (int a = 0; a < 3; a++) {
            /* ((((-n * m) * psum) / den) * SPIKY_GRAD) * pw, left to right */
            float t = -nrm[a];
            t = t * pj->mass; t = t * psum; t = t / den; t = t * SPIKY_GRAD; t = t * pw;
            pi->force_press[a] += t;
            /* (((vm * (vj - vi)) / rho_j) * VISC_LAP) * (h - r) */
            float u = pj->velocity[a] - pi->velocity[a];
            u = vm * u; u = u / pj->density; u = u * VISC_LAP; u = u * hr;
            pi->force_visc[a] += u;
        }
    }
}

/* computeCollision, particleSystem.cpp:261-271 (== .cu:52-65); the caller skips
 * pi.index == pj.index (:382) */
static inline void pair_collision(orc_particle* pi, const orc_particle* pj) {
    float vij[3], rij[3];
    for (int a = 0; a < 3; a++) { vij[a] = pi->velocity[a] - pj->velocity[a]; rij[a] = pi->position[a] - pj->position[a]; }
    float dij = sqrtf(sqn3(rij));
    /* COLLISION_PARAM is a double literal: the comparison is done in double */
    if ((double)dij <= COLLISION_PARAM * 2 * pi->radius && dot3(rij, vij) < 0) {
        float s = (pj->mass * (1.f + RESTITUTION)) * (dot3(rij, vij) / (dij * dij));
        for (int a = 0; a < 3; a++) pi->delta_velocity[a] += s * rij[a];
        pi->collision_count++;
    }
}

/* zcomputeDensities, particleSystem.cpp:303-322 */
void orc_compute_densities(orc_system* s) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t k = 0; k < (int64_t)s->n_occ; k++) {
        uint32_t block = s->occ[k], nb[27];
        int m = neighbours_of(s, block, nb);
        for (uint32_t i = s->B[block].start; i < s->B[block].start + s->B[block].nParticles; i++) {
            orc_particle* pi = &s->p[i];
            pi->density = 0.f;
            for (int q = 0; q < m; q++) {
                const orc_grid_item g = s->B[nb[q]];
                for (uint32_t j = g.start; j < g.start + g.nParticles; j++) pair_density(pi, &s->p[j]);
            }
            pressure_ideal(pi);
        }
    }
}

/* zcomputeForces, particleSystem.cpp:334-352 */
void orc_compute_forces(orc_system* s) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t k = 0; k < (int64_t)s->n_occ; k++) {
        uint32_t block = s->occ[k], nb[27];
        int m = neighbours_of(s, block, nb);
        for (uint32_t i = s->B[block].start; i < s->B[block].start + s->B[block].nParticles; i++) {
            orc_particle* pi = &s->p[i];
            for (int a = 0; a < 3; a++) { pi->force_press[a] = 0.f; pi->force_visc[a] = 0.f; }
            for (int q = 0; q < m; q++) {
                const orc_grid_item g = s->B[nb[q]];
                for (uint32_t j = g.start; j < g.start + g.nParticles; j++) pair_force(pi, &s->p[j]);
            }
        }
    }
}

/* zparticleCollisions, particleSystem.cpp:368-389 */
void orc_particle_collisions(orc_system* s) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t k = 0; k < (int64_t)s->n_occ; k++) {
        uint32_t block = s->occ[k], nb[27];
        int m = neighbours_of(s, block, nb);
        for (uint32_t i = s->B[block].start; i < s->B[block].start + s->B[block].nParticles; i++) {
            orc_particle* pi = &s->p[i];
            for (int a = 0; a < 3; a++) pi->delta_velocity[a] = 0.f;
            pi->collision_count = 0;
            for (int q = 0; q < m; q++) {
                const orc_grid_item g = s->B[nb[q]];
                for (uint32_t j = g.start; j < g.start + g.nParticles; j++) {
                    const orc_particle* pj = &s->p[j];
                    if (pi->index == pj->index) continue;
                    pair_collision(pi, pj);
                }
            }
            float den = pi->mass * (float)(1 + pi->collision_count);
            for (int a = 0; a < 3; a++) pi->delta_velocity[a] = -pi->delta_velocity[a] / den;
        }
    }
}

/* SEQUENTIAL computeDensities, particleSystem.cpp:293-301 */
void orc_compute_densities_n2(orc_system* s) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)s->n; i++) {
        orc_particle* pi = &s->p[i];
        pi->density = 0.f;
        for (uint32_t j = 0; j < s->n; j++) pair_density(pi, &s->p[j]);
        pressure_ideal(pi);
    }
}

/* zintegrate, particleSystem.cpp:437-482 (== kernelIntegrate, .cu:375-420) */
void orc_integrate(orc_system* s, float dt) {
#pragma omp parallel for schedule(static, 64)
    for (int64_t i = 0; i < (int64_t)s->n; i++) {
        orc_particle* p = &s->p[i];
        const float g = GRAVITY * G_MODIFIER;
        float force_grav[3] = { 0.f, g * p->density, 0.f };
        for (int a = 0; a < 3; a++) {
            float force = (p->force_press[a] + p->force_visc[a]) + force_grav[a];
            float accel = force / p->density;
            p->velocity[a] += dt * accel + p->delta_velocity[a];
        }
        for (int a = 0; a < 3; a++) p->position[a] += dt * p->velocity[a];
        for (int a = 0; a < 3; a++) {   /* X, Y, Z; lower wall first, then upper */
            if (p->position[a] - EPS_F < s->box_min[a]) {
                p->position[a] = s->box_min[a] + EPS_F;
                p->velocity[a] *= -.75f;
            }
            if (p->position[a] + EPS_F > s->box_max[a]) {
                p->position[a] = s->box_max[a] - EPS_F;
                p->velocity[a] *= -.75f;
            }
        }
        float* o = &s->hpos[(size_t)p->index * 4];
        o[0] = p->position[0]; o[1] = p->position[1]; o[2] = p->position[2]; o[3] = 1.0f;
    }
}

/* OMP branch of ParticleSystem::update, particleSystem.cpp:743-767 */
void orc_step(orc_system* s, float dt) {
    orc_map_zindex(s);
    orc_sort(s);
    orc_construct_bgrid(s);
    orc_construct_grid_array(s);
    orc_compute_densities(s);
    orc_compute_forces(s);
    orc_particle_collisions(s);
    orc_integrate(s, dt);
}

void orc_get_state(const orc_system* s, float* pos, float* vel, float* density, float* pressure) {
    for (uint32_t k = 0; k < s->n; k++) {
        const orc_particle* p = &s->p[k];
        size_t i = p->index;
        for (int a = 0; a < 3; a++) { if (pos) pos[3 * i + a] = p->position[a]; if (vel) vel[3 * i + a] = p->velocity[a]; }
        if (density) density[i] = p->density;
        if (pressure) pressure[i] = p->pressure;
    }
}

void orc_get_forces(const orc_system* s, float* fp, float* fv, float* dv, int32_t* count) {
    for (uint32_t k = 0; k < s->n; k++) {
        const orc_particle* p = &s->p[k];
        size_t i = p->index;
        for (int a = 0; a < 3; a++) {
            if (fp) fp[3 * i + a] = p->force_press[a];
            if (fv) fv[3 * i + a] = p->force_visc[a];
            if (dv) dv[3 * i + a] = p->delta_velocity[a];
        }
        if (count) count[i] = p->collision_count;
    }
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
