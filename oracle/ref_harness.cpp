// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Drives the reference's own CPU (OpenMP-mode) SPH step so that the C
// restatement in oracle/sph_oracle.c can be pinned against the real thing.
//
// The reference translation unit /root/reference/SPH/particleSystem.cpp is
// compiled WHERE IT LIES (see oracle/Makefile; no reference source is copied
// into this repository and no stand-in headers are written: the CUDA toolkit
// headers it includes ship in this image inside the triton wheel, GL/glx.h is
// in /usr/include, Eigen and the helper_* headers are vendored by the
// reference itself).  This file only adds a subclass that
//   * uses the protected default constructor (particleSystem.h:113) so that
//     neither the GL calls nor the CUDA seam of _initialize() are touched,
//   * fills the protected members the z* methods read, and
//   * calls the z* methods in the order of the OMP branch of
//     ParticleSystem::update() (particleSystem.cpp:740-768).
// The 15 extern "C" CUDA seam symbols (particleSystem.cuh:3-30) stay
// unresolved in the binary (lazy binding, never called).
//
// File protocol (all little-endian, see oracle/refio.py):
//   in : u32 magic 'SPHI', u32 n, f32 box[3], u32 gridDim, f32 dt, u32 steps,
//        u32 flags, then f32 pos[n*3], f32 vel[n*3]
//   out: u32 magic 'SPHO', u32 n, u32 nrec, then nrec records; each record is
//        u32 tag, u32 step, u32 count, u32 width, then count*width 4-byte words.
//   flags bit0: dump per-phase records for every step (tags below)
//         bit1: dump state records at steps listed via argv (comma list)
// helper_gl.h WITHOUT HELPERGL_EXTERN_GL_FUNC_IMPLEMENTATION is the reference's own way of
// defining the __HelperGL::gl* function pointers that particleSystem.cpp declares extern
// (common/inc/helper_gl.h:43-49; particles.cpp / render_particles.cpp do the same): they are
// filled by glXGetProcAddress from the image's libGL and never called here.
#include <helper_gl.h>
#include "particleSystem.h"
#ifdef REF_DROPIN
#include "particleSystem.cuh"     // the reference's declarations of its seam
#endif

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <set>
#include <omp.h>

enum Tag : uint32_t {
    TAG_ZINDEX = 1,      // after zMapZindex, by array slot (pre-sort): zindex
    TAG_ORDER = 2,       // after zSortParticles: original index per sorted slot
    TAG_SORTED_Z = 3,    // after zSortParticles: zindex per sorted slot
    TAG_BCELLS = 4,      // after zConstructBGrid: occupied cells {cell, nParticles, start}
    TAG_BPRIME = 5,      // after zConstructGridArray: {start, nParticles}
    TAG_DENS = 6,        // after zcomputeDensities, by ORIGINAL index: {density, pressure}
    TAG_FORCE = 7,       // after zcomputeForces, by original index: {fpress xyz, fvisc xyz}
    TAG_COLL = 8,        // after zparticleCollisions, by original index: {dv xyz, count(as float)}
    TAG_STATE = 9,       // after zintegrate, by original index: {pos xyz, vel xyz, density, pressure}
    TAG_HPOS = 10,       // m_hPos after zintegrate: {x,y,z,w} by original index
};

class RefHarness : public ParticleSystem {
public:
    RefHarness(uint32_t n, float3 box, uint32_t grid) : ParticleSystem() {
        m_bInitialized = false;          // never runs _finalize()
        m_numParticles = n;
        m_boxDims = box;
        m_compute_mode = OMP_PARALLEL;
        m_solverIterations = 1;
        // the ctor hard-wires 32 via BOX_SIZE (particleSystem.cpp:46); every
        // z* method reads these members instead, so other grids are reached
        // by setting them here (same formula with the box edge).
        m_h_B_dim = grid;
        m_h_B_size = grid * grid * grid;
        m_h_B = new Grid_item[m_h_B_size];
        m_h_B_prime = nullptr;
        m_h_B_prime_size = 0;
        m_params.particleRadius = 1.0f / 64.0f;          // particleSystem.cpp:51
        m_params.colliderPos = make_float3(-1.2f, -0.8f, 0.8f);
        m_params.gravity = make_float3(0.f, 0.f, 0.f);
        m_params.colliderRadius = 0.2f;
        m_params.boxMin.x = -box.x / 2;                  // particleSystem.cpp:55-60
        m_params.boxMin.y = -box.y / 2;
        m_params.boxMin.z = -box.z / 2;
        m_params.boxMax.x = box.x / 2;
        m_params.boxMax.y = box.y / 2;
        m_params.boxMax.z = box.z / 2;
        m_params.boxDims = box;
        m_params.gridDim = grid;
        m_hPos = new float[(size_t)n * 4];
        memset(m_hPos, 0, sizeof(float) * 4 * n);
        m_particles.resize(n);
    }

    // same fields initGrid() sets (particleSystem.cpp:854-864) plus the ones it
    // leaves to the first step.
    void load(const float* pos, const float* vel) {
        for (uint32_t i = 0; i < m_numParticles; i++) {
            Particle& p = m_particles[i];
            p.index = i;
            p.position = { pos[3 * i], pos[3 * i + 1], pos[3 * i + 2] };
            p.velocity = { vel[3 * i], vel[3 * i + 1], vel[3 * i + 2] };
            p.delta_velocity = { 0.f, 0.f, 0.f };
            p.force_press = { 0.f, 0.f, 0.f };
            p.force_visc = { 0.f, 0.f, 0.f };
            p.mass = MASS;
            p.density = 0.f;
            p.pressure = 0.f;
            p.radius = m_params.particleRadius;
            p.collision_count = 0;
            p.zindex = 0;
        }
    }

    struct Rec { uint32_t tag, step, count, width; std::vector<uint32_t> w; };
    std::vector<Rec> recs;

    static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

    void rec_by_slot_u32(uint32_t tag, uint32_t step, int which) {
        Rec r{ tag, step, m_numParticles, 1, {} };
        r.w.resize(m_numParticles);
        for (uint32_t i = 0; i < m_numParticles; i++)
            r.w[i] = which == 0 ? m_particles[i].zindex : m_particles[i].index;
        recs.push_back(std::move(r));
    }

    void rec_cells(uint32_t step) {
        Rec r{ TAG_BCELLS, step, 0, 3, {} };
        for (uint32_t c = 0; c < m_h_B_size; c++)
            if (m_h_B[c].nParticles) {
                r.w.push_back(c); r.w.push_back(m_h_B[c].nParticles); r.w.push_back(m_h_B[c].start);
                r.count++;
            }
        recs.push_back(std::move(r));
    }

    void rec_bprime(uint32_t step) {
        Rec r{ TAG_BPRIME, step, m_h_B_prime_size, 2, {} };
        for (uint32_t c = 0; c < m_h_B_prime_size; c++) {
            r.w.push_back(m_h_B_prime[c].start); r.w.push_back(m_h_B_prime[c].nParticles);
        }
        recs.push_back(std::move(r));
    }

    void rec_particles(uint32_t tag, uint32_t step) {
        uint32_t width = tag == TAG_DENS ? 2 : tag == TAG_FORCE ? 6 : tag == TAG_COLL ? 4 : 8;
        Rec r{ tag, step, m_numParticles, width, {} };
        r.w.assign((size_t)m_numParticles * width, 0);
        for (uint32_t s = 0; s < m_numParticles; s++) {
            const Particle& p = m_particles[s];
            uint32_t* o = &r.w[(size_t)p.index * width];
            if (tag == TAG_DENS) { o[0] = f2u(p.density); o[1] = f2u(p.pressure); }
            else if (tag == TAG_FORCE) {
                for (int k = 0; k < 3; k++) { o[k] = f2u(p.force_press[k]); o[3 + k] = f2u(p.force_visc[k]); }
            } else if (tag == TAG_COLL) {
                for (int k = 0; k < 3; k++) o[k] = f2u(p.delta_velocity[k]);
                o[3] = f2u((float)p.collision_count);
            } else {
                for (int k = 0; k < 3; k++) { o[k] = f2u(p.position[k]); o[3 + k] = f2u(p.velocity[k]); }
                o[6] = f2u(p.density); o[7] = f2u(p.pressure);
            }
        }
        recs.push_back(std::move(r));
    }

    void rec_hpos(uint32_t step) {
        Rec r{ TAG_HPOS, step, m_numParticles, 4, {} };
        r.w.resize((size_t)m_numParticles * 4);
        memcpy(r.w.data(), m_hPos, sizeof(float) * 4 * m_numParticles);
        recs.push_back(std::move(r));
    }

#ifdef REF_DROPIN
    // ---- drop-in mode: the reference's OWN update() in CUDA_PARALLEL mode (particleSystem.cpp:769-801),
    // with its extern "C" seam (particleSystem.cuh:3-30) served by libsph_hip.so on an MI355X.
    // What _initialize() does at particleSystem.cpp:121-128, minus the GL calls.
    void dropin_init() {
        m_compute_mode = CUDA_PARALLEL;
        allocateArray((void**)&m_d_params, sizeof(SimParams));
        allocateArray((void**)&m_d_particles, sizeof(Particle) * m_numParticles);
        allocateArray((void**)&m_d_B, sizeof(Grid_item) * m_h_B_size);
        allocateArray((void**)&m_d_B_prime, sizeof(Grid_item) * m_numParticles);
        m_posVbo = 1;
        registerGLBufferObject(m_posVbo, &m_cuda_posvbo_resource);
        copyArrayToDevice((void*)m_d_particles, m_particles.data(), m_numParticles * sizeof(Particle));   // :920
        m_bInitialized = true;           // update() asserts it; the destructor never runs (_Exit)
    }
    void dropin_step(float dt, uint32_t stepno, bool state, bool phases) {
        update(dt, 0.f);                 // the reference's code, CUDA branch
        if (state || phases) {
            threadSync();
            copyArrayFromDevice(m_particles.data(), m_d_particles, m_numParticles * sizeof(Particle));
            rec_particles(TAG_STATE, stepno);
        }
        if (phases) {
            // what the seam left in the caller's B / B' arrays and in Particle::zindex (particleSystem.cu:503-528):
            // read back through the seam's own copy call into the members rec_cells / rec_bprime print
            rec_by_slot_u32(TAG_ORDER, stepno, 1);
            rec_by_slot_u32(TAG_SORTED_Z, stepno, 0);
            copyArrayFromDevice(m_h_B, m_d_B, m_h_B_size * sizeof(Grid_item));
            rec_cells(stepno);
            m_h_B_prime = new Grid_item[m_h_B_prime_size ? m_h_B_prime_size : 1];
            copyArrayFromDevice(m_h_B_prime, m_d_B_prime, m_h_B_prime_size * sizeof(Grid_item));
            rec_bprime(stepno);
            delete[] m_h_B_prime;
            m_h_B_prime = nullptr;
        }
    }
#endif

    // one time step: the OMP branch of update(), particleSystem.cpp:743-767
    void step(float dt, uint32_t stepno, bool phases, bool state, double* phase_s) {
        double t0 = omp_get_wtime();
        zMapZindex();
        double t1 = omp_get_wtime();
        if (phases) rec_by_slot_u32(TAG_ZINDEX, stepno, 0);
        zSortParticles();
        double t2 = omp_get_wtime();
        if (phases) { rec_by_slot_u32(TAG_ORDER, stepno, 1); rec_by_slot_u32(TAG_SORTED_Z, stepno, 0); }
        zConstructBGrid();
        double t3 = omp_get_wtime();
        if (phases) rec_cells(stepno);
        zConstructGridArray();
        double t4 = omp_get_wtime();
        if (phases) rec_bprime(stepno);
        zcomputeDensities();
        double t5 = omp_get_wtime();
        if (phases) rec_particles(TAG_DENS, stepno);
        zcomputeForces();
        double t6 = omp_get_wtime();
        if (phases) rec_particles(TAG_FORCE, stepno);
        zparticleCollisions();
        double t7 = omp_get_wtime();
        if (phases) rec_particles(TAG_COLL, stepno);
        zintegrate(dt);
        double t8 = omp_get_wtime();
        if (phases) rec_hpos(stepno);
        if (phases || state) rec_particles(TAG_STATE, stepno);
        delete[] m_h_B_prime;            // particleSystem.cpp:767
        m_h_B_prime = nullptr;
        if (phase_s) {
            phase_s[0] += t1 - t0; phase_s[1] += t2 - t1; phase_s[2] += t3 - t2; phase_s[3] += t4 - t3;
            phase_s[4] += t5 - t4; phase_s[5] += t6 - t5; phase_s[6] += t7 - t6; phase_s[7] += t8 - t7;
        }
    }
};

static void die(const char* m) { fprintf(stderr, "sph_ref: %s\n", m); exit(2); }

int main(int argc, char** argv) {
    if (argc < 3) die("usage: sph_ref <in.bin> <out.bin> [dump_steps=comma,list] [threads=N]");
    std::set<uint32_t> dump_steps;
    int threads = 0;
    for (int a = 3; a < argc; a++) {
        std::string s = argv[a];
        if (s.rfind("dump_steps=", 0) == 0) {
            const char* p = s.c_str() + 11;
            while (*p) { dump_steps.insert((uint32_t)strtoul(p, (char**)&p, 10)); if (*p == ',') p++; }
        } else if (s.rfind("threads=", 0) == 0) threads = atoi(s.c_str() + 8);
    }
    if (threads > 0) omp_set_num_threads(threads);

    FILE* f = fopen(argv[1], "rb");
    if (!f) die("cannot open input");
    uint32_t magic, n, grid, steps, flags; float box[3], dt;
    if (fread(&magic, 4, 1, f) != 1 || magic != 0x49485053u) die("bad magic");
    if (fread(&n, 4, 1, f) != 1 || fread(box, 4, 3, f) != 3 || fread(&grid, 4, 1, f) != 1 ||
        fread(&dt, 4, 1, f) != 1 || fread(&steps, 4, 1, f) != 1 || fread(&flags, 4, 1, f) != 1) die("short header");
    std::vector<float> pos((size_t)n * 3), vel((size_t)n * 3);
    if (fread(pos.data(), 4, pos.size(), f) != pos.size() || fread(vel.data(), 4, vel.size(), f) != vel.size()) die("short body");
    fclose(f);

    RefHarness* h = new RefHarness(n, make_float3(box[0], box[1], box[2]), grid);  // never deleted on purpose
    h->load(pos.data(), vel.data());
    double phase_s[8] = { 0 };
#ifdef REF_DROPIN
    h->dropin_init();
    threadSync();
#endif
    double t0 = omp_get_wtime();
    for (uint32_t s = 1; s <= steps; s++) {
#ifdef REF_DROPIN
        h->dropin_step(dt, s, dump_steps.count(s) != 0 || s == steps, (flags & 1) != 0);
#else
        h->step(dt, s, (flags & 1) != 0, dump_steps.count(s) != 0 || s == steps, phase_s);
#endif
    }
#ifdef REF_DROPIN
    threadSync();
#endif
    double t1 = omp_get_wtime();

    FILE* o = fopen(argv[2], "wb");
    if (!o) die("cannot open output");
    uint32_t omagic = 0x4F485053u, nrec = (uint32_t)h->recs.size();
    fwrite(&omagic, 4, 1, o); fwrite(&n, 4, 1, o); fwrite(&nrec, 4, 1, o);
    for (auto& r : h->recs) {
        fwrite(&r.tag, 4, 1, o); fwrite(&r.step, 4, 1, o); fwrite(&r.count, 4, 1, o); fwrite(&r.width, 4, 1, o);
        fwrite(r.w.data(), 4, r.w.size(), o);
    }
    fclose(o);
    int nt = 1;
#pragma omp parallel
    {
#pragma omp master
        nt = omp_get_num_threads();
    }
    printf("{\"n\": %u, \"steps\": %u, \"threads\": %d, \"seconds\": %.6f, \"particle_steps_per_s\": %.1f, "
           "\"phase_s\": {\"z-index\": %.6f, \"sort\": %.6f, \"b-grid\": %.6f, \"b'-grid\": %.6f, \"dens\": %.6f, "
           "\"force\": %.6f, \"collision\": %.6f, \"integrate\": %.6f}}\n",
           n, steps, nt, t1 - t0, steps ? (double)n * steps / (t1 - t0) : 0.0,
           phase_s[0], phase_s[1], phase_s[2], phase_s[3], phase_s[4], phase_s[5], phase_s[6], phase_s[7]);
    fflush(stdout);
    _Exit(0);   // skip static destructors / ~ParticleSystem (would call the GL + CUDA seam)
}
