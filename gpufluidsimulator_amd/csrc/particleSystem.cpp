// particleSystem.cpp -- host C++ class of include/particleSystem.h on top of the C ABI
// (include/sph_hip.h).  Plain C++ (no HIP headers): everything on the device goes through
// libsph_hip's entry points.  Mirrors the public behaviour of the reference's
// SPH/particleSystem.cpp; citations give the reference lines each method stands for.
// Compile with -ffp-contract=off: the initial conditions must match gpufluidsimulator_amd/ic.py
// bit for bit.
#include "../../include/particleSystem.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

const float kH = 0.1f;                       // m_H, particles_kernel.cuh:22
const float kRadius = 1.0f / 64.0f;          // particleSystem.cpp:51
const uint32_t kSeed = 1973;                 // srand(1973), particleSystem.cpp:841

[[noreturn]] void die(const char* what) {
    // checkCudaErrors -> fprintf + exit(EXIT_FAILURE), common/inc/helper_cuda.h:566-579
    fprintf(stderr, "ParticleSystem: %s: %s\n", what, sph_last_error());
    exit(EXIT_FAILURE);
}
#define SPH_CHECK(call) do { if ((call) < 0) die(#call); } while (0)

uint32_t hash_u32(uint32_t x) {              // lowbias32, twin of ic._hash_u32
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

float uniform01(uint32_t counter, uint32_t stream, uint32_t seed) {   // twin of ic.uniform01
    uint32_t k = hash_u32(counter * 3u + stream + seed * 0x9E3779B9u);
    k = hash_u32(k ^ 0x85EBCA6Bu);
    return (float)(k >> 8) * (1.0f / 16777216.0f);
}

uint next_pow2(uint x) {                     // particleSystem.h:34-43
    x--; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; x++;
    return x;
}

}  // namespace

extern "C" void sph_ic_dam_break(const uint32_t lattice[3], const float box[3], int jitter, uint64_t start,
                                 uint64_t count, float* pos, float* vel) {
    // pos = spacing*i + particleRadius + boxMin + (w*u - w/2)*jitter, particleSystem.cpp:855-857
    const float spacing = 2.0f * kRadius;
    const float jit = kRadius * 0.01f;       // particleSystem.cpp:911
    const uint64_t nx = lattice[0], ny = lattice[1];
    for (uint64_t k = 0; k < count; k++) {
        const uint64_t idx = start + k;
        const float ia[3] = {(float)(idx % nx), (float)((idx / nx) % ny), (float)(idx / (nx * ny))};
        for (int a = 0; a < 3; a++) {
            const float w = box[a];
            const float bmin = -w / 2.0f;
            float base = (spacing * ia[a] + kRadius) + bmin;
            if (jitter) {
                const float u = uniform01((uint32_t)idx, (uint32_t)a, kSeed);
                base = base + (w * u - w / 2.0f) * jit;
            }
            pos[3 * k + a] = base;
            if (vel) vel[3 * k + a] = 0.f;
        }
    }
}

extern "C" void sph_ic_random_box(uint64_t n, const float box[3], float speed, uint32_t seed, float fill, float* pos,
                                  float* vel) {
    // reset(CONFIG_RANDOM), particleSystem.cpp:880-905: w*frand() - w/2
    const float two_speed = (float)(2.0 * (double)speed);
    for (uint64_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float w = box[a];
            const float u = uniform01((uint32_t)i, (uint32_t)a, seed);
            pos[3 * i + a] = w * (u * fill) - w / 2.0f;
            if (vel) {
                float v = 0.f;
                if (speed != 0.f) v = (uniform01((uint32_t)i, (uint32_t)(3 + a), seed) - 0.5f) * two_speed;
                vel[3 * i + a] = v;
            }
        }
}

// ---- lifetime (particleSystem.cpp:38-71, 108-190) ----------------------------------------------------
ParticleSystem::ParticleSystem(uint numParticles, float3 boxDims, ParticleComputeMode mode)
    : ParticleSystem(numParticles, boxDims, mode, uint3{0u, 0u, 0u}) {}

ParticleSystem::ParticleSystem(uint numParticles, float3 boxDims, ParticleComputeMode mode, uint3 gridDims)
    : m_bInitialized(false), m_numParticles(numParticles), m_boxDims(boxDims), m_solverIterations(1),
      m_compute_mode(mode), m_ctx(nullptr), m_hostStale(false), m_log(nullptr), m_logLastMs(0), m_logGlobalMs(0), m_logFreqMs(2000.0), m_logStyle(0), m_logFrames(0) {
    if (mode != CUDA_PARALLEL) {
        fprintf(stderr, "ParticleSystem: only the GPU compute mode exists in this build "
                        "(SEQUENTIAL / OMP_PARALLEL are the reference's CPU paths; there is no CPU fallback)\n");
        exit(EXIT_FAILURE);
    }
    m_grid.x = gridDims.x ? gridDims.x : next_pow2((uint)(boxDims.x / (0.66666f * kH)));   // particleSystem.cpp:46, per axis
    m_grid.y = gridDims.y ? gridDims.y : next_pow2((uint)(boxDims.y / (0.66666f * kH)));
    m_grid.z = gridDims.z ? gridDims.z : next_pow2((uint)(boxDims.z / (0.66666f * kH)));
    m_params.particleRadius = kRadius;                            // particleSystem.cpp:51-62
    m_params.colliderPos = make_float3(-1.2f, -0.8f, 0.8f);
    m_params.gravity = make_float3(0.f, 0.f, 0.f);
    m_params.colliderRadius = 0.2f;
    m_params.boxMin = make_float3(-boxDims.x / 2, -boxDims.y / 2, -boxDims.z / 2);
    m_params.boxMax = make_float3(boxDims.x / 2, boxDims.y / 2, boxDims.z / 2);
    m_params.boxDims = boxDims;
    m_params.gridDim = m_grid.x;
    _initialize((int)numParticles);
}

ParticleSystem::~ParticleSystem() {
    _finalize();
    m_numParticles = 0;
}

void ParticleSystem::_initialize(int numParticles) {
    m_numParticles = (uint)numParticles;
    m_hPos.assign((size_t)m_numParticles * 4, 0.f);
    m_hVel.assign((size_t)m_numParticles * 4, 0.f);
    sph_params p;
    const float box[3] = {m_boxDims.x, m_boxDims.y, m_boxDims.z};
    const uint32_t grid[3] = {m_grid.x, m_grid.y, m_grid.z};
    sph_default_params(&p, box, grid);
    SPH_CHECK(sph_create(&m_ctx, -1, m_numParticles ? m_numParticles : 1, &p));   // -1: sph_select_device's choice
    m_bInitialized = true;
}

void ParticleSystem::_finalize() {
    if (!m_bInitialized) return;
    if (m_log) fclose((FILE*)m_log);
    sph_destroy(m_ctx);
    m_ctx = nullptr;
    m_bInitialized = false;
}

// ---- host <-> device mirrors -----------------------------------------------------------------------------
void ParticleSystem::uploadAll() {
    const size_t n = m_numParticles;
    m_xyz.resize(n * 3); m_vxyz.resize(n * 3);
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { m_xyz[3 * i + a] = m_hPos[4 * i + a]; m_vxyz[3 * i + a] = m_hVel[4 * i + a]; }
    SPH_CHECK(sph_upload(m_ctx, (uint32_t)n, m_xyz.data(), m_vxyz.data(), nullptr));
    m_hostStale = false;
}

void ParticleSystem::downloadAll() {
    if (!m_hostStale) return;
    const size_t n = m_numParticles;
    m_xyz.resize(n * 3); m_vxyz.resize(n * 3); m_hDens.resize(n);
    SPH_CHECK(sph_download(m_ctx, 0, (uint32_t)n, m_xyz.data(), m_vxyz.data(), m_hDens.data(), nullptr));
    for (size_t i = 0; i < n; i++) {
        for (int a = 0; a < 3; a++) { m_hPos[4 * i + a] = m_xyz[3 * i + a]; m_hVel[4 * i + a] = m_vxyz[3 * i + a]; }
        m_hPos[4 * i + 3] = 1.0f;
        m_hVel[4 * i + 3] = 0.0f;
    }
    m_hostStale = false;
}

// ---- stepping (particleSystem.cpp:719-817) ------------------------------------------------------------------
void ParticleSystem::update(float deltaTime, float fps) {
    if (!m_bInitialized) { fprintf(stderr, "ParticleSystem::update before initialisation\n"); exit(EXIT_FAILURE); }
    if (m_solverIterations > 0) SPH_CHECK(sph_step(m_ctx, deltaTime, (uint32_t)m_solverIterations));
    m_hostStale = true;
    if (m_log) {
        // dumpBenchmark, particleSystem.cpp:697-716: at most one line every BENCHMARK_FREQ = 2000 ms
        using clk = std::chrono::steady_clock;
        const double now = std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
        if (m_logLastMs == 0) m_logLastMs = now;
        m_logFrames++;
        if (now - m_logLastMs > m_logFreqMs) {
            // LOG_OSCAR's FPS field is a RATE (the committed logs: updates per second as the window title showed them); the
            // current source passes a frame COUNT in `fps` and prints that (LOG_FRAMES)
            const double rate = now > m_logLastMs ? 1e3 * (double)m_logFrames / (now - m_logLastMs) : 0.0;
            m_logFrames = 0;
            float ms[SPH_PH_COUNT]; uint32_t steps = 0;
            SPH_CHECK(sph_timing_get(m_ctx, ms, &steps));
            SPH_CHECK(sph_timing_reset(m_ctx));
            m_logGlobalMs += now - m_logLastMs;
            m_logLastMs = now;
            const double k = steps ? 1e6 / steps : 0.0;     // ms sums -> ns per step
            double total = 0;
            for (int i = 0; i < SPH_PH_COUNT; i++) total += ms[i];
            const long long t_ns = (long long)(total * k), z_ns = (long long)(ms[SPH_PH_ZINDEX] * k), s_ns = (long long)(ms[SPH_PH_SORT] * k),
                            b_ns = (long long)(ms[SPH_PH_BGRID] * k), d_ns = (long long)(ms[SPH_PH_DENS] * k),
                            f_ns = (long long)(ms[SPH_PH_FORCE] * k), c_ns = (long long)(ms[SPH_PH_COLLISION] * k),
                            i_ns = (long long)(ms[SPH_PH_INTEGRATE] * k);
            const char* body = "\ttotal:%lldns,\t\tcopying:%lldns,\t\tz-index:%lldns,\t\tsort:%lldns,\t\tb-grid:%lldns,\t\t"
                               "b'-grid:%lldns,\t\tdens:%lldns,\t\tforce:%lldns,\t\tcollision:%lldns,\t\tintegrate:%lldns,\t\t";
            if (m_logStyle == LOG_OSCAR) fprintf((FILE*)m_log, "%.3fsec", m_logGlobalMs / 1000);    // always a decimal point: benchmark.py:12
            else fprintf((FILE*)m_log, "%gsec", m_logGlobalMs / 1000);
            fprintf((FILE*)m_log, body, t_ns, 0LL, z_ns, s_ns, b_ns, 0LL, d_ns, f_ns, c_ns, i_ns);
            if (m_logStyle == LOG_OSCAR) fprintf((FILE*)m_log, "FPS:%.3ffps\n", rate);
            else fprintf((FILE*)m_log, "frames:%lldframes\n", (long long)fps);
            fflush((FILE*)m_log);
        }
    }
}

// ---- initial conditions (particleSystem.cpp:839-921) ------------------------------------------------------------
void ParticleSystem::reset(ParticleConfig config) {
    const size_t n = m_numParticles;
    m_xyz.assign(n * 3, 0.f); m_vxyz.assign(n * 3, 0.f);
    const float box[3] = {m_boxDims.x, m_boxDims.y, m_boxDims.z};
    switch (config) {
        default:
        case CONFIG_RANDOM:
            sph_ic_random_box(n, box, 0.f, kSeed, 1.0f, m_xyz.data(), m_vxyz.data());
            break;
        case CONFIG_GRID: {
            // the smallest cube lattice that holds N (the reference: ceil(powf(N, 1/3)), which
            // overshoots by one for perfect cubes under glibc)
            uint32_t s = (uint32_t)std::floor(std::cbrt((double)n));
            while ((uint64_t)s * s * s < n) s++;
            const uint32_t lattice[3] = {s, s, s};
            // generated on the device (bit-identical to the host twin sph_ic_dam_break); the host mirror is
            // refreshed lazily by getArray()/dumpParticles()
            SPH_CHECK(sph_reset_lattice(m_ctx, lattice, 1, nullptr, 0, (uint32_t)n));
            m_hostStale = true;
            return;
        }
    }
    for (size_t i = 0; i < n; i++) {
        for (int a = 0; a < 3; a++) { m_hPos[4 * i + a] = m_xyz[3 * i + a]; m_hVel[4 * i + a] = 0.f; }
        m_hPos[4 * i + 3] = 1.0f; m_hVel[4 * i + 3] = 0.f;
    }
    SPH_CHECK(sph_upload(m_ctx, (uint32_t)n, m_xyz.data(), m_vxyz.data(), nullptr));
    m_hostStale = false;
}

// particleSystem.cpp:923-961.  Overwrites the positions of particles [start, ...) with a jittered
// sphere lattice; velocities are left alone and `vel` is unused, as upstream.  Unlike upstream the
// current device state is fetched first (the reference re-uploads a stale host copy in CUDA mode).
// particleSystem.cpp:928-961.  The reference edits its host copy from `start` on and copies that range to the
// device -- after update() the host copy is stale there, so the edit rewinds every particle it copies (SURVEY
// A.2).  Here only the sphere's own particles change, on the device (sph_set_by_index).
void ParticleSystem::addSphere(int start, float* pos, float* vel, int r, float spacing) {
    (void)vel;
    uint index = (uint)start;
    const float w = m_boxDims.x, h = m_boxDims.y, d = m_boxDims.z;
    const float jitter = m_params.particleRadius * 0.01f;
    uint32_t counter = 0;
    std::vector<float> xyz;
    for (int z = -r; z <= r; z++)
        for (int y = -r; y <= r; y++)
            for (int x = -r; x <= r; x++) {
                const float dx = x * spacing, dy = y * spacing, dz = z * spacing;
                const float l = sqrtf(dx * dx + dy * dy + dz * dz);
                if ((l <= m_params.particleRadius * 2.0f * r) && (index < m_numParticles)) {
                    xyz.push_back(pos[0] + dx + (w * uniform01(counter, 0, kSeed + 1) - w / 2) * jitter);
                    xyz.push_back(pos[1] + dy + (h * uniform01(counter, 1, kSeed + 1) - h / 2) * jitter);
                    xyz.push_back(pos[2] + dz + (d * uniform01(counter, 2, kSeed + 1) - d / 2) * jitter);
                    index++;
                    counter++;
                }
            }
    std::vector<float> xyzw(xyz.size() / 3 * 4);
    for (size_t k = 0; k < xyz.size() / 3; k++) {
        xyzw[4 * k] = xyz[3 * k]; xyzw[4 * k + 1] = xyz[3 * k + 1]; xyzw[4 * k + 2] = xyz[3 * k + 2]; xyzw[4 * k + 3] = 1.f;
    }
    setArray(POSITION, xyzw.data(), start, (int)(xyz.size() / 3));
}

void ParticleSystem::dumpParticles(uint start, uint count) {   // particleSystem.cpp:819-827
    downloadAll();
    for (uint i = start; i < start + count && i < m_numParticles; i++)
        printf("pos: (%.4f, %.4f, %.4f, %.4f)\n", m_hPos[i * 4 + 0], m_hPos[i * 4 + 1], m_hPos[i * 4 + 2], m_hPos[i * 4 + 3]);
}

// ---- additive API -----------------------------------------------------------------------------------------------
float* ParticleSystem::getArray(ParticleArray array) {
    m_hostStale = true;
    downloadAll();
    return array == POSITION ? m_hPos.data() : m_hVel.data();
}

void ParticleSystem::setArray(ParticleArray array, const float* data, int start, int count) {
    if (start < 0 || count < 0 || (size_t)start + (size_t)count > m_numParticles) {
        fprintf(stderr, "ParticleSystem::setArray: range [%d, %d) outside 0..%u\n", start, start + count, m_numParticles);
        exit(EXIT_FAILURE);
    }
    if (count == 0) return;
    std::vector<float>& dst = array == POSITION ? m_hPos : m_hVel;
    if (sph_num_particles(m_ctx) != m_numParticles) {       // nothing on the device yet: edit the host copy, upload it
        memcpy(&dst[(size_t)start * 4], data, (size_t)count * 4 * sizeof(float));
        uploadAll();
        return;
    }
    // the particles are on the device: change just these there, by creation index (no round trip of the state)
    std::vector<float> xyz((size_t)count * 3);
    for (int i = 0; i < count; i++)
        for (int a = 0; a < 3; a++) xyz[3 * (size_t)i + a] = data[4 * (size_t)i + a];
    SPH_CHECK(sph_set_by_index(m_ctx, (uint32_t)start, (uint32_t)count, array == POSITION ? xyz.data() : nullptr,
                               array == VELOCITY ? xyz.data() : nullptr));
    if (!m_hostStale) memcpy(&dst[(size_t)start * 4], data, (size_t)count * 4 * sizeof(float));   // a fresh mirror stays fresh
}

void ParticleSystem::setSimParams(const SimParams& p) {
    // the per-update SimParams upload of the reference (particleSystem.cpp:723); the grid is fixed at
    // construction, everything a kernel reads (box) is forwarded.
    m_params = p;
    sph_params q;
    SPH_CHECK(sph_get_params(m_ctx, &q));
    q.box_min[0] = p.boxMin.x; q.box_min[1] = p.boxMin.y; q.box_min[2] = p.boxMin.z;
    q.box_max[0] = p.boxMax.x; q.box_max[1] = p.boxMax.y; q.box_max[2] = p.boxMax.z;
    q.particle_radius = p.particleRadius;
    SPH_CHECK(sph_set_params(m_ctx, &q));
    m_boxDims = make_float3(p.boxMax.x - p.boxMin.x, p.boxMax.y - p.boxMin.y, p.boxMax.z - p.boxMin.z);
}

const float* ParticleSystem::getDensities() {
    m_hostStale = true;
    downloadAll();
    return m_hDens.data();
}

void* ParticleSystem::getPositionsDevice() {
    void* p = nullptr;
    SPH_CHECK(sph_positions_dev(m_ctx, &p));
    return p;
}

void ParticleSystem::saveState(const std::string& path) { SPH_CHECK(sph_snapshot_save(m_ctx, path.c_str())); }

void ParticleSystem::loadState(const std::string& path) {
    SPH_CHECK(sph_snapshot_load(m_ctx, path.c_str()));
    m_numParticles = sph_num_particles(m_ctx);
    m_hPos.assign((size_t)m_numParticles * 4, 0.f);
    m_hVel.assign((size_t)m_numParticles * 4, 0.f);
    m_hostStale = true;
}

void ParticleSystem::enablePhaseTimings(bool on) { SPH_CHECK(sph_timing_enable(m_ctx, on ? 1 : 0)); }

bool ParticleSystem::phaseTimings(float ms[SPH_PH_COUNT], uint* steps) {
    uint32_t s = 0;
    SPH_CHECK(sph_timing_get(m_ctx, ms, &s));
    SPH_CHECK(sph_timing_reset(m_ctx));
    if (steps) *steps = s;
    return s > 0;
}

void ParticleSystem::setBenchmarkLog(const std::string& path, double min_interval_ms, BenchmarkLogStyle style) {
    m_logFreqMs = min_interval_ms;
    m_logStyle = (int)style;
    if (m_log) { fclose((FILE*)m_log); m_log = nullptr; }
    m_logPath = path;
    if (path.empty()) { SPH_CHECK(sph_timing_enable(m_ctx, 0)); return; }
    FILE* f = fopen(path.c_str(), "w");
    if (!f) { fprintf(stderr, "ParticleSystem: cannot open %s\n", path.c_str()); exit(EXIT_FAILURE); }
    fprintf(f, "SPH Particle Simulation Benchmark\nCompute mode: HIP\n");   // particleSystem.cpp:165-166
    m_log = f;
    SPH_CHECK(sph_timing_enable(m_ctx, 1));
}
