// include/particleSystem.h -- host-side C++ class with the public interface of the reference's
// ParticleSystem (/root/reference/SPH/particleSystem.h:48-112), backed by libsph_hip.so.
//
// A program written against the reference header (its main loop, SPH/particles.cpp:176-192,
// 230-246, 306-318) compiles against this one unchanged: same class name, constructor, enums,
// methods and argument meaning.  Differences, all deliberate:
//   * headless: no OpenGL.  getCurrentReadBuffer()/getColorBuffer() return 0; positions are read
//     with getArray(POSITION) or through getPositionsDevice() (the `gl_pos` analogue).  The
//     reference constructor needs a GL context even in -benchmark mode.
//   * only the GPU mode exists: SEQUENTIAL / OMP_PARALLEL abort with a message (no CPU fallback).
//   * the grid follows the box: nextPow2((uint)(boxDims/(0.66666f*h))) per axis; the reference
//     always uses the BOX_SIZE macro, i.e. 32^3 whatever the box (particleSystem.cpp:46, A.2-3).
//   * reset(CONFIG_GRID) builds an exact lattice with a counter-based jitter (see
//     gpufluidsimulator_amd/ic.py); the reference's ceil(powf(N,1/3)) + rand() is not portable.
//   * additive API named by the north star: getArray / setArray / setSimParams, plus
//     getDensities(), getPositionsDevice(), phaseTimings(), saveState()/loadState().
//   * errors abort the process like checkCudaErrors (common/inc/helper_cuda.h:566-579).
#ifndef SPH_PARTICLESYSTEM_H
#define SPH_PARTICLESYSTEM_H

#include <cstdint>
#include <string>
#include <vector>

#include "sph_hip.h"

typedef unsigned int uint;

#if !defined(__HIPCC__) && !defined(HIP_INCLUDE_HIP_HIP_VECTOR_TYPES_H) && !defined(__VECTOR_TYPES_H__)
// the two CUDA vector types the reference interface uses (vector_types.h / vector_functions.h)
struct float3 { float x, y, z; };
struct uint3 { unsigned int x, y, z; };
static inline float3 make_float3(float x, float y, float z) { float3 v = {x, y, z}; return v; }
#endif

// SPH/particles_kernel.cuh:36-50, field for field, so that setSimParams() accepts the
// reference's struct.  gravity / collider* are carried but, as in the reference, no kernel
// reads them.
struct SimParams {
    float3 colliderPos;
    float colliderRadius;
    float3 gravity;
    float particleRadius;
    float3 boxMin;
    float3 boxMax;
    float3 boxDims;
    uint gridDim;
};

class ParticleSystem {
public:
    enum ParticleComputeMode {
        SEQUENTIAL,
        OMP_PARALLEL,
        CUDA_PARALLEL,
        HIP_PARALLEL = CUDA_PARALLEL   // what actually runs here
    };

    // Runs on the HIP device chosen with sph_select_device() (default 0), as the reference runs on
    // the device findCudaDevice() made current (`-device=N`).
    ParticleSystem(uint numParticles, float3 boxDims, ParticleComputeMode mode);
    // additive: an explicit grid instead of nextPow2(box / (0.66666 h)) per axis (a 0 keeps the formula)
    ParticleSystem(uint numParticles, float3 boxDims, ParticleComputeMode mode, uint3 gridDims);
    ~ParticleSystem();

    enum ParticleConfig { CONFIG_RANDOM, CONFIG_GRID, _NUM_CONFIGS };
    enum ParticleArray { POSITION, VELOCITY };

    // n = m_solverIterations full time steps; fps is only logged (particleSystem.cpp:720,806)
    void update(float deltaTime, float fps);
    void reset(ParticleConfig config);

    int getNumParticles() const { return (int)m_numParticles; }
    unsigned int getCurrentReadBuffer() const { return 0; }   // no GL buffer in headless builds
    unsigned int getColorBuffer() const { return 0; }

    void dumpParticles(uint start, uint count);

    void setIterations(int i) { m_solverIterations = i; }
    void setGravity(float x) { m_params.gravity = make_float3(0.0f, x, 0.0f); }   // a physics no-op, as upstream
    void setColliderPos(float3 x) { m_params.colliderPos = x; }
    float3 getColliderPos() { return m_params.colliderPos; }
    float getColliderRadius() { return m_params.colliderRadius; }
    float getParticleRadius() { return m_params.particleRadius; }
    float3 getBoxMin() { return m_params.boxMin; }
    float3 getBoxMax() { return m_params.boxMax; }

    void addSphere(int index, float* pos, float* vel, int r, float spacing);

    // ---- additive (absent upstream; named by BASELINE.json's north star) ------------------------
    // 4 floats per particle, by creation index (the original NVIDIA sample's layout); the pointer
    // stays valid until the next getArray call.
    float* getArray(ParticleArray array);
    void setArray(ParticleArray array, const float* data, int start, int count);
    void setSimParams(const SimParams& p);
    const SimParams& getSimParams() const { return m_params; }
    const float* getDensities();                    // density per creation index
    void* getPositionsDevice();                     // device float4 (x,y,z,1) per creation index
    uint3 getGridSize() const { return m_grid; }
    sph_ctx* context() { return m_ctx; }
    // device-accurate per-phase times in ms since the last call (names as in dumpBenchmark)
    void enablePhaseTimings(bool on);
    bool phaseTimings(float ms[SPH_PH_COUNT], uint* steps);
    // checkpoint / resume (sph_snapshot_save / sph_snapshot_load); a resumed run is bit-identical
    void saveState(const std::string& path);
    void loadState(const std::string& path);
    // opt-in text log in the reference's dumpBenchmark format (particleSystem.cpp:697-716).  Two line styles, because the
    // reference has two: LOG_FRAMES is what its current source writes ("<int>sec ... frames:<n>frames"), LOG_OSCAR is the
    // form of its 18 committed logs (benchmarks/oscar/<N>/*.txt: "2.005sec ... FPS:189.322fps"), which is the only form its
    // benchmark.py:12 regex reads -- so a log written in that style can be summarised next to the published ones
    // (tools/bench_log_summary.py reads both).
    enum BenchmarkLogStyle { LOG_FRAMES = 0, LOG_OSCAR = 1 };
    void setBenchmarkLog(const std::string& path, double min_interval_ms = 2000.0 /* BENCHMARK_FREQ */,
                         BenchmarkLogStyle style = LOG_FRAMES);

protected:
    void _initialize(int numParticles);
    void _finalize();
    void uploadAll();
    void downloadAll();

    bool m_bInitialized;
    uint m_numParticles;
    std::vector<float> m_hPos, m_hVel, m_hDens;     // host mirrors: xyzw, xyzw, scalar
    std::vector<float> m_xyz, m_vxyz;                // packed xyz staging for the C ABI
    SimParams m_params;
    float3 m_boxDims;
    uint3 m_grid;
    uint m_solverIterations;
    ParticleComputeMode m_compute_mode;
    sph_ctx* m_ctx;
    bool m_hostStale;
    std::string m_logPath;
    void* m_log;
    double m_logLastMs, m_logGlobalMs, m_logFreqMs;
    int m_logStyle;
    unsigned long long m_logFrames;
};

extern "C" {
// Host-only twins of gpufluidsimulator_amd/ic.py (bit-identical output); no GPU needed.
void sph_ic_dam_break(const uint32_t lattice[3], const float box[3], int jitter, uint64_t start, uint64_t count,
                      float* pos_xyz, float* vel_xyz);
void sph_ic_random_box(uint64_t n, const float box[3], float speed, uint32_t seed, float fill, float* pos_xyz,
                       float* vel_xyz);
}

#endif  // SPH_PARTICLESYSTEM_H
