// sph_headless.cpp -- headless driver with the reference's command line
// (SPH/particles.cpp:676-706: -n= -box= -i= -benchmark -device=) and its runBenchmark()
// output line (:176-192), on top of include/particleSystem.h.  No GLUT / OpenGL.
//   sph_headless -benchmark -n=262144 -box=8 -i=100 [-device=0] [-grid=128] [-ic=grid|random] [-steps=1] [-dump=8]
//                [-log=benchmark.txt]
// Several GPUs: one process per GPU (z-slabs, RCCL) -- `python bench.py --gpus N`; this driver is the
// reference's single-device program.
#include "../../include/particleSystem.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static bool flag(int argc, char** argv, const char* name) {   // checkCmdLineFlag, helper_string.h:111
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        while (*a == '-') a++;
        size_t l = strlen(name);
        if (!strncmp(a, name, l) && (a[l] == 0 || a[l] == '=')) return true;
    }
    return false;
}

static const char* value(int argc, char** argv, const char* name) {
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        while (*a == '-') a++;
        size_t l = strlen(name);
        if (!strncmp(a, name, l) && a[l] == '=') return a + l + 1;
    }
    return nullptr;
}

int main(int argc, char** argv) {
    uint numParticles = 262144;            // NUM_PARTICLES, particleSystem.h:15
    float box = 2.0f;                      // BOX_SIZE
    int iterations = 100, substeps = 1, dump = 0;
    const float timestep = 0.0000005f;     // particles.cpp:88
    if (const char* v = value(argc, argv, "n")) numParticles = (uint)strtoul(v, nullptr, 10);
    if (const char* v = value(argc, argv, "box")) box = (float)atof(v);
    if (const char* v = value(argc, argv, "i")) iterations = atoi(v);
    if (const char* v = value(argc, argv, "steps")) substeps = atoi(v);
    if (const char* v = value(argc, argv, "dump")) dump = atoi(v);
    int device = 0;                        // findCudaDevice: -device=N, else the first device (helper_cuda.h:845)
    if (const char* v = value(argc, argv, "device")) device = atoi(v);
    uint gridDim = 0;                      // 0: the reference's formula nextPow2(box / (0.66666 h))
    if (const char* v = value(argc, argv, "grid")) gridDim = (uint)strtoul(v, nullptr, 10);
    ParticleSystem::ParticleConfig ic = ParticleSystem::CONFIG_GRID;   // initParticleSystem resets to CONFIG_GRID
    if (const char* v = value(argc, argv, "ic")) {
        if (!strcmp(v, "random")) ic = ParticleSystem::CONFIG_RANDOM;
        else if (strcmp(v, "grid")) { fprintf(stderr, "-ic=%s: expected grid or random\n", v); return EXIT_FAILURE; }
    }
    if (value(argc, argv, "gpus") && atoi(value(argc, argv, "gpus")) > 1) {
        fprintf(stderr, "-gpus=N>1: one process per GPU -- launch `python bench.py --gpus N` (z-slabs over RCCL)\n");
        return EXIT_FAILURE;
    }
    const bool benchmark = flag(argc, argv, "benchmark");
    if (flag(argc, argv, "help")) {
        printf("usage: sph_headless [-benchmark] [-n=<particles>] [-box=<edge>] [-i=<iterations>] [-device=<id>] [-grid=<cells per axis>] "
               "[-ic=grid|random] [-steps=<per update>] "
               "[-dump=<count>] [-log=<file>] [-sphere=<update>[,<radius>]] [-out=<file>] [-save=<file>] [-load=<file>]\n");
        return 0;
    }
    int is950 = 0;
    if (sph_device_count(&is950) <= 0 || !is950) {
        fprintf(stderr, "No gfx950 (MI355X) device found, exiting\n");   // cudaInit, particleSystem.cu:432-435
        return EXIT_FAILURE;
    }
    if (sph_select_device(device) < 0) {                  // cudaInit -> findCudaDevice -> cudaSetDevice
        fprintf(stderr, "-device=%d: %s\n", device, sph_last_error());
        return EXIT_FAILURE;
    }
    ParticleSystem* psystem = new ParticleSystem(numParticles, make_float3(box, box, box), ParticleSystem::HIP_PARALLEL,
                                                 uint3{gridDim, gridDim, gridDim});
    psystem->reset(ic);                                   // initParticleSystem, particles.cpp:119-132
    if (const char* v = value(argc, argv, "load")) {
        psystem->loadState(v);
        numParticles = (uint)psystem->getNumParticles();  // the snapshot decides how many particles there are
    }
    psystem->setIterations(substeps);
    if (const char* v = value(argc, argv, "log")) {
        const char* fq = value(argc, argv, "logfreq");
        psystem->setBenchmarkLog(v, fq ? atof(fq) : 2000.0);
    }
    uint3 g = psystem->getGridSize();
    printf("Run %u particles simulation for %d iterations... (grid %ux%ux%u, box %g)\n\n", numParticles, iterations, g.x, g.y, g.z, box);
    if (!flag(argc, argv, "nowarmup")) psystem->update(timestep, 0);   // warm-up step, not timed
    sph_sync(psystem->context());
    auto t0 = std::chrono::steady_clock::now();
    // -sphere=<k>[,<r>]: before update k, drop a ball of (2r+1)^3 lattice points (radius r spacings) at the top of
    // the box the way the GUI's key '4' does (addSphere, particles.cpp:306-318; centred here instead of frand())
    int sphereAt = -1, ballr = 10;
    if (const char* v = value(argc, argv, "sphere")) {
        sphereAt = atoi(v);
        if (const char* c = strchr(v, ',')) ballr = atoi(c + 1);
    }
    for (int i = 0; i < iterations; ++i) {
        if (i == sphereAt) {
            const float pr = psystem->getParticleRadius(), tr = pr + (pr * 2.0f) * ballr;
            float pos[4] = {0.0f, psystem->getBoxMax().y - tr, 0.0f, 0.0f}, vel[4] = {0.f, 0.f, 0.f, 0.f};
            psystem->addSphere(0, pos, vel, ballr, pr * 2.0f);
        }
        psystem->update(timestep, (float)i);
    }
    sph_sync(psystem->context());
    auto t1 = std::chrono::steady_clock::now();
    const double secs = std::chrono::duration<double>(t1 - t0).count();
    const double avg = secs / (iterations > 0 ? iterations : 1);
    // the reference's line, particles.cpp:191-192
    printf("particles, Throughput = %.4f KParticles/s, Time = %.5f s, Size = %u particles, NumDevsUsed = %u, Workgroup = %u\n",
           (1.0e-3 * numParticles) / avg, avg, numParticles, 1, 0);
    printf("{\"particle_steps_per_s\": %.1f, \"particles\": %u, \"iterations\": %d, \"steps_per_update\": %d, \"seconds\": %.6f}\n",
           (double)numParticles * iterations * substeps / secs, numParticles, iterations, substeps, secs);
    if (dump > 0) psystem->dumpParticles(0, (uint)dump);
    if (const char* v = value(argc, argv, "out")) {      // xyzw per creation index, then vxyz0, raw fp32
        FILE* f = fopen(v, "wb");
        if (!f) { fprintf(stderr, "cannot open %s\n", v); return EXIT_FAILURE; }
        fwrite(psystem->getArray(ParticleSystem::POSITION), sizeof(float) * 4, numParticles, f);
        fwrite(psystem->getArray(ParticleSystem::VELOCITY), sizeof(float) * 4, numParticles, f);
        fclose(f);
    }
    if (const char* v = value(argc, argv, "save")) psystem->saveState(v);
    (void)benchmark;
    delete psystem;
    return 0;
}
