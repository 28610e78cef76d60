// sph_headless.cpp -- headless driver with the reference's command line
// (SPH/particles.cpp:676-706: -n= -box= -i= -benchmark -device= -file=) and its runBenchmark()
// output line (:176-192), on top of include/particleSystem.h.  No GLUT / OpenGL.
//   sph_headless -benchmark -n=262144 -box=8 -i=100 [-device=0] [-grid=128] [-ic=grid|random] [-steps=1] [-dump=8]
//                [-log=benchmark.txt [-logstyle=oscar|frames]] [-file=<snapshot>]
// Several GPUs (no counterpart in the reference, which is a single-device program): -gpus=N cuts the dam into N
// z-slabs and steps them with sph_slab_step through the C ABI --
//   -gpus=N            N child PROCESSES, forked before anything touches a GPU, rank r on device r (+ -device=), messages
//                      over RCCL (sph_rccl_transport_create; rank 0 hands the id to the others through pipes);
//   -gpus=N -onegpu    N THREADS of this process on one device over the device-to-device transport (a one-GPU box).
//   -protocol=1        the one-message slab step (sph_slab_set_protocol: two ghost layers, ghost densities recomputed locally).
// The reference's line is printed with NumDevsUsed = N.
#include "../../include/particleSystem.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <cerrno>
#include <fcntl.h>
#include <poll.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

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

// ---- -gpus=N: z-slabs through the C ABI ---------------------------------------------------------------------------
namespace {

struct SlabJob {
    uint32_t lattice[3];
    uint64_t particles;              // the first `particles` lattice points (creation index = x + nx (y + ny z))
    float box;
    uint32_t grid;
    int iterations, substeps;
    bool warmup;
    float dt;
    const char* out;                 // xyzw per creation index, then vxyz0 (as the one-device -out)
    int protocol;                    // message groups per step: 3, or 1 (-protocol=1: sph_slab_set_protocol, two ghost layers)
};

struct Plan { std::vector<uint32_t> cuts; std::vector<uint64_t> hist; uint64_t per_layer = 0; };

// Count-balanced cuts of whole cell layers, every slab at least two layers (gpufluidsimulator_amd/slab.py: choose_cuts).
// The lattice planes must not straddle a cell face (every BASELINE config: planes sit a quarter of a cell from the faces,
// the jitter is 0.08 of a cell), so that a slab is one run of creation indices generated on its own device.
bool plan_slabs(const SlabJob& j, int world, Plan& pl, std::string& err) {
    const uint32_t nx = j.lattice[0], ny = j.lattice[1], nz = j.lattice[2], gz = j.grid;
    const double radius = 1.0 / 64.0, spacing = 2.0 * radius, amp = 0.5 * j.box * 0.01 * radius + 1e-4;
    pl.hist.assign(gz, 0);
    auto layer = [&](double z) { long c = (long)std::floor((z + j.box / 2.0) / j.box * gz); return (uint32_t)std::min<long>(std::max<long>(c, 0), gz - 1); };
    for (uint32_t iz = 0; iz < nz; iz++) {
        const uint64_t first = (uint64_t)iz * nx * ny;
        if (first >= j.particles) break;
        const uint64_t cnt = std::min<uint64_t>((uint64_t)nx * ny, j.particles - first);
        const double zc = -j.box / 2.0 + radius + spacing * iz;
        if (layer(zc - amp) != layer(zc + amp)) { err = "a lattice plane straddles a cell face: use `python bench.py --gpus N`"; return false; }
        pl.hist[layer(zc)] += cnt;
    }
    const uint32_t min_layers = j.protocol == 1 ? 4u : 2u;      // (the one-message step: four layers per slab, slab.py choose_cuts)
    if ((uint64_t)world * min_layers > gz) { err = "more slabs than groups of " + std::to_string(min_layers) + " cell layers"; return false; }
    std::vector<uint64_t> prefix(gz + 1, 0);
    for (uint32_t z = 0; z < gz; z++) { prefix[z + 1] = prefix[z] + pl.hist[z]; pl.per_layer = std::max(pl.per_layer, pl.hist[z]); }
    pl.cuts.assign(1, 0u);
    for (int r = 1; r < world; r++) {
        const double target = (double)j.particles * r / world;
        uint32_t z = (uint32_t)(std::lower_bound(prefix.begin(), prefix.end(), (uint64_t)std::ceil(target)) - prefix.begin());
        if (z > 0 && std::fabs((double)prefix[z - 1] - target) <= std::fabs((double)prefix[std::min(z, gz)] - target)) z--;
        z = std::max(z, pl.cuts.back() + min_layers);
        z = std::min(z, gz - min_layers * (uint32_t)(world - r));
        pl.cuts.push_back(z);
    }
    pl.cuts.push_back(gz);
    return true;
}

struct RankResult { double seconds = 0.0; uint32_t owned = 0; double ping_us[3] = {0, 0, 0}; int rc = 0; char err[256] = {0}; };

// a barrier between the ranks: threads share a counter, processes talk to the parent through pipes.  vote(ok) is the same
// barrier carrying one bit each way: every rank says whether it is fine, everybody learns whether ALL are (the parent --
// or the last thread to arrive -- forms the AND).  The "ready" round in front of the transport uses it: a rank whose
// context, lattice or device set-up failed must stop the others BEFORE they enter ncclCommInitRank, where they would wait
// for it for ever (no time-out of the library covers that call).
struct Gate {
    std::mutex mu; std::condition_variable cv; int world = 1, waiting = 0; uint64_t round = 0;
    bool all_ok = true, result = true;
    int to_parent = -1, from_parent = -1;        // process mode
    bool vote(bool ok) {
        if (to_parent >= 0) {
            char b = ok ? 1 : 0;
            if (write(to_parent, &b, 1) != 1 || read(from_parent, &b, 1) != 1) _exit(3);      // the parent is gone
            return b == 1;
        }
        std::unique_lock<std::mutex> g(mu);
        const uint64_t my = round;
        all_ok = all_ok && ok;
        if (++waiting == world) { waiting = 0; result = all_ok; all_ok = true; round++; cv.notify_all(); }
        else cv.wait(g, [&] { return round != my; });
        return result;
    }
    void wait() { (void)vote(true); }
};

void rank_main(const SlabJob& j, const Plan& pl, int rank, int world, int device, sph_local_hub* hub, const uint8_t* rccl_id, Gate& gate,
               RankResult& res) {
    auto fail = [&](const char* what) { res.rc = -1; snprintf(res.err, sizeof res.err, "rank %d: %s: %s", rank, what, sph_last_error()); };
    const uint32_t z_lo = pl.cuts[rank], z_hi = pl.cuts[rank + 1];
    uint64_t n_own = 0, first = 0;
    {   // my lattice planes: one run of creation indices
        const uint32_t nx = j.lattice[0], ny = j.lattice[1];
        const double radius = 1.0 / 64.0, spacing = 2.0 * radius;
        bool any = false;
        for (uint32_t iz = 0; iz < j.lattice[2]; iz++) {
            const uint64_t f = (uint64_t)iz * nx * ny;
            if (f >= j.particles) break;
            const double zc = -j.box / 2.0 + radius + spacing * iz;
            long c = (long)std::floor((zc + j.box / 2.0) / j.box * j.grid);
            c = std::min<long>(std::max<long>(c, 0), (long)j.grid - 1);
            if ((uint32_t)c < z_lo || (uint32_t)c >= z_hi) continue;
            if (!any) { first = f; any = true; }
            n_own += std::min<uint64_t>((uint64_t)nx * ny, j.particles - f);
        }
    }
    const float dims[3] = {j.box, j.box, j.box};
    const uint32_t grid[3] = {j.grid, j.grid, j.grid};
    sph_params prm;
    sph_default_params(&prm, dims, grid);
    const uint32_t ghost_layers = j.protocol == 1 ? 2u : 1u;
    const uint32_t gcap = (uint32_t)((j.protocol == 1 ? 5 : 3) * pl.per_layer + 1024);
    const uint32_t cap = (uint32_t)(1.5 * (double)std::max<uint64_t>(n_own, j.particles / world)) + 4096u;
    sph_ctx* ctx = nullptr; sph_transport* tr = nullptr; sph_slab* slab = nullptr;
    bool ok = sph_create_slab_layers(&ctx, device, cap, &prm, z_lo, z_hi, gcap, ghost_layers) == 0;
    if (!ok) fail("sph_create_slab_layers");
    if (ok && n_own && sph_reset_lattice(ctx, j.lattice, 1, nullptr, first, (uint32_t)n_own) < 0) { ok = false; fail("sph_reset_lattice"); }
    if (ok && sph_sync(ctx) < 0) { ok = false; fail("set-up"); }
    // test hooks (tests/test_gpu_host_class.py): this rank fails its set-up / never reaches the READY round
    if (const char* e = getenv("SPH_HEADLESS_TEST_FAIL_SETUP")) if (ok && atoi(e) == rank) { ok = false; res.rc = -1; snprintf(res.err, sizeof res.err, "rank %d: set-up failed (test hook)", rank); }
    if (const char* e = getenv("SPH_HEADLESS_TEST_HANG_RANK")) if (atoi(e) == rank) for (;;) pause();
    // READY round: nobody creates its transport unless every rank has its context and particles (see Gate::vote)
    if (!gate.vote(ok)) {
        if (ok) { res.rc = -1; snprintf(res.err, sizeof res.err, "rank %d: stopped before the transport was created: another rank failed its set-up", rank); }
        if (ctx) sph_destroy(ctx);
        return;
    }
    if (ok && (hub ? sph_local_transport_create(&tr, hub, rank) : sph_rccl_transport_create(&tr, rccl_id, rank, world, device)) < 0) {
        ok = false; fail("transport");
    }
    if (ok && sph_slab_create(&slab, ctx, rank, world, tr, 0) < 0) { ok = false; fail("sph_slab_create"); }
    if (ok && j.protocol == 1 && sph_slab_set_protocol(slab, 1) < 0) { ok = false; fail("sph_slab_set_protocol"); }
    if (ok) {       // preflight: one exchange-shaped group at the step's three message sizes, contents checked
        const size_t sizes[3] = {8192, (size_t)std::min<uint64_t>(pl.per_layer * 32, (uint64_t)(gcap + 1) * 32),
                                 (size_t)std::min<uint64_t>(pl.per_layer * 8, (uint64_t)(gcap + 1) * 32)};
        for (int k = 0; k < 3 && ok; k++) {
            double o[3];
            if (sph_slab_ping(slab, std::max<size_t>(sizes[k] & ~(size_t)3, 4), 3, o) < 0) { ok = false; fail("sph_slab_ping"); }
            else res.ping_us[k] = o[0];
        }
        // the innermost layers' force pass goes in front of the step's wait (sph_slab_set_early_force)
        // unless the links are so fast that it cannot pay (a migrant message under ~12 us and a halo-A message under ~45)
        // -- nor when the slab is so big that the deep density launch, queued in front of the wait anyway, outlasts the two
        // messages on the path (~half the particles at ~18,800 per us) -- and never when the ranks share a device (-onegpu):
        // their big kernels would evict each other's L2 working sets.  (gpufluidsimulator_amd/slab.py: early_force_rule)
        if (ok) {
            const bool slow = res.ping_us[0] >= 12.0 || res.ping_us[1] >= 45.0;
            const bool exposed = res.ping_us[0] + res.ping_us[1] + 20.0 > 0.5 * (double)n_own / 18.8e3;
            sph_slab_set_early_force(slab, (!hub && slow && exposed) ? 1 : 0);
        }
    }
    if (ok && j.warmup && (sph_slab_step(slab, j.dt, (uint32_t)j.substeps) < 0 || sph_slab_sync(slab) < 0)) { ok = false; fail("warm-up step"); }
    gate.wait();
    const auto t0 = std::chrono::steady_clock::now();
    if (ok && (sph_slab_step(slab, j.dt, (uint32_t)(j.iterations * j.substeps)) < 0 || sph_slab_sync(slab) < 0)) { ok = false; fail("sph_slab_step"); }
    gate.wait();
    res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (ok) res.owned = sph_num_particles(ctx);
    if (ok && j.out) {          // every rank writes the rows of the particles it owns into the shared file
        const uint32_t n = res.owned;
        std::vector<float> p((size_t)n * 3), v((size_t)n * 3);
        std::vector<uint32_t> idx(n);
        if (sph_download_owned(ctx, p.data(), v.data(), idx.data()) < 0) { ok = false; fail("sph_download_owned"); }
        const int fd = ok ? open(j.out, O_RDWR) : -1;
        const size_t bytes = (size_t)j.particles * 4 * sizeof(float) * 2;
        float* m = fd >= 0 ? (float*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : (float*)MAP_FAILED;
        if (ok && m == (float*)MAP_FAILED) { ok = false; res.rc = -1; snprintf(res.err, sizeof res.err, "rank %d: cannot map %s", rank, j.out); }
        if (ok) {
            float* vel = m + (size_t)j.particles * 4;
            for (uint32_t s = 0; s < n; s++) {
                const size_t i = idx[s];
                if (i >= j.particles) continue;
                for (int a = 0; a < 3; a++) { m[4 * i + a] = p[3 * (size_t)s + a]; vel[4 * i + a] = v[3 * (size_t)s + a]; }
                m[4 * i + 3] = 1.0f; vel[4 * i + 3] = 0.0f;
            }
            munmap(m, bytes);
        }
        if (fd >= 0) close(fd);
    }
    if (slab) sph_slab_destroy(slab);
    if (tr) { if (hub) sph_local_transport_destroy(tr); else sph_rccl_transport_destroy(tr); }
    if (ctx) sph_destroy(ctx);
}

// a per-round deadline of the -gpus=N parent, in seconds (environment override for tests)
double deadline_for(const char* env, double dflt) { const char* e = getenv(env); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : dflt; }
bool read_all(int fd, void* p, size_t n) { char* c = (char*)p; while (n) { ssize_t k = read(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; } return true; }
bool write_all(int fd, const void* p, size_t n) { const char* c = (const char*)p; while (n) { ssize_t k = write(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; } return true; }

int run_slabs(const SlabJob& j, int world, bool onegpu, int device0) {
    setenv("SPH_SLAB_TIMEOUT_S", "30", 0);      // a rank that failed must not leave the others in the library's 120 s waits
    Plan pl; std::string err;
    if (!plan_slabs(j, world, pl, err)) { fprintf(stderr, "-gpus=%d: %s\n", world, err.c_str()); return EXIT_FAILURE; }
    if (j.out) {
        const int fd = open(j.out, O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0 || ftruncate(fd, (off_t)(j.particles * 4 * sizeof(float) * 2)) != 0) { fprintf(stderr, "cannot create %s\n", j.out); return EXIT_FAILURE; }
        close(fd);
    }
    printf("Run %llu particles simulation for %d iterations... (grid %u^3, box %g, %d z-slabs, cuts", (unsigned long long)j.particles,
           j.iterations, j.grid, j.box, world);
    for (uint32_t c : pl.cuts) printf(" %u", c);
    printf(")\n\n");
    fflush(stdout);
    std::vector<RankResult> res(world);
    if (onegpu) {       // N threads, one device, device-to-device transport
        int is950 = 0;
        if (sph_device_count(&is950) <= 0 || !is950 || sph_select_device(device0) < 0) { fprintf(stderr, "No gfx950 (MI355X) device found, exiting\n"); return EXIT_FAILURE; }
        sph_local_hub* hub = nullptr;
        if (sph_local_hub_create(&hub, world, device0) < 0) { fprintf(stderr, "%s\n", sph_last_error()); return EXIT_FAILURE; }
        Gate gate; gate.world = world;
        std::vector<std::thread> th;
        for (int r = 0; r < world; r++) th.emplace_back([&, r] { rank_main(j, pl, r, world, device0, hub, nullptr, gate, res[r]); });
        for (auto& t : th) t.join();
        sph_local_hub_destroy(hub);
    } else {            // N processes, forked BEFORE anything touches a GPU; rank r on device device0 + r, RCCL
        std::vector<int> id_pipe(2 * world, -1), up(2 * world, -1), down(2 * world, -1), out_pipe(2 * world, -1);
        for (int r = 0; r < world; r++)
            if (pipe(&id_pipe[2 * r]) || pipe(&up[2 * r]) || pipe(&down[2 * r]) || pipe(&out_pipe[2 * r])) { perror("pipe"); return EXIT_FAILURE; }
        std::vector<pid_t> kids(world);
        for (int r = 0; r < world; r++) {
            kids[r] = fork();
            if (kids[r] < 0) { perror("fork"); return EXIT_FAILURE; }
            if (kids[r] == 0) {
                // keep only this rank's ends of the pipes: a pipe whose other end is still open somewhere never reports that
                // its owner died (the parent's read would block for ever on a rank that crashed)
                for (int q = 0; q < world; q++) {
                    close(up[2 * q]); close(down[2 * q + 1]); close(out_pipe[2 * q]);
                    if (q != r) { close(up[2 * q + 1]); close(down[2 * q]); close(out_pipe[2 * q + 1]); close(id_pipe[2 * q]); }
                    if (r != 0 || q == 0) close(id_pipe[2 * q + 1]);
                }
                RankResult rr;
                uint8_t msg[129] = {0};                    // [0]: 1 = an RCCL id follows, 0 = rank 0 has none (abort)
                int ndev = sph_device_count(nullptr);
                if (ndev < device0 + world) { rr.rc = -1; snprintf(rr.err, sizeof rr.err, "rank %d: %d GPUs asked for, %d visible (one GPU: -onegpu)", r, device0 + world, ndev); }
                if (!rr.rc && sph_select_device(device0 + r) < 0) { rr.rc = -1; snprintf(rr.err, sizeof rr.err, "rank %d: %s", r, sph_last_error()); }
                if (r == 0) {
                    if (!rr.rc && sph_rccl_unique_id(msg + 1) < 0) { rr.rc = -1; snprintf(rr.err, sizeof rr.err, "rank 0: %s", sph_last_error()); }
                    msg[0] = rr.rc ? 0 : 1;                // an explicit "no id" instead of 128 zeros the others would try to connect with
                    for (int q = 1; q < world; q++) write_all(id_pipe[2 * q + 1], msg, sizeof msg);
                } else if (!read_all(id_pipe[2 * r], msg, sizeof msg)) { if (!rr.rc) { rr.rc = -1; snprintf(rr.err, sizeof rr.err, "rank %d: no RCCL id from rank 0", r); } }
                else if (msg[0] != 1 && !rr.rc) { rr.rc = -1; snprintf(rr.err, sizeof rr.err, "rank %d: rank 0 could not make an RCCL id", r); }
                Gate gate; gate.to_parent = up[2 * r + 1]; gate.from_parent = down[2 * r];
                if (!rr.rc) rank_main(j, pl, r, world, device0 + r, nullptr, msg + 1, gate, rr);
                else (void)gate.vote(false);               // the READY round: the parent tells everybody to stop
                write_all(out_pipe[2 * r + 1], &rr, sizeof rr);
                _exit(rr.rc ? 1 : 0);
            }
        }
        signal(SIGPIPE, SIG_IGN);                      // (a write to a rank that died must be an error code, not this process's end)
        for (int r = 0; r < world; r++) { close(up[2 * r + 1]); close(down[2 * r]); close(out_pipe[2 * r + 1]); close(id_pipe[2 * r]); close(id_pipe[2 * r + 1]); }
        // The parent never blocks on one rank: it polls every live rank's pipe with a DEADLINE per round.  A rank that hangs
        // where no time-out of the library reaches (ncclCommInitRank waiting for a rank that never comes, a wedged device)
        // would otherwise hold the whole job for ever: on expiry the children are killed, reaped, and the run fails.
        std::vector<bool> dead(world, false);
        auto kill_all = [&](const char* why) {
            fprintf(stderr, "-gpus=%d: %s: killing the ranks\n", world, why);
            for (int r = 0; r < world; r++) kill(kids[r], SIGKILL);
            for (int r = 0; r < world; r++) { int st = 0; waitpid(kids[r], &st, 0); }
        };
        // one byte from every live rank (0/1), then the AND of them back to every live rank; false: deadline expired
        auto round_trip = [&](const char* name, double deadline_s, bool& all_ok) -> bool {
            std::vector<int> got(world, -1);
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                std::vector<pollfd> fds; std::vector<int> who;
                for (int r = 0; r < world; r++) if (!dead[r] && got[r] < 0) { fds.push_back(pollfd{up[2 * r], POLLIN, 0}); who.push_back(r); }
                if (fds.empty()) break;
                const double left = deadline_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (left <= 0.0) {
                    std::string late;
                    for (int r : who) late += " " + std::to_string(r);
                    fprintf(stderr, "-gpus=%d: no answer at the '%s' barrier from rank(s)%s after %.0f s\n", world, name, late.c_str(), deadline_s);
                    return false;
                }
                const int k = poll(fds.data(), (nfds_t)fds.size(), (int)std::min(left * 1e3 + 1.0, 1000.0));
                if (k < 0 && errno != EINTR) { perror("poll"); return false; }
                for (size_t i = 0; k > 0 && i < fds.size(); i++) {
                    if (!(fds[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
                    char b = 0;
                    if (read(fds[i].fd, &b, 1) == 1) got[who[i]] = b ? 1 : 0;
                    else { dead[who[i]] = true; fprintf(stderr, "rank %d died\n", who[i]); }      // (end of file: it closed its end)
                }
            }
            all_ok = true;
            for (int r = 0; r < world; r++) all_ok = all_ok && !dead[r] && got[r] == 1;
            const char b = all_ok ? 1 : 0;
            for (int r = 0; r < world; r++) if (!dead[r] && write(down[2 * r + 1], &b, 1) != 1) dead[r] = true;
            return true;
        };
        const double d_setup = deadline_for("SPH_HEADLESS_SETUP_S", 300.0), d_run = deadline_for("SPH_HEADLESS_RUN_S", 3600.0);
        bool go = false, dummy = false;
        if (!round_trip("ready", d_setup, go)) { kill_all("set-up did not finish"); return EXIT_FAILURE; }
        if (go) {                                      // the two barriers of rank_main: in front of and behind the timed steps
            if (!round_trip("start", d_setup, dummy)) { kill_all("transport set-up / warm-up did not finish"); return EXIT_FAILURE; }
            if (!round_trip("end", d_run, dummy)) { kill_all("the timed steps did not finish"); return EXIT_FAILURE; }
        }
        for (int r = 0; r < world; r++) { if (dead[r] || !read_all(out_pipe[2 * r], &res[r], sizeof(RankResult))) { res[r].rc = -1; snprintf(res[r].err, sizeof res[r].err, "rank %d: no result", r); } }
        for (int r = 0; r < world; r++) { int st = 0; waitpid(kids[r], &st, 0); }
    }
    double secs = 0.0; uint64_t owned = 0; bool bad = false;
    for (int r = 0; r < world; r++) {
        if (res[r].rc) { fprintf(stderr, "%s\n", res[r].err); bad = true; }
        secs = std::max(secs, res[r].seconds); owned += res[r].owned;
    }
    if (bad) return EXIT_FAILURE;
    if (owned != j.particles) { fprintf(stderr, "the slabs hold %llu of %llu particles\n", (unsigned long long)owned, (unsigned long long)j.particles); return EXIT_FAILURE; }
    const double avg = secs / (j.iterations > 0 ? j.iterations : 1);
    printf("particles, Throughput = %.4f KParticles/s, Time = %.5f s, Size = %llu particles, NumDevsUsed = %u, Workgroup = %u\n",
           (1.0e-3 * (double)j.particles) / avg, avg, (unsigned long long)j.particles, (unsigned)world, 0);
    printf("{\"particle_steps_per_s\": %.1f, \"particles\": %llu, \"iterations\": %d, \"steps_per_update\": %d, \"seconds\": %.6f, "
           "\"gpus\": %d, \"ranks_as\": \"%s\", \"ping_us\": [%.1f, %.1f, %.1f]}\n",
           (double)j.particles * j.iterations * j.substeps / secs, (unsigned long long)j.particles, j.iterations, j.substeps, secs, world,
           onegpu ? "threads" : "processes", res[0].ping_us[0], res[0].ping_us[1], res[0].ping_us[2]);
    return 0;
}

}  // namespace

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
    // -file=<path>: the reference runs ONE update and would compare with a reference file (particles.cpp:690-692, 194-217;
    // the comparison itself is commented out upstream).  Here the file is a state snapshot (-save): it is loaded like -load.
    const char* file = value(argc, argv, "file");
    const char* load = value(argc, argv, "load");
    if (file && !load) load = file;
    if (file && !value(argc, argv, "i")) iterations = 1;                    // numIterations = 1, particles.cpp:692
    // -benchmark selects the headless path upstream (without it a GLUT window opens); this build has no other path
    const bool benchmark = flag(argc, argv, "benchmark");
    if (!benchmark && !file) fprintf(stderr, "note: no OpenGL in this build -- running headless, as with -benchmark\n");
    if (flag(argc, argv, "help")) {
        printf("usage: sph_headless [-benchmark] [-n=<particles>] [-box=<edge>] [-i=<iterations>] [-device=<id>] [-grid=<cells per axis>] "
               "[-ic=grid|random] [-steps=<per update>] [-gpus=<N> [-onegpu] [-slab] [-lattice=nx,ny,nz] [-protocol=1|3]] "
               "[-dump=<count>] [-log=<file> [-logfreq=<ms>] [-logstyle=oscar|frames]] [-sphere=<update>[,<radius>]] [-out=<file>] [-save=<file>] [-load=<file>] [-file=<file>]\n");
        return 0;
    }
    const int gpus = value(argc, argv, "gpus") ? atoi(value(argc, argv, "gpus")) : 1;
    if (gpus > 1 || (gpus == 1 && flag(argc, argv, "slab"))) {     // (-gpus=1 -slab: the same machinery with one rank -- fork, RCCL communicator of one)
        if (ic != ParticleSystem::CONFIG_GRID || load || value(argc, argv, "sphere") || value(argc, argv, "save") || value(argc, argv, "log") || dump) {
            fprintf(stderr, "-gpus=%d runs the dam-break lattice (-ic=grid) and takes -n -box -grid -i -steps -lattice -out -device -onegpu -nowarmup\n", gpus);
            return EXIT_FAILURE;
        }
        SlabJob j{};
        uint32_t s = (uint32_t)std::floor(std::cbrt((double)numParticles));
        while ((uint64_t)s * s * s < numParticles) s++;                     // the lattice of reset(CONFIG_GRID)
        j.lattice[0] = j.lattice[1] = j.lattice[2] = s;
        j.particles = numParticles;
        if (const char* v = value(argc, argv, "lattice")) {
            if (sscanf(v, "%u,%u,%u", &j.lattice[0], &j.lattice[1], &j.lattice[2]) != 3) { fprintf(stderr, "-lattice=nx,ny,nz\n"); return EXIT_FAILURE; }
            j.particles = (uint64_t)j.lattice[0] * j.lattice[1] * j.lattice[2];
        }
        j.box = box;
        j.grid = gridDim ? gridDim : sph_grid_dim_for_edge(box, 0.1f);
        j.iterations = iterations; j.substeps = substeps; j.warmup = !flag(argc, argv, "nowarmup"); j.dt = timestep;
        j.out = value(argc, argv, "out");
        j.protocol = value(argc, argv, "protocol") ? atoi(value(argc, argv, "protocol")) : 3;
        if (j.protocol != 1 && j.protocol != 3) { fprintf(stderr, "-protocol=%d: 3 (three message groups per step) or 1 (one)\n", j.protocol); return EXIT_FAILURE; }
        return run_slabs(j, gpus, flag(argc, argv, "onegpu"), device);
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
    if (const char* v = load) {
        psystem->loadState(v);
        numParticles = (uint)psystem->getNumParticles();  // the snapshot decides how many particles there are
    }
    psystem->setIterations(substeps);
    if (const char* v = value(argc, argv, "log")) {
        const char* fq = value(argc, argv, "logfreq");
        // -logstyle=oscar: the line form of the reference's committed logs (benchmarks/oscar/, what its benchmark.py reads);
        // default: what its current source writes (particleSystem.cpp:703-714)
        const char* st = value(argc, argv, "logstyle");
        if (st && strcmp(st, "oscar") && strcmp(st, "frames")) { fprintf(stderr, "-logstyle=%s: expected oscar or frames\n", st); return EXIT_FAILURE; }
        psystem->setBenchmarkLog(v, fq ? atof(fq) : 2000.0, st && !strcmp(st, "oscar") ? ParticleSystem::LOG_OSCAR : ParticleSystem::LOG_FRAMES);
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
    delete psystem;
    return 0;
}
