/* include/sph_hip.h -- C ABI of libsph_hip.so, the MI355X (gfx950) SPH step library.
 *
 * This is the drop-in boundary for the hot path of oadrian/GPUFluidSimulator: it
 * replaces the extern "C" seam between the host class and the CUDA kernels,
 * /root/reference/SPH/particleSystem.cuh:3-30 (definitions particleSystem.cu:422-535).
 * The reference seam passes an 88-byte AoS `Particle*` and a device-resident
 * `SimParams*` to every call; here the device state (sorted SoA arrays, cell table,
 * sort scratch) lives in an opaque context and parameters travel by value.  A
 * binary-compatible rendition of the reference's own 19 symbols is declared in
 * include/sph_compat_seam.h on top of these entry points.
 *
 * Conventions
 *   - plain C types only; every pointer is a HOST pointer unless the name ends in
 *     `_dev` (then it is a device pointer on the context's GPU);
 *   - every function returns 0 on success or a negative SPH_E* code; the message is
 *     available from sph_last_error() (thread-local).  The reference's convention is
 *     abort-on-error (checkCudaErrors -> exit, common/inc/helper_cuda.h:566-579); the
 *     host class in include/particleSystem.h restores that behaviour on top;
 *   - all work is enqueued on the context's stream (sph_set_stream; default: the NULL
 *     stream like the reference) and is asynchronous unless the function returns data
 *     to the host; one context must not be used from two host threads at once;
 *   - there is NO CPU fallback: without a gfx950 device sph_create fails.
 *
 * What is ABI and what is not (every entry point belongs to exactly one class; the class is repeated as a tag in front of the
 * declarations that are NOT stable ABI):
 *   STABLE ABI     what a host class, a driver or a launcher binds; kept across versions (SPH_ABI_VERSION changes otherwise):
 *                  sph_abi_version sph_last_error sph_device_count sph_select_device sph_selected_device sph_default_params
 *                  sph_grid_dim_for_edge | sph_create sph_create_slab sph_create_slab_layers sph_destroy sph_set_stream
 *                  sph_set_params sph_get_params sph_sync sph_num_particles sph_capacity sph_ghost_layers sph_set_precision
 *                  sph_get_precision | sph_upload sph_set_by_index sph_reset_lattice sph_download sph_download_owned
 *                  sph_positions_dev sph_download_positions4 sph_snapshot_save sph_snapshot_load sph_snapshot_info |
 *                  sph_hash sph_sort sph_build_cells sph_density sph_force sph_collide sph_integrate sph_step sph_step_phased
 *                  sph_force_collide_integrate | sph_timing_enable sph_timing_get sph_timing_reset | the z-slab phase calls
 *                  (sph_migrants_* sph_slab_counts sph_halo_* sph_layer_histogram) | sph_rccl_unique_id
 *                  sph_rccl_transport_create/_destroy sph_local_hub_* sph_local_transport_* | sph_slab_create sph_slab_destroy
 *                  sph_slab_step sph_slab_sync sph_slab_set_wait_timeout sph_slab_set_protocol sph_slab_recut sph_slab_ping
 *                  sph_slab_failed sph_slab_timing_enable/_reset/_get
 *   [introspect]   read-only views for parity tests, profiles and bench lines; may grow or change with the implementation:
 *                  sph_get_keys sph_get_order sph_get_cell_range sph_get_cells sph_cell_key sph_download_forces sph_sort_stats
 *                  sph_sort_forms sph_last_sort_skipped sph_slab_stats sph_slab_counters sph_slab_in_place_merges
 *                  sph_slab_exchanges sph_slab_recut_stats sph_slab_early_force_stats sph_slab_protocol sph_rccl_transport_info
 *   [tuning]       switches whose defaults are the product's settings; in fp32 results do not depend on them (A/B runs):
 *                  sph_set_sort_mode sph_set_direct_hull sph_set_pair_small_launch sph_set_block_order sph_slab_set_early_force
 *   [test hook]    exist for tests and measurement harnesses only: sph_test_trust_mover_hint sph_slab_test_raise_flag
 *                  sph_rccl_transport_selftest sph_loop_transport_create/_destroy
 */
#ifndef SPH_HIP_H
#define SPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_ABI_VERSION 2

enum {
    SPH_OK = 0,
    SPH_E_INVALID = -1,   /* bad argument */
    SPH_E_DEVICE = -2,    /* HIP runtime error, no device, wrong architecture */
    SPH_E_NOMEM = -3,     /* device or host allocation failed */
    SPH_E_CAPACITY = -4,  /* more particles / ghosts than the context was created for */
    SPH_E_STATE = -5,     /* phase called out of order */
    SPH_E_PEER = -6       /* slab step: a neighbouring rank reported a failure and stopped (its own error says why) */
};

typedef struct sph_ctx sph_ctx;

/* Replaces `struct SimParams` (SPH/particles_kernel.cuh:36-50) plus the physics macros of
 * particles_kernel.cuh:20-33, which the reference bakes in at compile time.  The reference
 * never reads SimParams.gravity / colliderPos / colliderRadius in any kernel (setGravity is a
 * physics no-op), so they are not part of the device parameters. */
typedef struct sph_params {
    float box_min[3];        /* SimParams.boxMin                                   */
    float box_max[3];        /* SimParams.boxMax                                   */
    uint32_t grid[3];        /* cells per axis; the reference uses gridDim for all three */
    float h;                 /* m_H            0.1                                  */
    float mass;              /* MASS           65                                   */
    float rest_density;      /* REST_DENS      1000                                 */
    float gas_constant;      /* GAS_CONSTANT   2000                                 */
    float viscosity;         /* VISC           250                                  */
    float gravity_y;         /* GRAVITY * G_MODIFIER = -9.81f * 11000               */
    float wall_eps;          /* EPS_F          1e-5                                 */
    float wall_damping;      /* DAMPING_FACTOR -0.75 (particleSystem.cu:376)        */
    float restitution;       /* RESTITUTION    0                                    */
    float collision_param;   /* COLLISION_PARAM 1.0                                 */
    float particle_radius;   /* SimParams.particleRadius 1/64 (particleSystem.cpp:51) */
} sph_params;

/* Phases of one step, in the order of ParticleSystem::update's CUDA branch
 * (SPH/particleSystem.cpp:773-795); names follow dumpBenchmark (:697-716). */
enum {
    SPH_PH_ZINDEX = 0,   /* cudaMapZIndex            -> sph_hash            */
    SPH_PH_SORT,         /* cudaSortParticles        -> sph_sort (radix sort + reorder) */
    SPH_PH_BGRID,        /* cudaConstructBGrid + cudaConstructGridArray -> sph_build_cells */
    SPH_PH_DENS,         /* cudaComputeDensities     -> sph_density         */
    SPH_PH_FORCE,        /* cudaComputeForces        -> sph_force           */
    SPH_PH_COLLISION,    /* cudaParticleCollisions   -> sph_collide         */
    SPH_PH_INTEGRATE,    /* cudaIntegrate            -> sph_integrate       */
    SPH_PH_COUNT
};

/* ---- library ------------------------------------------------------------------------- */
int sph_abi_version(void);
const char* sph_last_error(void);
/* number of visible HIP devices, or a negative code; *is_gfx950 (optional) tells whether
 * device 0 is a gfx950.  Replaces cudaInit/findCudaDevice (particleSystem.cu:427-436). */
int sph_device_count(int* is_gfx950);

/* Choose the HIP device that contexts created with device = -1 use (default 0): the cudaSetDevice of
 * findCudaDevice behind the reference's `-device=N` flag (common/inc/helper_cuda.h:845,
 * particleSystem.cu:427-436, particles.cpp:676-706).  Fails if the device is not a gfx950. */
int sph_select_device(int device);
int sph_selected_device(void);

/* Fill *p with the reference's constants for a box of dimensions box_dims centred on the
 * origin (particleSystem.cpp:50-62) and the given grid. */
void sph_default_params(sph_params* p, const float box_dims[3], const uint32_t grid[3]);
/* nextPow2((uint)(edge / (0.66666f * h))): the reference's grid formula, particleSystem.cpp:46 */
uint32_t sph_grid_dim_for_edge(float edge, float h);

/* ---- context ------------------------------------------------------------------------- */
/* Whole-domain context for up to `capacity` particles on HIP device `device` (-1: the device chosen
 * with sph_select_device).
 * Replaces the allocateArray calls of _initialize (particleSystem.cpp:121-124). */
int sph_create(sph_ctx** out, int device, uint32_t capacity, const sph_params* p);
/* z-slab context (multi-GPU): owns the cell layers [z_lo, z_hi) of the global grid and keeps
 * one ghost layer on either side (up to ghost_capacity particles each). */
int sph_create_slab(sph_ctx** out, int device, uint32_t capacity, const sph_params* p,
                    uint32_t z_lo, uint32_t z_hi, uint32_t ghost_capacity);
/* The same with `ghost_layers` = 1 or 2 ghost cell layers per side (sph_create_slab: 1).  Two are what the ONE-MESSAGE slab
 * step needs (sph_slab_set_protocol): a rank then holds two layers of each neighbour's particles and recomputes the densities
 * of the inner one itself, instead of receiving them in a message of their own.  ghost_capacity then bounds the particles
 * of BOTH layers of a side. */
int sph_create_slab_layers(sph_ctx** out, int device, uint32_t capacity, const sph_params* p,
                           uint32_t z_lo, uint32_t z_hi, uint32_t ghost_capacity, uint32_t ghost_layers);
uint32_t sph_ghost_layers(const sph_ctx* c);        /* 0 for a whole-domain context */
void sph_destroy(sph_ctx* c);                       /* freeArray x4, particleSystem.cpp:180-183 */
int sph_set_stream(sph_ctx* c, void* hip_stream);   /* hipStream_t; NULL = default stream */
int sph_set_params(sph_ctx* c, const sph_params* p);/* the per-update SimParams upload, :723 */
int sph_get_params(const sph_ctx* c, sph_params* p);
int sph_sync(sph_ctx* c);                           /* threadSync, particleSystem.cu:467 */
uint32_t sph_num_particles(const sph_ctx* c);       /* owned particles */
uint32_t sph_capacity(const sph_ctx* c);

/* Arithmetic of the neighbour passes.  SPH_PRECISION_F32 (default): everything in fp32, the reference's
 * precision.  SPH_PRECISION_MIXED_F16 (BASELINE config 5): positions, velocities, densities and forces are
 * stored and integrated in fp32; inside the DENSITY traversal the per-pair arithmetic and the per-row
 * accumulators are packed fp16 (two candidates per lane-instruction) on coordinates relative to a reference point
 * of the wave in units of h (x as a coarse + a fine half, one reference per group of lanes that lie within 6 h in
 * y and z: exact at any extent of the wave), with NORMALISED kernel sums (the reference's densities ~2e6 do not
 * fit fp16); row sums are added up in fp32 and scaled once.  The force and collision passes stay fp32 (decided on
 * a device measurement).  Looser tolerance: DESIGN.md section 4. */
enum { SPH_PRECISION_F32 = 0, SPH_PRECISION_MIXED_F16 = 1 };
int sph_set_precision(sph_ctx* c, int precision);
int sph_get_precision(const sph_ctx* c);

/* ---- state transfer ------------------------------------------------------------------- */
/* Replace the particle set: n particles, xyz triples; index[i] is the immutable creation
 * index (Particle::index), NULL = 0..n-1.  Replaces copyArrayToDevice of the AoS array
 * (particleSystem.cpp:920,960).  Density/pressure/forces are reset to 0. */
int sph_upload(sph_ctx* c, uint32_t n, const float* pos_xyz, const float* vel_xyz, const uint32_t* index);
/* Overwrite position and/or velocity (xyz triples, either may be NULL) of the particles whose creation
 * index lies in [first_index, first_index + count), wherever they sit in the sorted arrays -- no
 * round trip of the whole state: replaces the host edit + copyArrayToDevice(start, count) of addSphere /
 * setArray (particleSystem.cpp:928-961).  Densities and forces of the current step become stale: the
 * next step starts with sph_hash as after sph_upload (the sort still only moves the particles whose
 * cell changed).  A slab context changes the particles it owns and ignores the others. */
int sph_set_by_index(sph_ctx* c, uint32_t first_index, uint32_t count, const float* pos_xyz, const float* vel_xyz);
/* Generate the dam-break lattice ON THE DEVICE (no host arrays, no PCIe): particles with creation index
 * index_start .. index_start+count of an (nx, ny, nz) lattice in the min corner of the box, spacing 2R,
 * zero velocity, counter-based jitter (amplitude from jitter_dims, NULL = the box; jitter = 0 switches it
 * off).  Bit-identical to sph_ic_dam_break / gpufluidsimulator_amd.ic.dam_break_lattice.  Replaces
 * initGrid + the AoS upload of reset(CONFIG_GRID) (SPH/particleSystem.cpp:839-874, 909-920). */
int sph_reset_lattice(sph_ctx* c, const uint32_t lattice[3], int jitter, const float jitter_dims[3],
                      uint64_t index_start, uint32_t count);
/* Gather the state BY CREATION INDEX relative to index_base: out[(index-index_base)*3+k].  The output
 * arrays hold index_count entries (x3 for the vectors); particles whose creation index lies outside
 * [index_base, index_base + index_count) are skipped, nothing beyond the arrays is written.
 * Any pointer may be NULL.  (The reference never copies particles back in CUDA mode; this is
 * the additive getArray of the north star.) */
int sph_download(sph_ctx* c, uint32_t index_base, uint32_t index_count, float* pos_xyz, float* vel_xyz, float* density,
                 float* pressure);
/* The owned particles compactly, in slot order: n x xyz, n x xyz, n creation indices (any may be NULL).
 * What a slab driver needs to move whole particles between ranks (re-balancing). */
int sph_download_owned(sph_ctx* c, float* pos_xyz, float* vel_xyz, uint32_t* index);
/* [introspect] */ int sph_download_forces(sph_ctx* c, uint32_t index_base, uint32_t index_count, float* fpress_xyz, float* fvisc_xyz,
                        float* dv_xyz, int32_t* collision_count);
/* The `gl_pos` analogue of cudaIntegrate (particleSystem.cu:416-419): float4 (x,y,z,1) per
 * creation index, written by sph_integrate / sph_step.  Device pointer, n*16 bytes. */
int sph_positions_dev(sph_ctx* c, void** out_dev);
int sph_download_positions4(sph_ctx* c, float* pos_xyzw);

/* State snapshot (checkpoint / resume; absent in the reference, whose device state is never
 * serialised -- SURVEY.md section 5).  The file holds the parameters and, IN SLOT ORDER, position,
 * velocity and creation index of every owned particle, so that a resumed run repeats the
 * original one bit for bit (the sort is stable).  Little-endian, see csrc/sph_capi.hip. */
int sph_snapshot_save(sph_ctx* c, const char* path);
/* Load into an existing context (same grid; capacity >= the stored particle count). */
int sph_snapshot_load(sph_ctx* c, const char* path);
/* Number of particles and the parameters stored in a snapshot (to size a context for it). */
int sph_snapshot_info(const char* path, uint32_t* n, sph_params* p);

/* introspection for per-phase parity tests (sorted order) */
/* [introspect] */ int sph_get_keys(sph_ctx* c, uint32_t* keys);          /* cell key per slot */
/* [introspect] */ int sph_get_order(sph_ctx* c, uint32_t* index);        /* creation index per slot */
/* [introspect] */ int sph_get_cell_range(sph_ctx* c, uint32_t cell, uint32_t* start, uint32_t* end);
/* occupied cells in ascending key order: {key, start, count}; returns the number written
 * (<= max_cells) or a negative code */
/* [introspect] */ int sph_get_cells(sph_ctx* c, uint32_t max_cells, uint32_t* key, uint32_t* start, uint32_t* count);
/* local cell key of global cell coordinates (x, y, z): (z_local*gy + y)*gx + x */
/* [introspect] */ uint32_t sph_cell_key(const sph_ctx* c, uint32_t x, uint32_t y, uint32_t z);

/* ---- phases (each replaces one seam call; see the SPH_PH_* table) ------------------------ */
int sph_hash(sph_ctx* c);
int sph_sort(sph_ctx* c);
int sph_build_cells(sph_ctx* c);
int sph_density(sph_ctx* c);
int sph_force(sph_ctx* c);
int sph_collide(sph_ctx* c);
int sph_integrate(sph_ctx* c, float dt);
/* n full time steps = n iterations of the loop body of ParticleSystem::update
 * (particleSystem.cpp:725-811), using the fused kernels (force+collision+integrate in one
 * neighbour traversal).  Results equal the phase-by-phase sequence to fp32 rounding. */
int sph_step(sph_ctx* c, float dt, uint32_t n_steps);
/* same, phase by phase (the exact call sequence of particleSystem.cpp:773-795) */
int sph_step_phased(sph_ctx* c, float dt, uint32_t n_steps);
/* the fused tail of sph_step on its own: force + collision + integrate in one neighbour traversal
 * (needs sph_density; slab drivers call it after the density halo exchange) */
int sph_force_collide_integrate(sph_ctx* c, float dt);

/* ---- device timing (replaces the host-side TIME_FUNCTION macros, particleSystem.h:20-24,
 *      which time launches, not kernels) ------------------------------------------------------ */
int sph_timing_enable(sph_ctx* c, int on);
/* sums of per-phase device milliseconds since the last reset, and the number of steps */
int sph_timing_get(sph_ctx* c, float ms[SPH_PH_COUNT], uint32_t* n_steps);
int sph_timing_reset(sph_ctx* c);
/* sort statistics: sorts run so far; how many of them took the merge path (only the particles whose
 * cell changed are sorted; same result as the full sort); how many of those found that no particle had
 * changed cell and did nothing at all (`skips`: the order, the keys and the cell table of the previous
 * step are still exact -- a fluid at rest; taken only when the device has already reported the count, the
 * host never waits for it); the mover count the device last reported; and the sum of the mover counts of
 * all sorts so far (movers_total: particles that changed cell, summed over steps).  Synchronises the stream.
 * Any pointer may be NULL. */
/* [introspect] */ int sph_sort_stats(sph_ctx* c, uint64_t* sorts, uint64_t* merges, uint64_t* skips, uint32_t* last_movers,
                   uint64_t* movers_total);
/* How the movers' sorts of the merge path were launched so far: out[0] both forms (the count lives on the device, each
 * kernel looks at it and leaves if it is not its turn), out[1] the one-block sort alone, out[2] the multi-block passes
 * alone -- the last two from the count the device last reported: the previous sort's in a host-paced (slab) context, at most
 * four sorts old in a whole-domain one (there with a margin: both forms while the count is within a quarter of the one-block
 * sort's capacity).  No synchronisation. */
/* [introspect] */ int sph_sort_forms(const sph_ctx* c, uint64_t out[3]);
/* merge = 1 (default): the sort takes the merge path while few particles change cell (up to 1/8 of them, by
 * the count the device last reported); 0: the full radix sort every step (what SPH_SORT_MERGE=0 in the
 * environment selects at sph_create time); 2: the merge path whenever the previous order is intact, whatever
 * the count (it is exact for any count; for tests).  Takes effect at the next sort. */
/* [tuning] */ int sph_set_sort_mode(sph_ctx* c, int merge);
/* The neighbour passes stage, per (dz, dy) row, the hull of a wave's candidate ranges through LDS.  A row whose hull
 * is longer than `slots` (default 512; usual hulls are ~80) is read straight from global memory by every lane instead
 * -- sparse particles next to a dense layer would otherwise stage thousands of slots for a handful of candidates each
 * and become the tail of the whole launch.  In SPH_PRECISION_F32 both ways give the same bits (same candidates, same order,
 * same arithmetic).  In SPH_PRECISION_MIXED_F16 they do NOT: the staged walk pairs fp16 candidates per 128-slot piece, the
 * direct walk per row, so the fp16 row sums differ within the mixed tolerance -- and since the way a wave goes depends on
 * its 63 wave-mates, mixed-mode results depend on this setting and on how a domain is cut into slabs.  The test in front
 * of the exact one is a heuristic (key span of the wave <= 64 cells: taken as "no hull beyond `slots`", which cells of
 * more than slots / 66 particles can break -- such a wave stages a long hull, slowly but correctly).
 * 0: every row direct; 0xFFFFFFFF: never (for tests and A/B runs).  Takes effect at the next launch. */
/* [tuning] */ int sph_set_direct_hull(sph_ctx* c, uint32_t slots);
/* A context that owns FEWER than `slots` particles launches the neighbour passes in workgroups of 128 threads instead of 256
 * (default 524288).  Waves do not depend on their block (wave-private LDS slices, no block barrier in the walk): same results
 * bit for bit; config 2's flowing dam steps 12 % faster this way, large launches lose ("small steps").
 * 0: never; 0xFFFFFFFF: always.  Takes effect at the next launch; SPH_PAIR_SMALL_SLOTS in the environment at create time does
 * the same for A/B runs. */
/* [tuning] */ int sph_set_pair_small_launch(sph_ctx* c, uint32_t slots);
/* The ORDER in which the neighbour passes' workgroups take the sorted slots (results do not depend on it).  xcd = 1
 * (default): each of the 8 XCDs walks one contiguous eighth of the slots; ztile = 1 (default): inside its eighth an XCD
 * walks strips of 2^strip_blocks_log2 workgroups (default 4: 16 workgroups = 4096 slots) through all cell layers of the
 * eighth before the next strip, so that the rows of neighbouring layers are re-used out of the XCD's L2: half the HBM-side
 * reads of the plain order.  0 / 0 restores the plain front-to-back order (A/B runs). */
/* [tuning] */ int sph_set_block_order(sph_ctx* c, int xcd, int ztile, uint32_t strip_blocks_log2);
/* 1 if the last sph_sort found that no particle had changed cell and left everything as it was (then
 * every count derived from the sorted order -- sph_slab_counts, sph_halo_count -- is that of the step
 * before), else 0.  No synchronisation. */
/* [introspect] */ int sph_last_sort_skipped(const sph_ctx* c);

/* ---- z-slab halo / migration (multi-GPU; no counterpart in the reference) ---------------- */
/* side: 0 = towards lower z (rank-1), 1 = towards higher z (rank+1).
 * Records are 8 floats: x, y, z, index bits, vx, vy, vz, 0. */
#define SPH_HALO_RECORD_FLOATS 8
/* after sph_hash + sph_sort: number of owned particles that left the slab through `side` */
int sph_migrants_count(sph_ctx* c, uint32_t count[2]);
/* everything a slab driver needs after the sort, in one device round trip:
 * {migrants down, owned in the lowest layer, owned in the highest layer, migrants up} */
int sph_slab_counts(sph_ctx* c, uint32_t count[4]);
/* pack them into buf_dev[side] (device, capacity records each) and drop them */
int sph_migrants_pack(sph_ctx* c, void* buf_dev[2], uint32_t capacity);
/* append n received particles (device records) to the owned set; call sph_hash+sph_sort again */
int sph_migrants_append(sph_ctx* c, const void* buf_dev, uint32_t n);
/* number of owned particles in the boundary layers (what the neighbours need as ghosts) */
int sph_halo_count(sph_ctx* c, uint32_t count[2]);
int sph_halo_pack(sph_ctx* c, void* buf_dev[2], uint32_t capacity);
/* same with the two counts supplied by the caller (a slab driver knows them from sph_slab_counts and
 * the migrant counts), which saves a device round trip; they are remembered for the density halo */
int sph_halo_pack_counts(sph_ctx* c, void* buf_dev[2], uint32_t capacity, const uint32_t count[2]);
/* install n_lo / n_hi ghost records received from the lower / upper neighbour */
int sph_halo_unpack(sph_ctx* c, const void* lo_dev, uint32_t n_lo, const void* hi_dev, uint32_t n_hi);
/* second exchange: (density, pressure) float2 of the same boundary particles, same order */
int sph_halo_pack_density(sph_ctx* c, void* buf_dev[2], uint32_t capacity);
int sph_halo_unpack_density(sph_ctx* c, const void* lo_dev, const void* hi_dev);
/* per z cell layer histogram of owned particles (global layer ids), for count-balanced cuts */
int sph_layer_histogram(sph_ctx* c, uint32_t* hist, uint32_t n_layers);

/* ---- the slab step under the C ABI: one call = one time step of this rank's slab, with its neighbours ------- */
/* How the messages of a step reach the two z-neighbours.  `exchange` moves four buffers at once: send_lo goes to
 * rank-1 (and arrives there in its recv_hi), send_hi to rank+1, recv_lo / recv_hi are filled by the neighbours;
 * a size of 0 means "nothing on that side" (domain ends, empty layers) -- both ends of a link compute the same sizes.
 *   host_buffers = 0 (product): the pointers are DEVICE pointers and the call enqueues the transfers on hip_stream
 *                    without blocking the host (RCCL: ncclSend/ncclRecv inside one group);
 *   host_buffers = 1 (tests): the pointers are pinned HOST buffers the library staged, the call blocks until its
 *                    receives are complete (several slabs of one GPU in one process; processes over gloo).
 * Return 0 or a negative SPH_E* code. */
enum { SPH_TAG_MIGRANTS = 1, SPH_TAG_HALO_A = 2, SPH_TAG_HALO_B = 3, SPH_TAG_MIGRANTS_REST = 4, SPH_TAG_PING = 5,
       SPH_TAG_RECUT_COUNTS = 6, SPH_TAG_RECUT = 7,
       SPH_TAG_ONE = 8,        /* the one-message step: header + leavers + two layers of residents, size fixed in advance */
       SPH_TAG_ONE_REST = 9 }; /* ... and what did not fit that size (exact; a burst) */
/* ZERO-INITIALISE the struct before filling it in (`sph_transport t = {0};`): sph_slab_create copies it by value, and
 * members added at its end (so far: `abort`, ABI v2 of round 4) must read as NULL in a caller built against an older
 * header -- there is no size field, a garbage `abort` pointer would be called on the first failure. */
typedef struct sph_transport {
    void* self;
    int (*exchange)(void* self, int tag, const void* send_lo, size_t send_lo_bytes, void* recv_lo, size_t recv_lo_bytes,
                    const void* send_hi, size_t send_hi_bytes, void* recv_hi, size_t recv_hi_bytes, void* hip_stream);
    int host_buffers;
    /* optional (may be NULL): take down everything the transport still has queued on a stream -- called when an exchange
     * failed or a neighbour stopped answering, so that no receive without a sender blocks a later synchronisation */
    void (*abort)(void* self);
} sph_transport;

/* RCCL transport over the direct xGMI links (librccl is loaded on first use): rank 0 makes the id, every rank gets
 * the same 128 bytes (the launcher broadcasts them) and joins the communicator with its rank. */
int sph_rccl_unique_id(uint8_t id[128]);
int sph_rccl_transport_create(sph_transport** out, const uint8_t id[128], int rank, int world, int device);
void sph_rccl_transport_destroy(sph_transport* t);
/* two messages of `bytes` and `bytes`/2 bytes from this rank to ITSELF through the transport's communicator (one
 * ncclGroup), compared on the host: checks the dlopen binding of librccl with real traffic on a one-GPU box */
/* [test hook] */ int sph_rccl_transport_selftest(sph_transport* t, size_t bytes);
/* What the communicator says about itself: {ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclCommGetAsyncError}
 * (-1 where librccl lacks the call).  The multi-GPU bench prints them: a communicator of the wrong size, or one that sits on
 * another device than the slab's context, explains a hang or a slow run before any step is taken. */
/* [introspect] */ int sph_rccl_transport_info(const sph_transport* t, int out[4]);

/* Device-to-device transport between the slabs of ONE process that share a GPU (one thread per rank): the buffers are
 * device pointers, the copies are queued on the caller's comm stream behind the sender's event, nothing blocks on the
 * device -- the same stream/event edges as with RCCL, on a one-GPU box.  The host threads rendezvous with bounded
 * waits; message sizes and tags of both ends are compared (a disagreement is SPH_E_STATE, not a hang).  One hub per
 * group of ranks, one transport per rank; destroy the transports, then the hub. */
typedef struct sph_local_hub sph_local_hub;
int sph_local_hub_create(sph_local_hub** out, int world, int device);
void sph_local_hub_destroy(sph_local_hub* hub);
int sph_local_hub_set_timeout(sph_local_hub* hub, double seconds);
int sph_local_transport_create(sph_transport** out, sph_local_hub* hub, int rank);
void sph_local_transport_destroy(sph_transport* t);

/* Loop transport: ONE slab whose two neighbours are its own periodic images along z -- what it sends up arrives from below
 * with every position shifted down by `z_shift` (the height of the slab's owned layers), and the other way round.  The slab is
 * created as the middle rank of three (sph_slab_create(..., rank 1, world 3, ...)) on cell layers away from the box's z faces
 * and then does ALL the work of a rank between two neighbours -- migrants, both halo messages at their real sizes, ghost
 * unpack, boundary launches, the host wait -- on one device, with nothing else sharing it: the measured stand-in for a middle
 * rank of an N-GPU run that a one-GPU box allows (`bench.py --force-slab --periodic-z`).  Every message is held back by
 * latency_us + bytes / link_gbs (0 / 0: no delay) on the comm stream before it is delivered: the link is a parameter, not a
 * measurement.  sph_slab_ping does not apply (a slab cannot tell its images apart). */
/* [test hook] */ int sph_loop_transport_create(sph_transport** out, float z_shift, double link_gbs, double latency_us);
/* [test hook] */ void sph_loop_transport_destroy(sph_transport* t);

typedef struct sph_slab sph_slab;
/* Bind a slab context (sph_create_slab, particles uploaded) to its place in the chain of `world` slabs.  The halo
 * capacity is the context's ghost capacity; migrant_capacity (records per side and step, 0 = half the ghost capacity)
 * bounds the leavers / arrivals of one step and side (only 255 of them ride in the fixed-size message of every step).  The transport struct is copied.
 * The step orders its streams by sequence numbers (hipStreamWriteValue32 / hipStreamWaitValue32 on a word of device memory) where the
 * device serves that pair -- tried once here, with both streams drained -- and by events otherwise; SPH_SLAB_HOPS=event in the
 * environment forces events (A/B runs, kernel traces: the runtime's wait is a spinning one-workgroup kernel). */
int sph_slab_create(sph_slab** out, sph_ctx* ctx, int rank, int world, const sph_transport* transport,
                    uint32_t migrant_capacity);
void sph_slab_destroy(sph_slab* s);
/* n time steps: sort, migrants, halo A, density, halo B, force + collision + integrate (csrc/sph_slab.hip), queued
 * on the context's stream and a second, high-priority stream; the host waits ONCE per step (for the layer counts),
 * behind the density pass of the slab's deep interior, and that wait is bounded (below).  A particle that crosses
 * more than one cell layer in a step ("far") is handled (one more wait on that step) as long as it lands in an
 * interior layer of the receiving slab; anything else is reported as SPH_E_STATE, never merged out of order. */
int sph_slab_step(sph_slab* s, float dt, uint32_t n_steps);
/* drains both streams; also reports what device-side checks of the last steps flagged */
int sph_slab_sync(sph_slab* s);
/* the step's wait gives up after `seconds` (default 120, or SPH_SLAB_TIMEOUT_S) with SPH_E_DEVICE: a neighbour that
 * stopped with an error never sends its messages */
int sph_slab_set_wait_timeout(sph_slab* s, double seconds);
/* {steps, particles sent away, steps with arrivals, ghosts received, host waits} */
/* [introspect] */ int sph_slab_stats(const sph_slab* s, uint64_t out[5]);
/* the same five + {in-place merges, steps with far arrivals, steps that needed a second migrant message} */
/* [introspect] */ int sph_slab_counters(const sph_slab* s, uint64_t out[8]);
/* of the steps with arrivals, those that merged them into the two boundary layers in place (the rest ran a pass over
 * all particles: more than 2048 arrivals on a side, or no valid cell table) */
/* [introspect] */ uint64_t sph_slab_in_place_merges(const sph_slab* s);
/* transport calls so far: 3 in a usual step (migrants, halo A, halo B), 4 when a side has more leavers than ride in the
 * fixed-size migrant message */
/* [introspect] */ uint64_t sph_slab_exchanges(const sph_slab* s);
/* Re-cut (re-balancing): this slab takes over the cell layers [new_z_lo, new_z_hi) of the global grid.  Entirely on the
 * device, the context and the slab object are kept: the owned particles are split by their layer into "to rank - 1" |
 * "stay" | "to rank + 1" (stable), the leaving runs travel point to point through the slab's transport in chunks of the
 * halo buffers, arrivals from below are put in front of what stays and arrivals from above behind it -- the order of the
 * whole-domain sorted array, so an N-slab run keeps the bits of the one-context run across re-cuts -- and the next step
 * starts with a full stable sort, as after an upload.  COLLECTIVE: every rank of the chain calls it (a rank whose layers do
 * not change passes its old range), with cuts that agree across ranks; a particle moves at most ONE rank per call, i.e.
 * new cut r lies within [old cut r-1, old cut r+1] (gpufluidsimulator_amd/slab.py: single_hop_cuts steps a larger move).
 * SPH_E_CAPACITY if the new layers hold more particles than the context's capacity (nothing is lost: the slab is failed,
 * download and destroy).  Replaces nothing in the reference (single GPU); SURVEY.md section 8e "re-cut every K steps". */
int sph_slab_recut(sph_slab* s, uint32_t new_z_lo, uint32_t new_z_hi);
/* {re-cuts so far, particles that changed owner in them} */
/* [introspect] */ int sph_slab_recut_stats(const sph_slab* s, uint64_t out[2]);
/* Neighbour ping: `reps` timed rounds (+ one untimed round first) of ONE exchange-shaped group -- `bytes` to and from
 * rank - 1 and rank + 1, through this slab's transport, comm stream and halo buffers -- every word checked on arrival for
 * sender, direction and round.  out = {mean us per group, max us, wrong words}; event times on the comm stream, i.e. what a
 * step's group of that size costs the device (waiting for the neighbour included).  bytes: a multiple of 4, at most the halo
 * buffer ((ghost capacity + 1) * 32).  COLLECTIVE over the chain.  A neighbour that does not answer within the wait
 * time-out, an RCCL error or a wrong word is an error (the slab is failed); there is no fall-back to another transport. */
int sph_slab_ping(sph_slab* s, size_t bytes, uint32_t reps, double out[3]);
/* Where a step's time goes.  Host side, always measured: the step's one wait (WAIT), the host time in front of it (PRE:
 * hash, sort, bounds kernel and the migrant exchange being queued), behind it (POST: everything else being queued) and the
 * whole call (HOST), each {sum us, max us} over the steps since the last reset; WAITS_READY counts the waits that found the
 * header already there -- the host, not the device, paced those steps.  Device side, only between sph_slab_timing_enable(1)
 * and (0): an event pair around every transport call on the comm stream, per message group {calls, sum us, max us} -- the
 * time the group occupies the comm stream, the neighbour's lateness included.  (An event recorded on a stream costs the
 * device ~5 us at its next dispatch: switch it on for a probe pass, not for a timed one.)  _get drains the comm stream. */
enum {
    SPH_SLAB_T_STEPS = 0, SPH_SLAB_T_WAITS_READY = 1,
    SPH_SLAB_T_WAIT = 2,   /* [2] sum, [3] max */
    SPH_SLAB_T_PRE = 4, SPH_SLAB_T_POST = 6, SPH_SLAB_T_HOST = 8,
    SPH_SLAB_T_GROUPS = 10, /* + 3 * (tag - 1): {calls, sum us, max us} for SPH_TAG_MIGRANTS, _HALO_A, _HALO_B, _MIGRANTS_REST */
    SPH_SLAB_T_GROUPS_ONE = 22, /* {calls, sum us, max us} for SPH_TAG_ONE, then for SPH_TAG_ONE_REST (the one-message protocol) */
    SPH_SLAB_T_WORDS = 28
};
int sph_slab_timing_enable(sph_slab* s, int on);
int sph_slab_timing_reset(sph_slab* s);
int sph_slab_timing_get(sph_slab* s, double out[SPH_SLAB_T_WORDS]);
/* on = 1 (default): the fused force pass of the slab's innermost layers (at least six cell layers from either cut; slabs of
 * 11 owned layers and more) is queued IN FRONT of the step's host wait, behind the deep density, on a stream of its own: work
 * that needs nothing from a neighbour keeps the device busy while the migrant message and halo A are on their way, and its
 * tail runs beside the interior launch.  The step's time becomes nearly independent of the links' latency (DESIGN.md
 * section 6, measured on a slab between its periodic images: +3 us per step with fast links, -10 at 20 us per message
 * group, -30 at 40).  For ONE rank per device: ranks that share a GPU (rehearsals over the local / host transports) should
 * switch it off -- their big kernels then run beside each other all the time and evict each other's L2 working sets (two
 * 8.4 M-particle slabs on one GPU: 4.81 against 3.86 ms per step); the launchers do.  The launch covers at most 2^20 slots
 * of those layers (~150 us of k_force: what a link's latency needs; a longer one runs beside the interior launch for its whole
 * length and the two evict each other's L2 working sets -- a 16.7 M-particle slab: 3.90 against 3.69 ms); on > 1 sets that
 * number of slots (environment: SPH_SLAB_EARLY_SPAN).  A slab without neighbours (world 1) never launches it.  Same bits either way.
 * out = {steps that launched it, steps that used its result} (a step whose arrivals re-sort the slab discards it). */
/* [tuning] */ int sph_slab_set_early_force(sph_slab* s, int on);
/* The message protocol of a step.  3 (default): MIGRANTS (header + leavers) -> the host's wait -> HALO A (boundary layers) ->
 * HALO B (their densities), three dependent groups per step.  1: ONE group -- header, leavers and the RESIDENTS of the two
 * cell layers next to each cut in one message per neighbour ("a 2-layer halo, ghost densities recomputed
 * locally (one message)"): the receiver merges its own leavers into the copy (the order the neighbour will give them), computes
 * the densities of the inner ghost layer itself -- same candidates, same order, same bits in fp32 -- and the force pass needs no
 * message.  The message's size is fixed in advance by a rule on the counts both ends saw in the previous step's headers (margin
 * 1/16 + 1024 records); what does not fit follows in an exact second message (out[2] counts those); the first step after
 * sph_slab_create / sph_slab_recut / this call runs the three-group protocol to learn the counts.  Needs a context with two
 * ghost layers (sph_create_slab_layers) and slabs of >= 4 cell layers; a particle that crosses more than TWO layers in a step is
 * SPH_E_STATE under it.  COLLECTIVE in effect: every rank of the chain must run the same protocol.  Price: ~2x the halo bytes and
 * the density of one more layer per side; gain: two message latencies off the step's critical path.
 * sph_slab_protocol: out = {protocol, one-message steps so far, of which needed the second message}. */
int sph_slab_set_protocol(sph_slab* s, int groups);
/* [introspect] */ int sph_slab_protocol(const sph_slab* s, uint64_t out[3]);
/* [introspect] */ int sph_slab_early_force_stats(const sph_slab* s, uint64_t out[2]);
/* TEST HOOK: sph_upload / sph_set_by_index / sph_reset_lattice / sph_set_params make the next five movers' sorts of a
 * whole-domain context launch BOTH forms (the caller may have changed every particle: the count the device last reported says
 * nothing).  This takes that back, so that a test can put a changed state in front of a sort launched on the old count. */
/* [test hook] */ int sph_test_trust_mover_hint(sph_ctx* c);
/* TEST HOOK: raise sticky device-side error word `flag` (0: an arrival outside its boundary layer, 1: an arrival outside
 * the slab) as the insert / unpack kernels would; the next sph_slab_step then fails before it has sent anything. */
/* [test hook] */ int sph_slab_test_raise_flag(sph_slab* s, int flag);
/* 0, or the first error of this slab: a slab that failed stays failed -- it has told its neighbours (they return
 * SPH_E_PEER at their next step), every later sph_slab_step returns the same error, and the state of its context is that
 * of a half-done step: download / destroy only */
int sph_slab_failed(const sph_slab* s);

#ifdef __cplusplus
}
#endif
#endif /* SPH_HIP_H */
