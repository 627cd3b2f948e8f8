/* perf_hip.h -- C ABI of libperf_hip.so: the MI355X (gfx950) panoramic-NeRF hot path.
 *
 * This is the drop-in boundary for PeRF's modules/fields + modules/scene hot path.
 * Every entry point replaces an operator PeRF reaches through a third-party CUDA
 * extension (tinycudann / nerfacc / torch_efficient_distloss) or through a chain of
 * torch ops; the replaced call site is cited per function (paths relative to the
 * reference tree).  INTEGRATION.md shows the ctypes binding and the Python shim
 * packages (`tinycudann`, `nerfacc`, `torch_efficient_distloss`) that sit on top.
 *
 * Contract
 *  - plain C: raw device pointers, element counts, POD descriptors, a hipStream_t
 *    passed as void*.  No torch types, no C++ exceptions cross the boundary.
 *  - every function returns 0 on success or a negative PERF_E_* code; the message is
 *    available from perf_last_error() (thread local).
 *  - the library never allocates or frees device memory: outputs and workspaces are
 *    caller-owned (PyTorch caching allocator).  Variable-size outputs use the
 *    count -> scan -> (caller allocates) -> write protocol.
 *  - all work is enqueued on the given stream; no implicit device synchronisation;
 *    fixed-shape call sequences are hipGraph-capturable.
 *  - "16-bit" buffers hold bf16 (PERF_DTYPE_BF16) or IEEE fp16 (PERF_DTYPE_FP16).
 *  - device-side counts: per-sample entry points take `n` = the CAPACITY of their arrays (also the stride of
 *    level-major buffers) and `const int64_t* n_dev` (device memory, may be NULL): when given, only the first
 *    min(n, *n_dev) samples are processed -- the count a preceding perf_exclusive_scan_i32 left on the device.
 *    A batch whose sample count is only known on the GPU (occupancy marching, early termination) is thus a fixed
 *    sequence of launches with no host read-back: hipGraph-capturable (BASELINE config 4, and the training step).
 *  - the library is re-entrant: no mutable global state besides std::call_once-guarded kernel attribute setup and
 *    environment switches read once at first use.
 */
#ifndef PERF_HIP_H
#define PERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PERF_ABI_VERSION 13

#define PERF_OK 0
#define PERF_E_INVALID (-1)   /* bad argument */
#define PERF_E_LAUNCH (-2)    /* HIP launch / runtime error */
#define PERF_E_UNSUPPORTED (-3)

#define PERF_DTYPE_BF16 0
#define PERF_DTYPE_FP16 1

#define PERF_ACT_NONE 0
#define PERF_ACT_SIGMOID 1
#define PERF_ACT_EXP 2        /* y -> exp(y - exp_shift); backward clamps the exponent at 15 (trunc_exp) */

/* Marching lattice t_k of a ray (all four perf_occ_march_* entry points take one): REPEATED = t_0 = t0, t_{k+1} = fl(t_k + step),
 * the lattice of a marcher that advances by `t += dt` as nerfacc's traverse_grids is understood to (the package is absent:
 * unpinned either way) -- the default of the host layers and of oracle/perf_oracle.py since round 4; SINGLE = fl(t0 +
 * fl(k*step)), one rounding per sample (rounds 1-3).  The two differ by O(k ulp) in t_starts / t_ends, and in ray_indices
 * where a midpoint sits on a cell boundary.  A maintainer who holds nerfacc checks the choice with tools/pin_upstream.py
 * (NeRFOCCRenderer.lattice selects). */
#define PERF_LATTICE_SINGLE 0
#define PERF_LATTICE_REPEATED 1

#define PERF_INTERP_LINEAR 0
#define PERF_INTERP_SMOOTHSTEP 1

#define PERF_MAX_LEVELS 24

/* Table layout of a grid.  TCNN: tiny-cuda-nn's -- what every grid of the reference uses (modules/fields/ngp_nerf.py:96-134), the
 * only layout the gradient / second-order / fused entry points accept.  LINE_LOCAL: opt-in, inference only (perf_hashgrid_fwd,
 * perf_hashgrid_corners, perf_field_infer), for grids the reference never defines (BASELINE.json configs[4]: L = 20 tables sized to
 * HBM): a level with local[l] != 0 stores the vertices of a 4 x 4 x 2 block as one 128-byte line (entry x%4 + 4 (y%4) + 16 (z%2)),
 * the blocks of a super-block of 2^sb_shift[0] x 2^sb_shift[1] x 2^sb_shift[2] vertices contiguously (x-major), and addresses the
 * SUPER-BLOCK densely (sx + sy*nsx[l] + sz*nsxy[l], hashed[l] == 0) or by the prime-XOR hash of its coordinates modulo
 * size[l] >> (sb_shift[0] + sb_shift[1] + sb_shift[2]).  offset[l] of such a level is a multiple of 32 entries (its blocks ARE cache
 * lines; perf_amd.grid.GridConfig starts it on a super-block boundary: entries in front of it are padding) and the table pointer is
 * 128-byte aligned.  Levels with local[l] == 0 keep the TCNN rule.
 * LINE_OVERLAP (ABI 11): LINE_LOCAL whose 16-byte x runs overlap by one vertex, so that the x corner pair of a cell is ONE request:
 * the pair of CELL gx lives in run gx / 3 of its row at positions gx % 3 and gx % 3 + 1 -- the LINE_LOCAL rule applied to the storage
 * coordinate X = gx + gx / 3 (first corner) and X + 1 (second corner).  Position 3 of a run repeats position 0 of the next run of
 * the same super-block row: the owner of the table keeps the two entries equal (perf_amd.grid.GridConfig.canonicalize_).  The LAST
 * cell of a super-block row (X % 2^sb_shift[0] == 2^sb_shift[0] - 2) takes its second corner from storage coordinate X + 2 -- the
 * first vertex of the next super-block -- so no vertex is stored in two super-blocks (hashed super-blocks could not keep such
 * copies equal).  A super-block row holds 3 * 2^(sb_shift[0] - 2) cells: nsx[l] counts such rows; a hashed level's 2^T entries hold
 * 3/4 as many distinct vertices.  A cell's eight corners lie in (1 + 1/4)(1 + 1/2) = 1.9 lines instead of 2.3. */
#define PERF_LAYOUT_TCNN 0
#define PERF_LAYOUT_LINE_LOCAL 1
#define PERF_LAYOUT_LINE_OVERLAP 2

/* Geometry of a multiresolution hash grid (tcnn "HashGrid", 3 input dims, 2 features/level).
 * Entry e of level l lives at table[(offset[l] + e) * 2 + f].  Levels with hashed[l]==0 are
 * dense (index x + y*res + z*res^2), the others use the prime-XOR hash; both modulo size[l]. */
typedef struct perf_grid_desc {
    int32_t n_levels;              /* <= PERF_MAX_LEVELS */
    int32_t interpolation;         /* PERF_INTERP_* */
    float scale[PERF_MAX_LEVELS];
    uint32_t res[PERF_MAX_LEVELS];
    uint32_t size[PERF_MAX_LEVELS];
    uint64_t offset[PERF_MAX_LEVELS];   /* 64-bit: tables beyond 2^32 entries (BASELINE config 5) */
    uint32_t hashed[PERF_MAX_LEVELS];
    int32_t layout;                /* PERF_LAYOUT_* */
    uint32_t sb_shift[3];          /* LINE_LOCAL / LINE_OVERLAP: log2 vertices (x: storage positions) of a super-block along x, y (>= 2) and z (>= 1) */
    uint32_t local[PERF_MAX_LEVELS];    /* LINE_LOCAL / LINE_OVERLAP: level stored line-local */
    uint32_t nsx[PERF_MAX_LEVELS];      /* dense line-local levels: super-blocks per row ... */
    uint32_t nsxy[PERF_MAX_LEVELS];     /* ... and per z-slice */
} perf_grid_desc;

/* Bias-free 64-wide MLP (tcnn "FullyFusedMLP"): n_levels*2 inputs (zero padded to a multiple
 * of 16), n_hidden_layers in {1,2} ReLU layers of 64, output padded to 16 rows.  Weights are
 * row-major [out,in] 16-bit matrices, concatenated: [64 x n_in_pad][64 x 64]*(h-1)[16 x 64]. */
typedef struct perf_mlp_desc {
    int32_t n_levels;              /* inputs = 2*n_levels */
    int32_t n_hidden_layers;       /* 1 or 2 */
    int32_t n_out;                 /* 1..16 */
    int32_t out_act;               /* PERF_ACT_* */
    float exp_shift;               /* PERF_ACT_EXP only */
} perf_mlp_desc;

int perf_version(void);                 /* == PERF_ABI_VERSION of the header the library was built from */
const char* perf_last_error(void);
/* sizeof(perf_grid_desc) / sizeof(perf_mlp_desc) as compiled: a binding checks its own struct layout against these. */
int64_t perf_sizeof_grid_desc(void);
int64_t perf_sizeof_mlp_desc(void);

/* ---- parameters ------------------------------------------------------------------------ */

/* fp32 master -> 16-bit working copy.  Replaces the per-call half cast inside tcnn's torch
 * binding (reached from modules/fields/ngp_nerf.py:142,158). */
int perf_cast_params(const float* src, void* dst16, int64_t n, int dtype, void* stream);

/* One Adam step (torch.optim.Adam semantics, modules/scene/nerf.py:171,180,253,293) fused with
 * the refresh of the 16-bit working copy and the zeroing of the gradient.
 * p,m,v,g fp32 [n]; step >= 1; w16 may be NULL; zero_grad != 0 clears g. */
int perf_adam_step(float* p, float* m, float* v, float* g, void* w16, int64_t n, int dtype,
                   int32_t step, float lr, float beta1, float beta2, float eps, int zero_grad,
                   void* stream);

/* Same with the step count (int32, >= 1) and the learning rate read from DEVICE memory, so that a captured
 * hipGraph of the training step can be replayed while the schedule advances.  gate_dev (device int64, may be NULL):
 * the whole update is skipped when *gate_dev <= 0 -- the reference skips the step of a batch without samples
 * (modules/scene/nerf.py:204-206,277-279); the caller advances *step_dev accordingly. */
int perf_adam_step_dev(float* p, float* m, float* v, float* g, void* w16, int64_t n, int dtype,
                       const int32_t* step_dev, const float* lr_dev, const int64_t* gate_dev, float beta1, float beta2,
                       float eps, int zero_grad, int32_t* clear_flag, void* stream);
/* clear_flag (ABI 13, device int32, may be NULL): *clear_flag = 0 is written by this launch whether or not the update is taken --
 * how the fixed-point overflow flag is consumed when the step's bookkeeping rode in the repair launch (perf_field_bwd_book), whose
 * workgroups all read the flag and can therefore not clear it themselves. */

/* Device-side bookkeeping of one sync-free training step, one tiny launch, issued between the backward and
 * perf_adam_step_dev.  The step is TAKEN when the batch has samples (*gate_dev > 0 or gate_dev == NULL; the reference skips
 * batches without samples, nerf.py:204-206), the fixed-point grid backward did not flag an overflow (*overflow_flag != 0,
 * or remote_flags[0] > 0: the sum of ALL ranks' flags under data parallelism) and the batch was not truncated
 * (*n_marched_dev > capacity > 0: late rays lost their samples; or remote_flags[1] > 0: some rank's batch was).  remote_flags
 * (device, TWO floats, may be NULL) is what perf_dp_slot_unpack leaves: with it every rank of a job decides alike.  *step_dev += 1 and *eff_gate_out = 1 when taken,
 * *eff_gate_out = 0 otherwise -- pass eff_gate_out to perf_adam_step_dev as its gate: a corrupted or truncated gradient is
 * never applied.  overflow_redone != 0: the caller has REPAIRED a flagged gradient in place (perf_hashgrid_bwd's redo launch,
 * fp32 accumulation) -- the flag no longer gates, the event is still counted: the single-GPU paths never drop a step for it.
 * *overflow_flag is cleared (the event is counted instead).  counters (int64 [PERF_STEP_COUNTERS], may be
 * NULL) accumulate {marched samples, kept samples, steps, largest marched count of one batch, steps skipped for overflow,
 * steps skipped for truncation, 0, 0}: throughput and health accounting never read the device inside the training loop.
 * schedule (device, n_schedule rows of {learning rate, distortion-loss ramp min(2 progress, 1)}; NULL = none) with iter_dev
 * (device int32, the iteration counter, advanced by this launch): *lr_out = the current iteration's learning rate (what
 * perf_adam_step_dev reads next), *ratio_out = the NEXT iteration's ramp (what its perf_geo_loss reads, nerf.py:235) -- the
 * reference's update_lr (nerf.py:300-311) evaluated ahead of time, so that replaying a captured step needs no host-side scalar
 * update.  All pointers device memory; every pointer may be NULL. */
#define PERF_STEP_COUNTERS 8
int perf_step_bookkeeping(int32_t* step_dev, const int64_t* gate_dev, int64_t* counters,
                          const int64_t* n_marched_dev, const int64_t* n_kept_dev, int64_t capacity,
                          int32_t* overflow_flag, const float* remote_flags, int32_t overflow_redone, int64_t* eff_gate_out,
                          const float* schedule, int32_t n_schedule, int32_t* iter_dev, float* lr_out, float* ratio_out,
                          void* stream);

/* The arguments of perf_step_bookkeeping as a POD block (ABI 13), for calls that carry the bookkeeping in one of their own launches
 * (perf_field_bwd_book).  Same meaning field by field; every pointer device memory, every pointer may be NULL. */
typedef struct perf_step_book {
    int32_t* step_dev;
    const int64_t* gate_dev;
    int64_t* counters;
    const int64_t* n_marched_dev;
    const int64_t* n_kept_dev;
    int64_t capacity;
    int32_t* overflow_flag;
    const float* remote_flags;
    int64_t* eff_gate_out;
    const float* schedule;
    int32_t* iter_dev;
    float* lr_out;
    float* ratio_out;
    int32_t n_schedule;
    int32_t overflow_redone;
} perf_step_book;
int64_t perf_sizeof_step_book(void);

/* ---- sample positions ------------------------------------------------------------------ */

/* x = o[ray] + d[ray]*(t0+t1)/2 (modules/scene/nerf_renderer.py:125-127), then
 * x01 = (x-aabb_min)/(aabb_max-aabb_min) and sel = all(0<x01<1) (modules/fields/ngp_nerf.py:137-140).
 * aabb: 6 host floats.  x01 [n,3], sel [n] (uint8).  ray_indices int64 [n]. */
int perf_points_from_rays(const float* rays_o, const float* rays_d, const int64_t* ray_indices,
                          const float* t_starts, const float* t_ends, const float* aabb,
                          float* x01, uint8_t* sel, int64_t n, const int64_t* n_dev, void* stream);

/* Same normalisation for explicit points x [n,3] (NGPNeRF.query_density/query_rgb called on points). */
int perf_points_normalize(const float* x, const float* aabb, float* x01, uint8_t* sel, int64_t n,
                          void* stream);

/* ---- multiresolution hash grid ----------------------------------------------------------- */

/* tcnn kernel_grid: x01 [n,3] -> features, LEVEL-MAJOR: feat[(l*n + i)*2 + f], 16-bit.
 * Replaces the encoding half of tcnn.NetworkWithInputEncoding.forward (ngp_nerf.py:142,158,258).
 * 15/16-level grids: level groups are dealt to the XCDs in eight rotating phases (every L2 holds two tables at a time and
 * every XCD serves every group for an eighth of the samples), and neighbouring lanes whose samples fall into the same cell
 * of a level share ONE gather (run de-duplication) -- same features, bit for bit, as the plain kernel. */
int perf_hashgrid_fwd(const perf_grid_desc* grid, const float* x01, const void* table16,
                      void* feat16, int64_t n, const int64_t* n_dev, int dtype, void* stream);

/* Two tables of identical geometry (PeRF's density and colour grids) at the same points in one pass:
 * corner indices/weights are shared.  Same layouts as perf_hashgrid_fwd. */
int perf_hashgrid_fwd2(const perf_grid_desc* grid, const float* x01, const void* table16_a,
                       const void* table16_b, void* feat16_a, void* feat16_b, int64_t n, int dtype,
                       void* stream);

/* Same with fp32 table and fp32 output feat[(l*n+i)*2+f] (tcnn.Encoding as used by
 * modules/geo_predictors/pano_joint_predictor.py:30-41 keeps full precision available). */
int perf_hashgrid_fwd_f32(const perf_grid_desc* grid, const float* x01, const float* table,
                          float* feat, int64_t n, void* stream);

/* tcnn kernel_grid_backward: reduce dfeat (fp32, level-major like feat) into grad_table
 * (fp32 [total*2]).  Every entry of grad_table is written exactly once: overwritten when
 * accumulate == 0 (no zero-fill needed), added to when accumulate != 0.  No global atomics.
 * level_absmax == NULL: fp32 LDS accumulation.  level_absmax != NULL (device, PERF_MAX_LEVELS floats, the
 * per-level max |dfeat| as produced by perf_mlp_bwd): packed fixed-point accumulation with full-rate integer
 * LDS atomics, unit_l = 2^ceil(log2 absmax_l) * 2^(h_l - 31) with the headroom h_l = clamp(ceil(log2(8 n / size_l)) + 6,
 * 12, 24) (64x the average number of contributions an entry of the level sums); *overflow_flag (device int32, may be NULL) is OR-ed with 1
 * when any field comes within 4x of the int32 range (then repeat the call with level_absmax == NULL).
 * workspace: 16-byte aligned device scratch of perf_hashgrid_bwd_workspace_bytes(grid, n) bytes (replica slabs of
 * the coarse levels + 4 bytes per (sample, hashed level) of tile codes + tiles x n / 8 bytes of per-tile bitmaps for levels of
 * 256-2048 hashed / 32-2048 dense tiles; a workspace without room for the codes is accepted and selects the slower
 * position-streaming owners, one without room for the bitmaps the global-atomics scatter for those levels).  With n_dev the headroom follows the live count.
 * headroom_state (device, PERF_HEADROOM_STATE_WORDS int32, zero-initialised by the caller and then owned by the sequence
 * of calls on one table, may be NULL): closes the loop on the headroom -- every call records the largest field each level's
 * FINAL sums reached and the next call's h_l is corrected to keep it between 2^21 and 2^25 units (16x below the level that raises the overflow flag) (entries next to a
 * panorama's common ray origin collect 30x the average number of contributions; hashed levels far fewer than the static
 * guess allows).  With the state the first call starts 3 bits on the safe side of the static h_l and h_l ranges over [4, 28].
 * Integer sums are exact and order independent: replicated (coarse) levels add their replicas as integers, so the table
 * does not depend on how the samples were dealt to workgroups.
 * Data-parallel training extends that to ranks:
 *   shifts_dev (device, PERF_MAX_LEVELS int32, may be NULL): the per-level units (one unit = 2^-shift) are GIVEN -- the
 *     job-wide ones of perf_dp_units -- instead of being derived from level_absmax / n / headroom_state (which must be
 *     NULL then; level_absmax may be NULL too);
 *   raw_fields != 0: grad_table receives the int32 field pairs {feature 0, feature 1} of every entry instead of floats
 *     (same addresses, reinterpret as int32).  The ranks' tables are then summed exactly by an integer reduce-scatter and
 *     converted by perf_fixed_unfix: the result equals the single-process table bit for bit.  Needs accumulate == 0 and no
 *     level on the global-atomics scatter (levels beyond 2048 tiles of 16,384 entries, or beyond 255 hashed / 64 dense tiles
 *     when the workspace cannot hold their per-tile bitmaps: perf_hashgrid_bwd_workspace_bytes includes them up to 2 GiB).
 * redo_flag (device int32, may be NULL): the call is the REPAIR of the fixed-point call issued just before it with the same
 *   x01 / dfeat / grad_table / n / n_dev: ONE launch that does nothing unless *redo_flag != 0 (pass that call's
 *   overflow_flag) and otherwise rewrites the whole gradient table with fp32 LDS accumulation (single owners streaming
 *   positions: slow, and needed a handful of times per million steps).  Fixed launch sequence, so a captured hipGraph of the
 *   training step never has to drop a step for an overflow.  Requires level_absmax == shifts_dev == NULL, accumulate ==
 *   raw_fields == 0, no workspace; PERF_E_UNSUPPORTED for grids with levels beyond LDS owners.  headroom_state (may be NULL):
 *   word [2 * PERF_MAX_LEVELS + 1] counts the repairs that ran. */
#define PERF_HEADROOM_STATE_WORDS (2 * PERF_MAX_LEVELS + 8)
int64_t perf_hashgrid_bwd_workspace_bytes(const perf_grid_desc* grid, int64_t n);
int perf_hashgrid_bwd(const perf_grid_desc* grid, const float* x01, const float* dfeat,
                      float* grad_table, int64_t n, const int64_t* n_dev, int accumulate, const float* level_absmax,
                      int32_t* overflow_flag, int32_t* headroom_state, const int32_t* shifts_dev, int raw_fields,
                      const int32_t* redo_flag, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- job-wide fixed-point units of a data-parallel step (no counterpart in the reference, which is single-GPU;
 *      rays shard over the GPUs, one gradient exchange per step) ------------------------------------------
 * Between the MLP backward and the grid backward every rank packs PERF_DP_STATS int32 words -- the float bits of its
 * level_absmax [PERF_MAX_LEVELS], the largest field per level of ITS slice of the previous step's summed table
 * (field_max_prev, from perf_fixed_unfix; NULL = no previous step), its live sample count min(n, *n_dev) as two words --
 * the blocks are all-gathered (bitwise), and perf_dp_units derives on every rank the SAME units the single process would
 * use: max of the absmax, sum of the counts, headroom feedback with the max of the field maxima (applied to
 * headroom_state first, exactly where the single process applies it: between two calls).  shifts_out: PERF_MAX_LEVELS
 * int32 for perf_hashgrid_bwd(shifts_dev) / perf_fixed_unfix; n_total_out (int64, may be NULL): the job's sample count
 * (the gate of the optimizer step).  margin_bits (0..8): make the units that many bits coarser (see perf_dp_slot_pack). */
#define PERF_DP_STATS 64
int perf_dp_stats_pack(const float* level_absmax, const int32_t* field_max_prev, const int64_t* n_dev, int64_t n,
                       int32_t* stats_out, void* stream);
int perf_dp_units(const perf_grid_desc* grid, const int32_t* stats_all, int32_t world, int32_t* headroom_state,
                  int32_t* shifts_out, int64_t* n_total_out, int32_t margin_bits, void* stream);
/* The small all-reduce of a data-parallel step: behind the MLP weight gradient the buffer holds one slot of PERF_DP_SLOT
 * floats per rank; every rank fills ITS slot (the others are zeroed), so a SUM all-reduce doubles as an all-gather.  A slot
 * carries the rank's level_absmax, the largest |field| per level of its slice of THIS step's summed table (field_max of
 * perf_fixed_unfix; integers travel as 16-bit pieces, exact in fp32), its live sample count, its overflow flag (local grid
 * backward or its slice of the summed table) and whether its batch was truncated (*n_marched_dev > capacity > 0).
 * perf_dp_slot_unpack turns the all-reduced slots into (a) job_flags (device, 2 floats) = {overflow, truncated} summed over
 * the ranks -- perf_step_bookkeeping's remote_flags: the step gate is the SAME on every rank --, (b) n_total_out (int64) = the
 * job's sample count, (c) stats_all (may be NULL): the block perf_dp_units reads, as if all-gathered.  With (c) the units of
 * the NEXT step can be derived from THIS step's statistics (perf_dp_units(..., margin_bits = 1): "lagged" units, one bit
 * coarser): the statistics all-gather between the MLP backward and the grid backward leaves the critical path. */
#define PERF_DP_SLOT 80
int perf_dp_slot_pack(const float* level_absmax, const int32_t* field_max, const int64_t* n_dev, int64_t n,
                      const int32_t* overflow_flag, const int64_t* n_marched_dev, int64_t capacity, int32_t rank,
                      int32_t world, float* slots, void* stream);
int perf_dp_slot_unpack(const float* slots, int32_t world, int32_t* stats_all, float* job_flags, int64_t* n_total_out,
                        void* stream);
/* In place: the int32 field pairs of table entries [entry_lo, entry_hi) (a rank's slice after the integer reduce-scatter;
 * `fields` points at entry_lo) -> fp32 gradients, value = field * 2^-shift of the entry's level.  field_max (device,
 * PERF_MAX_LEVELS int32, may be NULL) receives the largest |field| per level of the slice; *overflow_flag is OR-ed with 1
 * when one comes within 4x of the int32 range. */
int perf_fixed_unfix(const perf_grid_desc* grid, void* fields, int64_t entry_lo, int64_t entry_hi,
                     const int32_t* shifts_dev, int32_t* field_max, int32_t* overflow_flag, void* stream);

/* The integer half of the encoding: idx[(l*n + i)*8 + c] = absolute table entry (level offset included) of corner c
 * (bit0 = x, bit1 = y, bit2 = z) of sample i at level l.  Used for the arbitrarily-often differentiable composition
 * behind tcnn.Encoding's double backward (modules/geo_predictors/pano_joint_predictor.py:64-67, create_graph=True). */
int perf_hashgrid_corners(const perf_grid_desc* grid, const float* x01, int32_t* idx, int64_t n, void* stream);

/* tcnn kernel_grid_backward_input: dL/dx01 [n,3] from dfeat and the table (fp32 table). */
int perf_hashgrid_bwd_input(const perf_grid_desc* grid, const float* x01, const float* dfeat,
                            const float* table, float* dx, int64_t n, void* stream);

/* Second order (tcnn kernel_grid_backward_input_backward_*): the backward of perf_hashgrid_bwd_input.  The input gradient
 * gx = perf_hashgrid_bwd_input(x01, dfeat, table) is linear in dfeat and in the table and non-linear in x01; given
 * ggx = dL/d gx [n,3] this writes d_dfeat [L,n,2] = dL/d dfeat (may be NULL) and d_x [n,3] = dL/d x01 through gx (the
 * Hessian-vector product of the interpolation weights; may be NULL; needs dfeat).  fp32 table.  Consumer:
 * SphereDistanceField, modules/geo_predictors/pano_joint_predictor.py:50-69 (autograd.grad(..., create_graph=True)). */
int perf_hashgrid_bwd_bwd_input(const perf_grid_desc* grid, const float* x01, const float* dfeat, const float* table,
                                const float* ggx, float* d_dfeat, float* d_x, int64_t n, void* stream);
/* ... and its piece w.r.t. the table: grad_table [total*2] fp32 = dL/d table through gx (overwritten: zeroed, then
 * scattered with global fp32 atomics -- this consumer's batches are ~10^4 points). */
int perf_hashgrid_bwd_bwd_param(const perf_grid_desc* grid, const float* x01, const float* dfeat, const float* ggx,
                                float* grad_table, int64_t n, void* stream);

/* ---- 64-wide MLP on MFMA ----------------------------------------------------------------- */

/* tcnn kernel_mlp_fused: feat16 (level-major) -> out [n, n_out] fp32 after out_act, multiplied
 * by sel[i] when sel != NULL (ngp_nerf.py:146-149,161). */
int perf_mlp_fwd(const perf_mlp_desc* mlp, const void* w16, const void* feat16, const uint8_t* sel,
                 float* out, int64_t n, const int64_t* n_dev, int dtype, void* stream);

/* Field inference in one boundary call: out [n, n_out] = act(MLP(encode(x01))) * sel, i.e. NGPNeRF.query_density /
 * query_rgb without gradient (modules/fields/ngp_nerf.py:136-162; the no-grad density pass inside
 * OccGridEstimator.sampling and the eval render are this).  Batches of up to 4,096 rows (and grids of <= 16 levels) run as
 * ONE fused kernel: every wave encodes its 32 samples straight into the MFMA B-operand registers of the first layer, the
 * features never travel through memory (measured: from ~16 k rows on the two-kernel path is faster).  Larger batches run perf_hashgrid_fwd + perf_mlp_fwd back to back -- the encode of a
 * large batch is a level-group kernel pinned to XCDs, which a single kernel would have to give up -- with the
 * 16-bit level-major features in `scratch` (perf_field_infer_scratch_bytes(grid, n) bytes, caller owned; may be NULL when
 * feat_out is given or the batch takes the fused kernel).  feat_out (may be NULL): receives the level-major features
 * feat[(l*n + i)*2 + f] as perf_hashgrid_fwd writes them (bit-identical) -- the sampler's density pass keeps them for the
 * gradient pass.  n / n_dev as everywhere. */
int64_t perf_field_infer_scratch_bytes(const perf_grid_desc* grid, int64_t n);
int perf_field_infer(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const uint8_t* sel,
                     const void* table16, const void* w16, float* out, int64_t n, const int64_t* n_dev,
                     void* scratch, int64_t scratch_bytes, void* feat_out, int dtype, void* stream);

/* The whole backward of one field as ONE boundary call (the backward of a tcnn.NetworkWithInputEncoding forward:
 * modules/fields/ngp_nerf.py:142,158 under modules/scene/nerf.py:252-253): perf_mlp_bwd -> perf_hashgrid_bwd into the table part of the
 * same flat gradient -> (fixed != 0 && redo != 0) the predicated fp32 repair launch -- chained on the stream, nothing else runs.
 * grad: fp32 [n_net | 2 * table entries], overwritten.  w16_net: the network part of the 16-bit working copy.  fixed != 0: packed
 * fixed-point accumulation (overflow_flag / headroom_state as in perf_hashgrid_bwd).  workspace: perf_field_bwd_workspace_bytes(...)
 * bytes, 16-byte aligned, caller owned (holds the MLP partials, the tile codes, dfeat [L, n, 2] fp32 and the per-level maxima). */
int64_t perf_field_bwd_workspace_bytes(const perf_grid_desc* grid, const perf_mlp_desc* mlp, int64_t n, int64_t* mlp_ws_bytes,
                                       int64_t* grid_ws_bytes, int64_t* dfeat_bytes);
int perf_field_bwd(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const void* w16_net,
                   const void* feat16, const int32_t* feat_index, int64_t feat_stride, const uint8_t* sel, const float* dout,
                   float* grad, int32_t fixed, int32_t redo, int32_t* overflow_flag, int32_t* headroom_state,
                   void* workspace, int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, void* stream);

/* perf_field_bwd of a fixed-point field with its repair launch (fixed != 0, redo != 0 implied) that also does the step's bookkeeping
 * (ABI 13): ONE thread of the predicated repair launch -- a dispatch that exists in every step and computes nothing in all but a handful
 * per million -- runs perf_step_bookkeeping's arithmetic on `book` (book->overflow_flag must be this call's overflow_flag), one launch
 * per training step fewer.  The one difference: *overflow_flag is NOT cleared (every workgroup of the repair launch reads it as its
 * predicate); pass it to perf_adam_step_dev as clear_flag, or clear it before the next backward.  Everything book names is written
 * before the call's last launch ends: perf_adam_step_dev reads *eff_gate_out / *step_dev / *lr_out next on the same stream. */
int perf_field_bwd_book(const perf_grid_desc* grid, const perf_mlp_desc* mlp, const float* x01, const void* w16_net,
                        const void* feat16, const int32_t* feat_index, int64_t feat_stride, const uint8_t* sel, const float* dout,
                        float* grad, int32_t* overflow_flag, int32_t* headroom_state, void* workspace, int64_t workspace_bytes,
                        int64_t n, const int64_t* n_dev, int dtype, const perf_step_book* book, void* stream);

/* Bytes of caller-owned workspace perf_mlp_bwd needs for n samples. */
int64_t perf_mlp_bwd_workspace_bytes(const perf_mlp_desc* mlp, int64_t n);

/* tcnn kernel_mlp_fused_backward + weight-gradient GEMMs.  The forward is recomputed in
 * registers (nothing but feat16 is kept from the forward pass).  dout [n, n_out] is the
 * gradient w.r.t. the ACTIVATED output (before the sel multiply is undone: the kernel applies
 * sel and the activation derivative itself).  Outputs: dfeat fp32 level-major (may be NULL),
 * dw fp32 [n_net_params] (overwritten; deterministic two-stage reduction), level_absmax (may be NULL):
 * PERF_MAX_LEVELS floats, an upper bound of max |dfeat| of every level (the max over the group of 8 levels a
 * half-wave owns; zeroed by the call).
 * feat_index (device int32 [n], may be NULL) / feat_stride: the features of sample i are row feat_index[i] of a level-major
 * array of feat_stride rows per level (feat16[(l * feat_stride + feat_index[i]) * 2 + f]) -- the kept samples of a batch read
 * straight from the features the sampler's density pass wrote for ALL marched samples (perf_compact_prefix's
 * src_index_out) instead of from a compacted copy; NULL: row i of n rows (feat_stride is ignored). */
int perf_mlp_bwd(const perf_mlp_desc* mlp, const void* w16, const void* feat16, const int32_t* feat_index, int64_t feat_stride,
                 const uint8_t* sel, const float* dout, float* dfeat, float* dw, float* level_absmax, void* workspace,
                 int64_t workspace_bytes, int64_t n, const int64_t* n_dev, int dtype, void* stream);

/* ---- rays ---------------------------------------------------------------------------------- */

/* gen_pano_rays (utils/camera_utils.py:229-234) for rows [row0,row0+nrows) of an H x W panorama.
 * pose: 16 host floats, row major 4x4.  rays_o, rays_d [nrows*W,3]. */
int perf_pano_raygen(const float* pose, int32_t height, int32_t width, int32_t row0, int32_t nrows,
                     float* rays_o, float* rays_d, void* stream);
/* Same with the pose (12+ floats, row major) read from DEVICE memory: a hipGraph holding a whole frame of
 * CoreRunner.render_dense (core_exp_runner.py:223-246) is replayed with a new pose per frame. */
int perf_pano_raygen_dev(const float* pose_dev, int32_t height, int32_t width, int32_t row0, int32_t nrows,
                         float* rays_o, float* rays_d, void* stream);

/* ---- occupancy-grid marching (nerfacc traverse_grids; nerf_renderer.py:145-155) ------------ */

/* nerfacc OccGridEstimator.update_every_n_steps, the pieces around the caller's occupancy closure (PeRF calls it 256 times
 * per episode with a look-up closure, modules/scene/nerf.py:147-168): (1) the jittered evaluation points of cells
 * [cell_lo, cell_lo + n) of a res^3 grid (x-major linear index), x[i] = aabb_lo + ((coord + U[0,1)) / res) * (aabb_hi - aabb_lo),
 * U from the counter-based generator (Philox4x32-10; key = seed, counter = {cell, call}): chunks of one update draw from
 * the same stream whatever the chunking; (2) occs[i] = max(occs[i] * ema_decay, occ[i]) in place, *sum_out (device double,
 * zeroed by the caller before the first chunk) += the new values; (3) binaries[i] = occs[i] > min(*sum / n, occ_thre)
 * (bool bytes), over all n cells.  aabb: 6 host floats. */
int perf_occ_jitter_points(uint64_t seed, uint64_t call, int64_t cell_lo, int64_t n, int32_t res, const float* aabb,
                           float* x, void* stream);
int perf_occ_ema_update(float* occs, const float* occ, int64_t n, float ema_decay, double* sum_out, void* stream);
int perf_occ_threshold(const float* occs, int64_t n, const double* sum, float occ_thre, uint8_t* binaries, void* stream);

/* bool bytes [n_cells] -> bit field (uint32 words, bit i of word w = cell 32*w+i). */
int perf_occ_pack_bits(const uint8_t* binaries, uint32_t* bits, int64_t n_cells, void* stream);

/* number of uint64 mask words per ray for max_steps lattice intervals */
int64_t perf_occ_mask_words(int32_t max_steps);

/* All rays of a launch on ONE lattice (t0 == NULL: eval renders have no stratified jitter): t_k, k = 0 ..
 * perf_occ_lattice_table_len(max_steps) - 1, written once into a caller-owned device buffer; the four marching entry points
 * take it as `lattice_table` (may be NULL) and read a lattice point with one load instead of walking the repeated-addition
 * lattice (without it such launches use a table of runs built on the host; per-ray origins walk on the device). */
int64_t perf_occ_lattice_table_len(int32_t max_steps);
int perf_occ_lattice_table(float t0, float step, int32_t max_steps, int32_t lattice_mode, float* table, void* stream);

/* Per-ray origins (t0 != NULL: the stratified batches of training) on the repeated-addition lattice: the walk of every ray as a
 * table of runs -- inside a binade consecutive lattice points are equidistant bit patterns --, built by ONE launch that gives
 * every ray a lane (the marching kernels would otherwise spend a whole wavefront per ray on it: 13 of march_count's 27 us on a
 * batch of 8,192 rays).  runs: perf_occ_lattice_runs_len(n_rays) 32-bit words, caller-owned; t0 / t0_scale / t0_base as the marching
 * entry points take them.  Those take the buffer as `lattice_table` TOGETHER with t0 != NULL (lattice_mode must be
 * PERF_LATTICE_REPEATED); results are identical with and without it. */
int64_t perf_occ_lattice_runs_len(int64_t n_rays);
int perf_occ_lattice_runs(const float* t0, float t0_scale, float t0_base, int64_t n_rays, float step, int32_t max_steps,
                          int32_t* runs, void* stream);

/* Pass 1: per ray, test lattice intervals k=0..max_steps-1 (t_k on the lattice `lattice_mode`, midpoint
 * inside [max(tmin,t0), min(tmax,far)] and in an occupied cell); writes the keep bit masks
 * (masks [n_rays * mask_words]) and counts [n_rays].  aabb: 6 host floats.
 * Lattice origin of ray r (all four marching entry points): t0 == NULL: t0_base (the near plane); t0_scale == 0: t0[r];
 * otherwise t0 holds the stratified draws u in [0,1) and the origin is fl(u * t0_scale) (+ t0_base when != 0) --
 * OccGridEstimator.sampling's `near_plane + u * render_step_size` formed in the kernel instead of by two torch launches. */
int perf_occ_march_count(const float* rays_o, const float* rays_d, const float* t0, float t0_scale, float t0_base,
                         int64_t n_rays, const uint32_t* occ_bits, const uint32_t* occ_coarse, int32_t res,
                         const float* aabb, float far_plane, float step, int32_t max_steps, int32_t lattice_mode,
                         const float* lattice_table, uint64_t* masks, int32_t* counts, void* stream);

/* perf_occ_march_count that also WRITES the first head_k samples of every ray (the head of the two-phase sampler below):
 * rows r*head_k .. r*head_k + min(count, head_k) - 1 of arrays of n_rays*head_k rows get the same ray_indices / t_starts /
 * t_ends / x01 / sel that perf_occ_march_write_points writes for ranks [0, head_k); the remaining rows of a ray are padding
 * (sel = 0); packed_info[r] = (r*head_k, min(count, head_k)).  Replaces perf_head_tail_counts + perf_exclusive_scan_i32 +
 * perf_occ_march_write_points for the head.  head_k in [1, 64]. */
int perf_occ_march_count_head(const float* rays_o, const float* rays_d, const float* t0, float t0_scale, float t0_base,
                              int64_t n_rays, const uint32_t* occ_bits, const uint32_t* occ_coarse, int32_t res, const float* aabb,
                              float far_plane, float step, int32_t max_steps, int32_t lattice_mode, const float* lattice_table,
                              uint64_t* masks, int32_t* counts, int32_t head_k, int64_t* ray_indices, float* t_starts, float* t_ends, int32_t* packed_info,
                              const float* points_aabb6, float* x01, uint8_t* sel, void* stream);

/* Optional empty-space skip for perf_occ_march_count (what nerfacc's DDA traversal achieves): a dilated 4^3-block
 * occupancy (perf_occ_coarse_words(res) uint32 words: a dilated 4^3-block grid followed by a dilated 2^3-block grid) lets the kernel drop whole 64-interval chunks; conservative,
 * results are identical with occ_coarse == NULL.  res must be a multiple of 8. */
int64_t perf_occ_coarse_words(int32_t res);
int perf_occ_build_coarse(const uint32_t* occ_bits, int32_t res, uint32_t* coarse, void* stream);

/* Exclusive prefix sum of int32 (counts -> offsets); total [1] (int64, device) receives the sum; total_biased (device
 * int64, may be NULL) receives sum + total_bias (the two-phase sampler's evaluated-sample count = head rows + tail samples,
 * without a separate add).  workspace >= perf_scan_workspace_bytes(n).  The offsets are int32 (nerfacc's packed_info contract):
 * the caller keeps the sum below 2^31 (sample capacities are); `total` itself is exact in int64. */
int64_t perf_scan_workspace_bytes(int64_t n);
int perf_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t* total, int64_t n, int64_t total_bias,
                            int64_t* total_biased, void* workspace, int64_t workspace_bytes, void* stream);

/* Pass 2: expand masks into packed samples sorted by ray then t.  packed_info [n_rays,2] =
 * (start,count) int32.  capacity = allocated length of the sample arrays. */
int perf_occ_march_write(const float* t0, float t0_scale, float t0_base, int64_t n_rays, float step, int32_t max_steps, int32_t lattice_mode,
                         const float* lattice_table, const uint64_t* masks, const int32_t* counts, const int32_t* offsets,
                         int64_t capacity, int64_t* ray_indices, float* t_starts, float* t_ends,
                         int32_t* packed_info, void* stream);

/* perf_occ_march_write that also emits the sample positions of perf_points_from_rays (x01 [S,3], sel [S] or NULL) for
 * the samples it writes: one launch and one pass over the samples less when no visibility compaction follows.
 * aabb6: host pointer, {min xyz, max xyz}.  rank_lo: the samples of rank [rank_lo, rank_lo + counts[r]) of every ray are
 * written (rank = position among the ray's samples in t order; 0 with the march counts = everything) -- the two-phase
 * sampler below writes the first K samples of every ray first and the rest of the surviving rays later. */
int perf_occ_march_write_points(const float* t0, float t0_scale, float t0_base, int64_t n_rays, float step, int32_t max_steps, int32_t lattice_mode,
                                const float* lattice_table, const uint64_t* masks,
                                const int32_t* counts, const int32_t* offsets, int64_t capacity, int64_t* ray_indices,
                                float* t_starts, float* t_ends, int32_t* packed_info, const float* rays_o,
                                const float* rays_d, const float* aabb6, float* x01, uint8_t* sel, int32_t rank_lo,
                                void* stream);

/* ---- compositing (nerfacc render_weight_from_density / accumulate_along_rays /
 *      render_visibility_from_density; nerf_renderer.py:170-183) ---------------------------- */

/* Per-ray exclusive sum of sigma*delta in the canonical 64-chunk Kogge-Stone order, thresholded:
 * new_counts[r] = number of leading samples with exclusive_sum <= thr (thr = -ln(early_stop_eps)).
 * Also writes exsum [S] when not NULL.  tail_counts (may be NULL; with march_counts [R] and head_samples = K): the
 * two-phase sampler's tail counts in the same launch -- count - K for a ray whose whole head min(count, K) survived, else 0
 * (= perf_head_tail_counts(counts, K, new_counts)). */
int perf_visibility_count(const float* sigmas, const float* t_starts, const float* t_ends,
                          const int32_t* packed_info, int64_t n_rays, float thr, int32_t* new_counts,
                          float* exsum, const int32_t* march_counts, int32_t head_samples, int32_t* tail_counts,
                          void* stream);

/* Copy the first new_counts[r] samples of every ray to new_offsets[r] (the boolean-mask
 * compaction of nerfacc's sampling).  sigmas_in/out may be NULL; so may the sample positions x01 [S,3] / sel [S]
 * (as written by perf_occ_march_write_points), which are then compacted along instead of being recomputed, and the
 * level-major 16-bit features of the density pass (feat[l * stride + i], n_levels levels; NULL = none): the reference
 * evaluates the density field a second time on the kept samples (nerf_renderer.py:166-168) with the same parameters at the
 * same positions, so the compacted features ARE that evaluation's encoding.  src_index_out (int32 [S], may be NULL): the
 * source row of every kept sample instead -- perf_mlp_bwd(feat_index) then reads the uncompacted features in place and the
 * 128 B per sample of feature copy (two thirds of this call's traffic) do not happen. */
int perf_compact_prefix(const int32_t* packed_info, const int32_t* new_counts, const int32_t* new_offsets,
                        int64_t n_rays, const float* ts_in, const float* te_in, const float* sig_in,
                        int64_t* ray_indices_out, float* ts_out, float* te_out, float* sig_out,
                        int32_t* packed_out, const float* x01_in, const uint8_t* sel_in, float* x01_out,
                        uint8_t* sel_out, const void* feat_in, int64_t feat_stride_in, void* feat_out,
                        int64_t feat_stride_out, int32_t n_levels, int32_t* src_index_out, void* stream);

/* ---- two-phase early termination (same results as perf_visibility_count + perf_compact_prefix over all samples) -------
 * nerfacc's render_visibility_from_density keeps a PREFIX of every ray (the exclusive sum of sigma*delta never decreases),
 * so the density of samples behind the first one over the threshold never matters.  The sampler evaluates the first K
 * samples of every ray (head), then only the rest (tail) of the rays whose head survived entirely:
 *   perf_head_tail_counts(counts, K, NULL) -> head counts;   ... density of the heads, perf_visibility_count -> kept_head;
 *   perf_head_tail_counts(counts, K, kept_head) -> tail counts (0 for decided rays);   ... density of the tails;
 *   perf_visibility_count2 / perf_compact_prefix2 over (head, tail) pairs of arrays -> final packed samples.
 * Sample i of ray r is head sample i for i < packed_h[r].count, tail sample i - packed_h[r].count otherwise; the canonical
 * scan value of a sample only involves earlier samples, so kept counts are bit-identical to the one-phase path. */
int perf_head_tail_counts(const int32_t* counts, int64_t n_rays, int32_t head_samples, const int32_t* kept_head,
                          int32_t* out_counts, void* stream);
int perf_visibility_count2(const float* sig_h, const float* ts_h, const float* te_h, const int32_t* packed_h,
                           const float* sig_t, const float* ts_t, const float* te_t, const int32_t* packed_t,
                           int64_t n_rays, float thr, int32_t* new_counts, void* stream);
/* capacity: rows of the output arrays (a batch that keeps more is truncated ray by ray, like perf_occ_march_write);
 * sig/x01/sel outputs (and the matching inputs) may be NULL. */
int perf_compact_prefix2(const float* sig_h, const float* ts_h, const float* te_h, const int32_t* packed_h,
                         const float* x01_h, const uint8_t* sel_h, const float* sig_t, const float* ts_t,
                         const float* te_t, const int32_t* packed_t, const float* x01_t, const uint8_t* sel_t,
                         const int32_t* new_counts, const int32_t* new_offsets, int64_t n_rays, int64_t capacity,
                         int64_t* ray_indices_out, float* ts_out, float* te_out, float* sig_out, float* x01_out,
                         uint8_t* sel_out, int32_t* packed_out, const void* feat_h, int64_t feat_stride_h,
                         const void* feat_t, int64_t feat_stride_t, void* feat_out, int64_t feat_stride_out,
                         int32_t n_levels, void* stream);

/* weights/trans/alphas [S] and per-ray opacity [R], distance [R], colour [R,3] (rgbs may be NULL).
 * One wave per ray: no atomics. */
int perf_composite_fwd(const float* sigmas, const float* rgbs, const float* t_starts,
                       const float* t_ends, const int32_t* packed_info, int64_t n_rays,
                       float* weights, float* trans, float* alphas, float* opacity, float* distance,
                       float* color, void* stream);

/* Backward of the above w.r.t. sigmas (and rgbs when d_rgbs != NULL).  Incoming gradients, all optional
 * (NULL = zero): per sample g_weights, g_trans, g_alphas [S]; per ray g_opacity [R], g_distance [R],
 * g_color [R,3] (colour uses detached weights as nerf_renderer.py:183, so it only feeds d_rgbs). */
int perf_composite_bwd(const float* sigmas, const float* t_starts, const float* t_ends,
                       const int32_t* packed_info, int64_t n_rays, const float* weights, const float* trans,
                       const float* g_weights, const float* g_trans, const float* g_alphas,
                       const float* g_opacity, const float* g_distance, const float* g_color,
                       float* d_sigmas, float* d_rgbs, void* stream);

/* The geometry step's pair of compositing + distortion loss (nerf_renderer.py:170-183 + nerf.py:222-236) as ONE kernel
 * each way.  Forward: perf_composite_fwd plus distloss_per_ray [R] (= perf_distloss_fwd on the weights it forms).
 * Backward: d_sigmas of perf_composite_bwd with g_weights = distloss_scale * distloss_scale_dev[0] * d(distortion
 * loss)/d weights formed in the kernel; opacity / distance are the forward's per-ray outputs (they are the totals
 * sum w and sum w t_mid the distortion gradient needs).  distloss_scale_dev may be NULL. */
int perf_composite_distloss_fwd(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                                const int32_t* packed_info, int64_t n_rays, float* weights, float* trans,
                                float* opacity, float* distance, float* color, float* distloss_per_ray, void* stream);
int perf_composite_distloss_bwd(const float* sigmas, const float* t_starts, const float* t_ends,
                                const int32_t* packed_info, int64_t n_rays, const float* weights, const float* trans,
                                const float* opacity, const float* distance, const float* g_opacity,
                                const float* g_distance, float distloss_scale, const float* distloss_scale_dev,
                                float* d_sigmas, void* stream);

/* Eval tail of NeRFOCCRenderer.render (nerf_renderer.py:195-197), in place: distance[r] += 5 (1 - opacity[r]),
 * color[r,:] += 0.5 (1 - opacity[r]); nothing happens when n_dev != NULL and *n_dev <= 0 (a batch without samples
 * returns zeros before that tail in the reference, :156-162).  distance / color may be NULL. */
int perf_render_finish_eval(const float* opacity, float* distance, float* color, int64_t n_rays,
                            const int64_t* n_dev, void* stream);

/* nerfacc.accumulate_along_rays forward (nerf_renderer.py:173-183): out[r,c] = sum_i w_i * values[i,c]
 * (values == NULL: n_channels must be 1 and out[r] = sum_i w_i).  One wave per ray, no atomics. */
int perf_accumulate_fwd(const float* weights, const float* values, const int32_t* packed_info, int64_t n_rays,
                        int32_t n_channels, float* out, void* stream);

/* nerfacc.pack_info: (start,count) per ray from SORTED int64 ray_indices [n]. */
int perf_pack_info(const int64_t* ray_indices, int64_t n, int64_t n_rays, int32_t* packed_info, void* stream);

/* flatten_eff_distloss forward (per-ray partial losses [R], caller sums and divides by n_rays)
 * and analytic gradient w.r.t. w (modules/scene/nerf.py:226-230). */
int perf_distloss_fwd(const float* w, const float* t_starts, const float* t_ends,
                      const int32_t* packed_info, int64_t n_rays, float* loss_per_ray, void* stream);
int perf_distloss_bwd(const float* w, const float* t_starts, const float* t_ends,
                      const int32_t* packed_info, int64_t n_rays, float scale, const float* scale_dev,
                      float* g_w, void* stream);   /* scale_dev (device, may be NULL) multiplies scale */

/* ---- fused loss heads of the two training steps ----------------------------------------------------------------
 * Geometry step (modules/scene/nerf.py:208-252 + the training-time terms of nerf_renderer.py:193): d' = relu(distance +
 * (2 noise - 1)(1 - opacity)); depth loss = smooth_l1(d', gt, beta 1e-2) summed / global_batch; distortion loss =
 * sum(distloss_per_ray) / (last ray with samples + 1).  Writes the per-ray gradients of
 * loss_scale * (depth_weight * depth + distortion_weight * ratio * distortion) w.r.t. opacity and distance, and
 * scalars[0..2] = {depth loss, distortion loss, scale to pass to perf_distloss_bwd as scale_dev}; `scalars` is
 * PERF_LOSS_SCALARS floats of device memory (the rest holds per-workgroup partial sums, folded in a fixed order by the
 * last workgroup to finish -- one launch).  ticket: one device int32, zero before the first call and left at zero by every
 * call (not shared by calls that may run concurrently).
 * ratio_dev: device float (min(2 progress, 1), nerf.py:235) or NULL (= 1).  noise may be NULL (no noise term). */
#define PERF_LOSS_SCALARS 256
int perf_geo_loss(const float* opacity, const float* distance, const float* gt_distance, const float* noise,
                  const float* distloss_per_ray, const int32_t* packed_info, int64_t n_rays, int64_t global_batch,
                  float depth_weight, float distortion_weight, const float* ratio_dev, float loss_scale,
                  float* g_opacity, float* g_distance, float* scalars, int32_t* ticket, void* stream);

/* Colour step (nerf.py:281-293 + nerf_renderer.py:194): c' = color + bg (1 - opacity); loss = smooth_l1(c', gt, beta 5e-2)
 * summed / (3 global_batch); g_color [R,3] = d(loss_scale * color_weight * loss)/d color; scalars[0] = loss
 * (scalars: PERF_LOSS_SCALARS floats, as above). */
int perf_app_loss(const float* opacity, const float* color, const float* bg_color, const float* gt_color,
                  int64_t n_rays, int64_t global_batch, float color_weight, float loss_scale, float* g_color,
                  float* scalars, void* stream);

/* ---- the whole ray head of a training step in one launch -----------------------------------------------------------
 * perf_composite_distloss_fwd -> perf_geo_loss -> perf_composite_distloss_bwd (geometry step, nerf.py:195-252) and
 * perf_composite_fwd -> perf_app_loss -> perf_composite_bwd (colour step, nerf.py:268-293) as ONE kernel each, one wavefront per
 * ray, the same arithmetic expression by expression: nothing in either loss couples two rays except the batch size (an argument)
 * and, for the distortion loss, "the last ray that holds a sample", which every wavefront reads off the end of packed_info.
 * Outputs: weights / trans [n_samples], opacity / distance [R], color [R,3] (geometry: only with rgbs), the per-ray loss TERMS
 * (geometry: depth_terms[r] = smooth_l1(d' - gt), distloss_per_ray[r]; colour: color_terms[r] = sum over the three channels) and
 * the gradient the field backward starts from (d_sigmas [n_samples]; colour: d_rgbs [n_samples,3]).  The loss values are reports:
 * depth = sum(depth_terms) / global_batch, distortion = sum(distloss_per_ray) * inv_n_out[0], colour = sum(color_terms) /
 * (3 global_batch) -- left to whoever reads them.  A local batch smaller than global_batch (data parallel) normalises the
 * distortion loss by the global batch like perf_geo_loss.  noise, ratio_dev, bg_color, rgbs (geometry), color (geometry),
 * inv_n_out may be NULL.  sample_rows (ABI 12): the rows of the per-sample arrays -- the capacity in sync-free mode, a HINT about the
 * samples per ray that only selects the launch shape: fewer than 32 rows per ray (or <= 0: unknown) -> rays of <= 16 / <= 4 samples
 * are served by 16- / 4-lane teams (a training batch late in an episode keeps a sample or two per ray), else a wavefront per ray.
 * The same bits either way. */
int perf_train_head_geo(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                        const int32_t* packed_info, int64_t n_rays, int64_t sample_rows, const float* gt_distance, const float* noise,
                        int64_t global_batch, float depth_weight, float distortion_weight, const float* ratio_dev,
                        float loss_scale, float* weights, float* trans, float* opacity, float* distance, float* color,
                        float* depth_terms, float* distloss_per_ray, float* inv_n_out, float* d_sigmas, void* stream);
int perf_train_head_app(const float* sigmas, const float* rgbs, const float* t_starts, const float* t_ends,
                        const int32_t* packed_info, int64_t n_rays, int64_t sample_rows, const float* bg_color, const float* gt_color,
                        int64_t global_batch, float color_weight, float loss_scale, float* weights, float* trans,
                        float* opacity, float* distance, float* color, float* color_terms, float* d_rgbs, void* stream);

/* Supervision batch (SupInfoPool.rand_ray_color_data, modules/dataset/sup_info.py:236-259): out[i] = all[indices[i]] for
 * every non-NULL output (origins/directions/colours/normals [n,3], distances [n]) in one launch. */
int perf_gather_supervision(const int64_t* indices, int64_t n, const float* o_all, const float* d_all,
                            const float* color_all, const float* dist_all, const float* normal_all, float* o, float* d,
                            float* color, float* dist, float* normal, void* stream);

/* The whole batch draw of a training step in ONE launch: SupInfoPool.rand_ray_color_data (sup_info.py:236-259: uniform indices
 * into the pool range [pool_lo, pool_hi) + the five gathers) and the per-ray uniform draws of the step -- the stratified
 * jitter (nerf_renderer.py:152, nerfacc), the distance noise (:193) and the background colour (:185) -- from a counter-based
 * generator (Philox4x32-10, key = seed, counter = {position in the global batch, *counter_dev}).  Position = first_global +
 * i: the ranks of a data-parallel job pass the same seed, hold the same counter and draw disjoint slices of ONE global
 * batch.  The last workgroup advances *counter_dev (device int64) through `ticket` (device int32, zero before the first call,
 * left at zero).  A captured training step then contains no torch random-number op -- whose replays cost two extra launches
 * (seed / offset refresh of the graph-safe generator).  Outputs may be NULL individually; uniforms are 24-bit, in [0, 1). */
int perf_draw_train_batch(uint64_t seed, int64_t* counter_dev, int32_t* ticket, int64_t pool_lo, int64_t pool_hi,
                          int64_t n_local, int64_t first_global, const float* o_all, const float* d_all,
                          const float* color_all, const float* dist_all, const float* normal_all, float* o, float* d,
                          float* color, float* dist, float* normal, int64_t* indices_out, float* jitter, float* noise,
                          float* bg, void* stream);

/* ---- hierarchical resampling (nerfacc importance_sampling via PropNetEstimator.sampling,
 *      modules/scene/nerf_renderer.py:60-70; dead in the reference, semantics restated in oracle/) ------------- */

/* Per ray: n_in intervals with edges s_in [R, n_in+1] and CDF cdf [R, n_in+1] (non-decreasing, 0..1) ->
 * n_out+1 sorted edges s_out [R, n_out+1] at u_j = (j + tau_r)/(n_out+1); tau [R] (stratified) or NULL (0.5). */
int perf_pdf_resample(const float* s_in, const float* cdf, const float* tau, int64_t n_rays, int32_t n_in,
                      int32_t n_out, float* s_out, void* stream);

/* ---- reprojection visibility tests (modules/scene/nerf.py:321-358, modules/dataset/sup_info.py:261-302) ---------------
 * For every point pts[i] (a rendered panorama's back-projected surface point): direction and distance in the frame of a
 * registered panorama (pose: 16 host floats, row major 4x4, camera-to-world), equirect image coordinate
 * (utils/camera_utils.py:134-151), bilinear look-up of distance_map [height, width] (the panorama's distance map already
 * multiplied by its validity mask) with grid_sample(padding_mode='border', align_corners=False) arithmetic, then
 *   mode 0: mask[i] = max(mask[i], dist < looked-up + eps)      (get_pano_visibility_mask: eps = 1/256)
 *   mode 1: mask[i] = min(mask[i], looked-up < dist)             (geo_check)
 * Call once per registered panorama on a mask initialised to 0 (mode 0) / 1 (mode 1). */
int perf_pano_reproject(const float* pts, int64_t n, const float* pose, const float* distance_map, int32_t height,
                        int32_t width, int32_t mode, float eps, float* mask, void* stream);
/* Binary morphology of a 0/1 float image with a structuring element given as row bit masks (bit c of row_bits[r] = element
 * (r, c); rows <= 16, cols <= 32; OpenCV's MORPH_ELLIPSE elements are built by the caller).  op 0: dilation, pixels outside
 * the image count as background; op 1: erosion, outside counts as foreground (kornia's geodesic border, nerf.py:354-355).
 * in != out. */
int perf_morph_binary(const float* in, float* out, int32_t height, int32_t width, const uint32_t* row_bits, int32_t rows,
                      int32_t cols, int32_t op, void* stream);

/* ---- occupancy pre-grid (modules/dataset/sup_info.py:304-330) ------------------------------ */
int perf_occ_splat(const float* rays_o, const float* rays_d, const float* dist, int64_t n,
                   int32_t res, uint8_t* occ, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PERF_HIP_H */
