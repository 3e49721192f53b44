/*
 * curious_hip.h -- C ABI of libcurious_hip.so (MI355X / gfx950).
 *
 * The reference (flowersteam/curious) is 100 % Python and has no FFI; its hot path leans on
 * NumPy / TensorFlow-1 / mpi4py.  This header is the native boundary a maintainer binds (ctypes,
 * see INTEGRATION.md) to replace those calls.  Every entry point cites the reference lines it
 * replaces.  Conventions:
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work,
 *     they never synchronise, allocate or free (safe to capture in a hipGraph);
 *   - return value 0 = success, negative = error; curious_last_error() returns the message;
 *   - matrices are row-major float32 unless stated; "rows" of a batch are `stride` floats apart.
 */
#ifndef CURIOUS_HIP_H
#define CURIOUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CURIOUS_ABI_VERSION 10     /* bumped whenever a prototype or struct below changes */
#define CURIOUS_MAX_TASKS 16
#define CURIOUS_MAX_TASK_DIMS 8

typedef void* curious_stream_t;

/* ---- replay storage: one record per (episode, t), t in [0, T] -------------------------------
 * Replaces the dict of float64 arrays {key: [size, T(+1), dim]} of replay_buffer.py:23-24.  A record
 * row holds [o | ag | g | u | task_descr | extra] where `extra` carries the keys the sampler only
 * copies through (change[dimag], info_*), so that transition (ep,t) and its successor (o_2, ag_2 =
 * row t+1, replay_buffer.py:47-48) are two adjacent rows.  Rows t = T hold only o and ag.
 * Values are float32 (the rollout produces float32, rollout.py:50-52,194-195; float64 storage in the
 * reference holds the same values). */
typedef struct curious_layout {
  int32_t T;
  int32_t dimo, dimag, dimg, dimu, dimtd, dimextra;
  int32_t off_o, off_ag, off_g, off_u, off_td, off_extra;
  int32_t row_stride;          /* floats per record row */
} curious_layout_t;

/* ---- staged batch: one row per sampled transition -------------------------------------------
 * Replaces the dict returned by _sample_her_transitions (her.py:110-183) and the 8 staged arrays of
 * DDPG.sample_batch (ddpg.py:350-358).  Row: [o | td | u | g | o_2 | g_2 | r | ag | ag_2 | extra]. */
typedef struct curious_batch_layout {
  int32_t off_o, off_td, off_u, off_g, off_o2, off_g2, off_r, off_ag, off_ag2, off_extra;
  int32_t stride;              /* floats per batch row */
} curious_batch_layout_t;

/* task tables (tasks_g_id / tasks_ag_id of the env; her.py:145-149 truncates ag ids to len(g ids)) */
typedef struct curious_tasks {
  int32_t ntasks;
  int32_t len[CURIOUS_MAX_TASKS];
  int32_t g_id[CURIOUS_MAX_TASKS][CURIOUS_MAX_TASK_DIMS];
  int32_t ag_id[CURIOUS_MAX_TASKS][CURIOUS_MAX_TASK_DIMS];
} curious_tasks_t;

/* sampler behaviour (her.py:72-97) */
enum {
  CURIOUS_RELABEL_BUFFER_TASK = 0,   /* multi-buffer modes: replay task = buffer's task, or the sample's
                                        own task when task_to_replay < 0 (her.py:131-136) */
  CURIOUS_RELABEL_GIVEN_TASK = 1,    /* single-buffer random / cp modes: per-sample task supplied in
                                        task_to_replay[] (her.py:138-142) */
  CURIOUS_RELABEL_CURRENT_TASK = 2,  /* replay_current_task_transition: keep task_descr (her.py:158-164) */
  CURIOUS_RELABEL_FLAT = 3           /* make_sample_her_transitions: whole goal from future ag (her.py:43-47) */
};

typedef struct curious_sample_params {
  double future_p;          /* her.py:86-89 */
  double reward_eps;        /* sparse L2 threshold of the reward (config.py:158-159 -> env) */
  float clip_obs;           /* ddpg.py:125-126; +inf = ReplayBuffer.sample (no clipping) */
  int32_t relative_goals;   /* ddpg.py:119-124 */
  int32_t relabel_mode;     /* CURIOUS_RELABEL_* */
  int32_t flat_reward;      /* 1: reward over all goal slots, task_descr ignored (her.py:57) */
} curious_sample_params_t;

/* Host-drawn sample plan (parity mode): the NumPy legacy stream is consumed on the host in the
 * reference's order (her.py:108-116, ddpg.py:326-345) and handed over as arrays of length n. */
typedef struct curious_sample_plan {
  const int32_t* buf;             /* buffer index of each sample */
  const int32_t* ep;              /* episode_idxs   (her.py:108) */
  const int32_t* t;               /* t_samples      (her.py:109) */
  const double* u_her;            /* uniform draws  (her.py:115) */
  const double* u_off;            /* uniform draws  (her.py:116) */
  const int32_t* task_to_replay;  /* per sample, <0 = None */
  const int32_t* out_row;         /* destination row (inverse of ddpg.py:338-345 shuffle); NULL = identity */
} curious_sample_plan_t;

/* Device-drawn plan (throughput mode): Philox4x32-10 keyed by (seed, *step_ctr). */
typedef struct curious_sample_rng {
  uint64_t seed;
  const int64_t* step_ctr;        /* device counter, advanced by curious_ddpg_grads */
  int64_t step_host;              /* used when step_ctr == NULL */
  const int32_t* prop_prefix;     /* [nbuf+1] exclusive prefix of DDPG.proportions (ddpg.py:282-286) */
  const int32_t* cur_size;        /* [nbuf] episodes stored in each LOGICAL buffer (replay_buffer.py:27) */
  const int32_t* buf_alias;       /* [nbuf] physical buffer of each logical buffer (ddpg.py:106-110) */
  const int32_t* buf_task;        /* [nbuf] task_to_replay of each logical buffer, <0 = None */
  int32_t nbuf;
  /* Virtual ranks (several MPI ranks' batches drawn by ONE call; config.py:210-214 builds the buffers per process and
   * train.py:242-243 seeds every rank with seed + 1000000 * rank: both stay private to a rank here).  rank_rows > 0: sample
   * i belongs to rank r = i / rank_rows and is that rank's sample i' = i - r * rank_rows -- it is drawn exactly as
   * curious_her_sample would draw sample i' of a call with n = rank_rows, the four tables rank_tab_stride int32 elements
   * further (r times) and the Philox key seed + r * rank_seed_stride -- and lands in batch row i.  Each rank's tables name
   * ITS buffers through buf_alias (all ranks' buffers are slots of the one `storage`).  0: one rank. */
  int32_t rank_rows;
  int64_t rank_tab_stride;
  uint64_t rank_seed_stride;
} curious_sample_rng_t;

const char* curious_last_error(void);
int curious_abi_version(void);
/* sha256 of the sources (csrc, this header, compiler flags) the library was built from; the Python binding compares it
 * with the sources next to it and refuses a stale binary */
const char* curious_build_digest(void);
/* name of the device the library runs on + its CU count; fails when no gfx950 device is present */
int curious_device_info(char* name_host, int name_len, int* cu_count_host);

/* Per-kernel HIP-event timing for bench.py's roofline: while enabled, every kernel launch of this library is
 * bracketed by two events recorded on its launch stream.  Not for use during hipGraph capture.
 * curious_prof_collect synchronises and returns, per kernel id, the launch count and summed duration (ms). */
int curious_prof_enable(int on);
int curious_prof_kernel_count(void);
const char* curious_prof_kernel_name(int kid);
int curious_prof_collect(int64_t* counts_host, double* total_ms_host);
/* launches per kernel id since the library was loaded ([curious_prof_kernel_count()] entries), counted whether or not
 * event timing is on; a launch captured into a hipGraph counts once.  The GPU test session uses it to assert that no
 * kernel of the library went unexercised. */
int curious_prof_launch_counts(int64_t* counts_host);

/* Run-time options, read at every call (so one process can run both routes).  Names:
 *   "rows"         1 (default; env CURIOUS_ROWS): the row-local routes; 0: the tiled multi-launch routes
 *   "rows_xcd"     1 (default; env CURIOUS_ROWS_XCD): workgroup kinds of the row-local update placed by XCD
 *   "rows_pre"     1 (default; env CURIOUS_ROWS_PRE): the row-local update's role, input rows and first layer-0 matrix are
 *                  handed to the kernel as leading arguments (in scalar registers when a wave starts) where the shapes
 *                  allow; 0: fetched from the argument segment (same results; A/B)
 *   "rows8"        1 (default; env CURIOUS_ROWS8): the row-local update gives 8 batch rows to a workgroup instead of 4 when the
 *                  batch has >= 768 rows (three virtual ranks or more: half the weight stream per row); 0: always 4 (A/B; the
 *                  same bits -- the rows are independent and every row's arithmetic keeps its order)
 *   "dw_split"     0 (default; env CURIOUS_DW_SPLIT): batches of >= 1 024 rows split the reduction over the batch rows of the
 *                  SMALL weight-gradient tiles (layer-0 segments, output layers) over min(4, chunks of 256 rows / 2)
 *                  workgroups, the last of which to arrive adds the partial tiles in segment order and runs the optimiser;
 *                  10 h + s: h workgroups per hidden-layer tile, s per small tile (11 = no split; A/B)
 *   "dw_xcd"       1 (default; env CURIOUS_DW_XCD): blocks of the weight-gradient / optimiser launch placed by XCD
 *   "xcd_map"      0 (default; env CURIOUS_XCD_MAP) / 4 / 8: XCD-aware block placement of the tiled hidden-layer kernels
 *   "fault_inject" 0 (default) / k > 0: the producer of Q' of row group k - 1 never publishes; a member of group k - 1 of
 *                  the resident-weights rollout never shows up (fault-path tests)
 *   "qt_spins"     2^22 (default): polls before a consumer of Q' gives up and raises the workspace's fault word
 *   "resident"     1 (default; env CURIOUS_RESIDENT): curious_policy_rollout keeps the actor's hidden matrices in LDS for
 *                  the whole episode (groups of 4 workgroups per 4 envs) when n <= the device's CU count; 0: every
 *                  workgroup streams them at every step
 *   "res_spins"    2^20 (default): polls before a member of such a group gives up on a peer (flags[n] = 2)
 * Options are baked into a launch when it is enqueued (also into captured graphs). */
int curious_set_option(const char* name, int64_t value);
int64_t curious_get_option(const char* name);     /* -1 + curious_last_error() for an unknown name */

/* HER sample + goal/task relabel + reward + clip, written already permuted.
 * Replaces replay_buffer.py:37-55, her.py:99-183 (or :20-66), ddpg.py:326-353.
 * storage: [nbuf_phys][capacity][T+1][row_stride]; buf_stride = floats between buffers.
 * Exactly one of plan / rng is non-NULL. */
int curious_her_sample(const float* storage, int64_t buf_stride, const curious_layout_t* L,
                       const curious_tasks_t* tasks, const curious_sample_params_t* P,
                       const curious_sample_plan_t* plan, const curious_sample_rng_t* rng,
                       int32_t n, float* batch, const curious_batch_layout_t* BL, curious_stream_t stream);

/* Copy n_pairs episodes (T+1 rows each) from a staging block into replay slots.
 * Replaces replay_buffer.py:57-72 + the routing loop ddpg.py:178-197.
 * pair_src[i] = episode index in `staging`; pair_dst[i] = buffer*capacity + slot. */
int curious_store_episodes(float* storage, const float* staging, const curious_layout_t* L,
                           const int32_t* pair_src, const int64_t* pair_dst, int32_t n_pairs,
                           curious_stream_t stream);

/* The routing of DDPG.store_episode (ddpg.py:178-197) decided ON THE DEVICE + the copy (device RNG mode): episode b goes
 * to logical buffer 1 + j of every task j < n_route whose flag active[b * ntasks + j] is set (curious_episode_activity),
 * episodes in ascending order -- the order of the reference's loop: consecutive slots from cur_size[1 + j] on while the
 * buffer has room (replay_buffer.py:94-95), a random slot each once it is full (replay_buffer.py:101-102: a Philox draw
 * keyed by (seed, call, episode, task) instead of np.random.randint; curious_store_slots_host gives the same numbers on
 * the host); of two episodes of the batch on one slot the later one wins.  cur_size[1 + j] (device int32: the `cur_size`
 * table of curious_sample_rng_t) advances accordingly, capped at capacity.  buf_alias[i] = pool slot of logical buffer
 * i; capacity in episodes.  Nothing is stored when skip != NULL and *skip != 0 (the NaN word of the rollout flags).
 * pair_src / pair_dst / n_pairs: device scratch of n_episodes * n_route entries / 1 entry; they hold the routing
 * afterwards (src = -1: the pair lost its slot).  The host never has to wait for the activity flags before the updates
 * can be enqueued. */
int curious_route_store_episodes(float* storage, const float* staging, const curious_layout_t* L, const int32_t* active,
                                 int32_t ntasks, int32_t n_route, int32_t n_episodes, int32_t* cur_size,
                                 const int32_t* buf_alias, int64_t capacity, uint64_t seed, uint64_t call,
                                 const float* skip, int32_t* pair_src, int64_t* pair_dst, int32_t* n_pairs,
                                 curious_stream_t stream);
/* The same for the episodes of n_ranks VIRTUAL RANKS in one call (the reference: one process per rank, each storing its
 * own rollouts into its own buffers, config.py:210-214): `staging` holds n_ranks * n_episodes records, rank v's are records
 * v * n_episodes .. (v + 1) * n_episodes - 1 (`active` likewise); its size / alias tables are cur_size / buf_alias +
 * v * tab_stride (int32 elements), its slot draws use the key seed + v * seed_stride and ITS episode numbers
 * 0 .. n_episodes - 1; pair_src / pair_dst: n_ranks * n_episodes * n_route entries (rank v's segment at v * n_episodes *
 * n_route, src = record index in `staging`), n_pairs: n_ranks entries.  Rank by rank the result is what
 * curious_route_store_episodes gives for that rank alone.  n_episodes <= 2048 per rank. */
int curious_route_store_episodes_ranks(float* storage, const float* staging, const curious_layout_t* L,
                                       const int32_t* active, int32_t ntasks, int32_t n_route, int32_t n_episodes,
                                       int32_t n_ranks, int32_t* cur_size, const int32_t* buf_alias, int64_t tab_stride,
                                       int64_t capacity, uint64_t seed, uint64_t seed_stride, uint64_t call,
                                       const float* skip, int32_t* pair_src, int64_t* pair_dst, int32_t* n_pairs,
                                       curious_stream_t stream);
/* curious_episode_activity + curious_route_store_episodes_ranks in two launches instead of three: the routing launch
 * first evaluates the activity flags itself (ddpg.py:179-184; `change` in the extra block at off_change) and writes them
 * to `active` ([n_ranks * n_episodes * tasks->ntasks], an OUTPUT here: the host mirrors the buffer sizes from it later). */
int curious_activity_route_store_episodes(float* storage, const float* staging, const curious_layout_t* L,
                                          const curious_tasks_t* tasks, int32_t off_change, int32_t* active,
                                          int32_t n_route, int32_t n_episodes, int32_t n_ranks, int32_t* cur_size,
                                          const int32_t* buf_alias, int64_t tab_stride, int64_t capacity, uint64_t seed,
                                          uint64_t seed_stride, uint64_t call, const float* skip, int32_t* pair_src,
                                          int64_t* pair_dst, int32_t* n_pairs, curious_stream_t stream);
/* Host-side twin of the random slots above (no GPU involved): out[i] = slot of episode episodes[i] routed to `task`. */
int curious_store_slots_host(uint64_t seed, uint64_t call, int32_t task, int64_t size, int32_t n,
                             const int32_t* episodes, int64_t* out);

/* Per-episode task activity test any(change[b,-1,ids]) (ddpg.py:179-184).
 * `change` lives in the extra block at float offset off_change; active[b*ntasks+j] in {0,1}. */
int curious_episode_activity(const float* staging, const curious_layout_t* L, const curious_tasks_t* tasks,
                             int32_t off_change, int32_t n_episodes, int32_t* active, curious_stream_t stream);

/* Normalizer.update on `dim` columns of a row matrix: local_sum += sum(v), local_sumsq += sum(v*v),
 * local_count += rows; float64 column sums folded into the float32 accumulators like NumPy does.
 * Replaces normalizer.py:64-70.  acc = [local_sum[dim] | local_sumsq[dim] | local_count[1]]. */
int curious_norm_update(const float* rows, int32_t n_rows, int32_t stride, int32_t col_off, int32_t dim,
                        float* acc, double* scratch, curious_stream_t stream);
/* scratch floats needed by curious_norm_update (in doubles) */
int64_t curious_norm_scratch_doubles(int32_t n_rows, int32_t dim);

/* recompute_stats after the cross-rank SUM of acc: synced = acc / world_size (mean over ranks);
 * count += synced_count, sum += ..., sumsq += ...; mean = sum/count;
 * std = sqrt(max(eps^2, sumsq/count - (sum/count)^2)); acc is zeroed.
 * Replaces normalizer.py:50-61,84-118.  state = [sum[dim] | sumsq[dim] | count[1] | mean[dim] | std[dim]]. */
int curious_norm_recompute(float* acc, float* state, int32_t dim, float world_size, float eps,
                           curious_stream_t stream);

/* Normalizer.update of TWO normalisers fed from one row matrix (DDPG.store_episode: o_stats and g_stats from the
 * same HER-sampled batch, ddpg.py:216-223) in two launches: partial column sums over [a | b], then a finishing launch
 * (wavefront reductions, the accumulator update of normalizer.py:68-70).  With state_a / state_b given (single rank)
 * the finishing launch also runs recompute_stats (normalizer.py:96-118, world size 1) and zeroes the accumulators;
 * with several ranks pass NULL, all-reduce the accumulators and call curious_norm_recompute.
 * skip (device, optional): when *skip != 0 -- the NaN word of the rollout flags -- nothing is accumulated (the reference
 * discards a NaN rollout before store_episode ever sees it, rollout.py:268-271).  dim_a + dim_b <= 256. */
int64_t curious_norm_pair_scratch_doubles(int32_t n_rows, int32_t dim_a, int32_t dim_b);
int curious_norm_update_pair(const float* rows, int32_t n_rows, int32_t stride, int32_t off_a, int32_t dim_a,
                             int32_t off_b, int32_t dim_b, float* acc_a, float* acc_b, float* state_a, float* state_b,
                             float eps_a, float eps_b, double* scratch, const float* skip /* may be NULL */,
                             curious_stream_t stream);

/* ---- networks ---------------------------------------------------------------------------------
 * Parameter vector of one agent = [theta_Q | pad | theta_pi | pad] (ddpg.py:456 main_vars order); theta_pi
 * starts at curious_param_offset_pi() (P_Q rounded up to 64 floats), the vector is curious_param_total()
 * floats long, pads are zero and stay zero under Adam / Polyak.  Gradients, Adam moments and the target
 * vector use the same layout.  Each network is in TF creation
 * order (util.py:79-101): _0_state/kernel[in,H], _0_state/bias[H], _0_goal/kernel[G,H], then
 * (kernel[H,H], bias[H]) x (layers-1), _3/kernel[H,out], _3/bias[out]; kernels [in,out] row-major.
 * modular=0 is the flat ActorCritic (util.py:56-71): _0/kernel[(O+G(+U)),H], _0/bias, ... */
typedef struct curious_net_cfg {
  int32_t dimo, dimg, dimu, dimtd, hidden, layers, modular;
  float max_u, gamma, clip_return, action_l2;
  int32_t clip_pos_returns;
  int32_t normalize_obs;       /* actor_critic.py:76-83 (default off, train.py:351) */
  float norm_clip;             /* normalizer.py:72-77 default_clip_range */
  int32_t loss_rows;           /* 0 (or B): the losses of curious_ddpg_grads / _update are means over the whole batch
                                * (ddpg.py:439-441).  L > 0 with B % L == 0: the batch holds the minibatches of B / L
                                * (virtual) RANKS, L consecutive rows each; every rank's losses are means over ITS rows and
                                * the gradient is the SUM over the ranks of each rank's gradient -- what MpiAdam.update's
                                * Allreduce(SUM) makes of B / L processes with a batch of L rows each (mpi_adam.py:26-28,
                                * ddpg.py:452-453).  out_losses then holds B / L pairs [Q_loss, pi_loss], rank by rank */
} curious_net_cfg_t;

int64_t curious_param_count_Q(const curious_net_cfg_t* cfg);
int64_t curious_param_count_pi(const curious_net_cfg_t* cfg);
int64_t curious_param_offset_pi(const curious_net_cfg_t* cfg);
int64_t curious_param_total(const curious_net_cfg_t* cfg);
/* workspace floats for curious_ddpg_grads / curious_policy_forward at batch size B */
int64_t curious_workspace_floats(const curious_net_cfg_t* cfg, int32_t B);
/* Float offset, inside that workspace, of its FAULT WORD (one int32, the first of a block of 64 words that is zeroed
 * together with the workspace).  The
 * row-local update hands Q' of the target networks from one workgroup to another inside one launch; a consumer that
 * never receives its value (HIP does not promise the dispatch order the hand-off relies on) gives up after "qt_spins"
 * polls, turns the loss into NaN and increments this word.  While it is non-zero every optimiser of this library that
 * knows the workspace (curious_ddpg_update*, curious_adam_update* given curious_ddpg_transposed()) leaves theta, m, v
 * and the transposed copies untouched: the caller reads the word once per cycle, raises, and clears it. */
int64_t curious_workspace_fault_offset(const curious_net_cfg_t* cfg, int32_t B);
/* Lab only (option "lab_dw_stamps", tools/dw_stamps.py): float offset of the area of the workspace -- unused on the
 * row-local route -- into which the weight-gradient / optimiser launch writes 8 64-bit cycle stamps per block. */
int64_t curious_workspace_stamps_offset(const curious_net_cfg_t* cfg, int32_t B);

/* Where the device-drawn HER gather of the NEXT update's batch goes (curious_her_sample with `rng`) when it rides along
 * with another call: replay storage + layouts + sampler description + the staging tensor to fill. */
typedef struct curious_next_batch {
  const float* storage; int64_t buf_stride;
  const curious_layout_t* L; const curious_tasks_t* tasks; const curious_sample_params_t* P;
  const curious_sample_rng_t* rng;
  float* batch;
} curious_next_batch_t;

/* One DDPG._grads(): target/main forward, losses, flat gradients (ddpg.py:235-243,419-449).
 * batch rows as written by curious_her_sample.  o_stats/g_stats = normaliser state vectors (may be
 * NULL when normalize_obs = 0).  out_losses = [Q_loss, pi_loss]; out_Q_pi[B] = main.Q_pi_tf (the
 * "actor_loss" returned by DDPG.train, ddpg.py:237-243); grad = [Q_grad | pad | pi_grad | pad] (pads untouched).
 * If step_ctr != NULL, *step_ctr is incremented once (device-side step counter for RNG / Adam).
 * params_unchanged != 0: since the previous call on THIS workspace (same cfg and B) nothing has written theta_main
 * except an optimiser call that was given curious_ddpg_transposed() of this workspace (`keep`), and the workspace was
 * left alone -- the library then trusts the transposed weight copies it keeps there (0: always safe, +1 launch).
 * next (may be NULL; multi-rank training, where the optimiser is a launch of its own behind the all-reduce): the gather
 * of the NEXT update's batch (ddpg.py:251-360, device-drawn plan keyed by THIS call's step_ctr: next->rng->step_ctr ==
 * step_ctr) is part of this call -- on the row-local route it runs in spare workgroups of the gradient launch, hidden
 * behind the layer chains; batches of the 16-row form (>= 1 280 rows: several virtual ranks) have it in extra blocks of
 * the weight-gradient launch instead (elsewhere: a launch behind the gradients).  Same batch as curious_her_sample called
 * right after this call; next->batch must not alias `batch`. */
int curious_ddpg_grads(const curious_net_cfg_t* cfg, const float* theta_main, const float* theta_target,
                       const float* batch, const curious_batch_layout_t* BL, int32_t B,
                       const float* o_stats, const float* g_stats, float* workspace, float* grad,
                       float* out_losses, float* out_Q_pi, int64_t* step_ctr, int32_t params_unchanged,
                       const curious_next_batch_t* next, curious_stream_t stream);

/* Actor (and optionally critic) forward for acting: pi = max_u*tanh(net(o,td,g)), Q = critic(o,td,pi,g)
 * (ddpg.py:129-146, actor_critic.py:87-94).  Inputs are separate row matrices with their strides;
 * o and g are clipped to +-clip_obs first (ddpg.py:118-127).  out_Q may be NULL.
 * Option "fwd16" (curious_set_option; default 0): calls with n >= 1 024, n % 16 == 0 take 16 rows per workgroup -- 2.6 x
 * the rows per second, the sums over the hidden units in another order (results equal to ~1e-6, not bit for bit with the
 * fused acting entry points below, which share the default form's bits). */
int curious_policy_forward(const curious_net_cfg_t* cfg, const float* theta, const float* o, int32_t ldo,
                           const float* ag, int32_t ldag, const float* g, int32_t ldg, const float* td,
                           int32_t ldtd, int32_t n, float clip_obs, int32_t relative_goals,
                           const float* o_stats, const float* g_stats, float* workspace, float* out_pi,
                           float* out_Q, curious_stream_t stream);

/* Action post-processing (ddpg.py:149-152), NumPy promotion kept: u = f32(f64(u) + noise_scale*randn);
 * u = clip(u, +-max_u); u = f32(f64(u) + binom*(unif - f64(u))).  noise_scale = noise_eps*max_u (double).
 * Parity mode: randn[n*dimu], binom[n], unif[n*dimu] (float64, drawn on the host in that order; unif is
 * already uniform(-max_u, max_u)).  Throughput mode (all three NULL): Philox keyed by (seed, counter). */
int curious_action_noise(float* u, int32_t ldu, int32_t n, int32_t dimu, double noise_scale, double random_eps,
                         double max_u, const double* randn, const double* binom, const double* unif,
                         uint64_t seed, uint64_t counter, curious_stream_t stream);

/* MpiAdam.update after the all-reduce (mpi_adam.py:29-35) on the fused [theta_Q | theta_pi] vector;
 * the two optimisers keep separate step sizes a_Q / a_pi = lr*sqrt(1-b2^t)/(1-b1^t), rounded to float32.
 * alpha_tab (device, [tab_len][2]) is a ring indexed by (*step_ctr - 1 - tab_base) mod tab_len when non-NULL, else
 * alpha_host is used. */
/* The transposed copies of square parameter matrices that the row-local gradient pass keeps in its workspace
 * (curious_ddpg_grads / curious_ddpg_update*; mlp_rows.h): matrix i is the dim x dim block at parameter index
 * src_off[i] of theta, its transpose lives at dst[i].  An optimiser call that is handed this description (`keep`)
 * writes WT next to the W it updates, so the next gradient pass may be told params_unchanged. */
typedef struct curious_transposed {
  int32_t n;                           /* number of matrices, 0 = none kept for this (cfg, B) */
  int32_t dim;
  int64_t src_off[8];
  float* dst[8];
  const int32_t* fault;                /* the workspace's fault word (curious_workspace_fault_offset); set even when n == 0 */
  int64_t fault_flag;                  /* 1 + index, in the GRADIENT vector, of the collective fault flag (0 = none): a
                                        * padding element of the fused layout (the last one in front of theta_pi) that
                                        * curious_ddpg_grads* sets to 1.0 when the workspace's fault word is non-zero, 0.0
                                        * otherwise.  It travels through the ranks' gradient all-reduce (SUM) with the
                                        * gradients, so an optimiser call given this description skips on EVERY rank when
                                        * ANY rank's hand-off failed (and raises its own fault word: the freeze is sticky
                                        * everywhere) -- the replicas stay identical (mpi_adam.py:42-50) */
} curious_transposed_t;
int curious_ddpg_transposed(const curious_net_cfg_t* cfg, int32_t B, float* workspace, curious_transposed_t* out);

int curious_adam_update(float* theta, float* m, float* v, const float* grad, int64_t n_Q, int64_t n_pi,
                        const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base, int32_t tab_len,
                        const float* alpha_host, float beta1, float one_minus_beta1, float beta2,
                        float one_minus_beta2, float epsilon, const curious_transposed_t* keep /* may be NULL */,
                        curious_stream_t stream);

/* curious_adam_update + the device-drawn HER gather of the NEXT update (curious_her_sample with `rng`) in ONE launch:
 * the gather does not depend on the parameters, so it rides along with the optimiser instead of heading the next
 * update as a dependent launch.  The caller resamples explicitly after anything that changes the buffers
 * (DDPG.store_episode), which keeps "sample after store" (train.py:150-154). */
int curious_adam_update_and_sample(float* theta, float* m, float* v, const float* grad, int64_t n_Q, int64_t n_pi,
                                   const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base, int32_t tab_len,
                                   const float* alpha_host, float beta1, float one_minus_beta1, float beta2,
                                   float one_minus_beta2, float epsilon, const float* storage, int64_t buf_stride,
                                   const curious_layout_t* L, const curious_tasks_t* tasks,
                                   const curious_sample_params_t* P, const curious_sample_rng_t* rng, int32_t n,
                                   float* batch, const curious_batch_layout_t* BL,
                                   const curious_transposed_t* keep /* may be NULL */, curious_stream_t stream);

/* One whole single-rank update: DDPG._grads + both MpiAdam.update calls (ddpg.py:235-248, mpi_adam.py:29-35 with no
 * ranks to reduce over) with the optimiser applied in the epilogue of the weight-gradient launch, and optionally the
 * HER gather of the NEXT update's batch (ddpg.py:251-360, device-drawn plan) riding on that launch.  Results (theta,
 * m, v, grad, losses, Q_pi, next batch) are bit-identical to curious_ddpg_grads followed by
 * curious_adam_update_and_sample.  `next->batch` must not alias `batch`.  Multi-rank training cannot use it: the
 * gradient all-reduce sits between the two halves (mpi_adam.py:26-28). */
typedef struct curious_adam_state {
  float* m; float* v;                  /* [curious_param_total] moments, same layout as theta */
  const float* alpha_tab;              /* [tab_len][2] ring of (Q, pi) step sizes indexed by *step_ctr, or NULL */
  int64_t tab_base;
  int32_t tab_len;
  float alpha_Q, alpha_pi;             /* step sizes when alpha_tab == NULL */
  float beta1, one_minus_beta1, beta2, one_minus_beta2, epsilon;
  int32_t params_unchanged;            /* non-zero: since the previous curious_ddpg_update* call on THIS workspace returned,
                                        * nothing else has written theta_main (and the workspace was left alone).  The
                                        * library keeps transposed copies of the main networks' hidden matrices in the
                                        * workspace for its backward layers; the optimiser epilogue keeps them current,
                                        * and with 0 (always safe, +1 launch) they are rebuilt from theta_main first. */
} curious_adam_state_t;

int curious_ddpg_update(const curious_net_cfg_t* cfg, float* theta_main, const float* theta_target,
                        const float* batch, const curious_batch_layout_t* BL, int32_t B,
                        const float* o_stats, const float* g_stats, float* workspace, float* grad,
                        float* out_losses, float* out_Q_pi, int64_t* step_ctr,
                        const curious_adam_state_t* adam, const curious_next_batch_t* next, curious_stream_t stream);

/* Batched task_experts update (BASELINE configs[4]): curious_ddpg_update for n_experts agents of identical shape in
 * ONE launch sequence (2 launches on the row-local route, grid.z / grid.y carry the expert).  The reference keeps one DDPG per task on shared
 * buffers and trains them one after the other (train.py:65-121; sampling rule ddpg.py:302-318,335).  Every per-expert
 * array -- theta_main, theta_target, batch, workspace, out_losses, out_Q_pi, step_ctr, adam->m / v / alpha_tab,
 * next->batch and the sampling tables of next->rng (prop_prefix, cur_size, buf_alias, buf_task, step_ctr) -- is passed
 * for expert 0 and lives at the same offset of a per-expert slab; expert e's copy is expert_stride FLOATS (a multiple
 * of 64) further.  The gradient vectors have a stride of their own (grad_stride floats, a multiple of 64 >=
 * curious_param_total(): a contiguous [n_experts][grad_stride] block, see curious_ddpg_grads_experts).  Shared:
 * next->storage, the layouts, tasks, sampler parameters.  Expert e draws its batches with the
 * Philox key next->rng->seed + e * seed_stride.  Results per expert are bit-identical to curious_ddpg_update on that
 * expert alone.  Requires the fast routes (modular nets, hidden 256, 2-3 layers, dimu 4, B % 256 == 0) and fails with an
 * error otherwise (the caller then updates the experts one by one).  o_stats / g_stats (input normalisation,
 * actor_critic.py:76-83; NULL without): expert 0's normaliser state vectors -- every expert has its own
 * (train.py:285-291), expert e's live expert_stride floats further like everything else of its state. */
int curious_ddpg_update_experts(const curious_net_cfg_t* cfg, int32_t n_experts, int64_t expert_stride,
                                int64_t grad_stride, uint64_t seed_stride, float* theta_main,
                                const float* theta_target, const float* batch, const curious_batch_layout_t* BL,
                                int32_t B, const float* o_stats, const float* g_stats, float* workspace, float* grad,
                                float* out_losses, float* out_Q_pi, int64_t* step_ctr, const curious_adam_state_t* adam,
                                const curious_next_batch_t* next, curious_stream_t stream);

/* The two halves of curious_ddpg_update_experts for DATA-PARALLEL batched experts (BASELINE configs[4] on several
 * GPUs): the reference runs task_experts under MPI like everything else -- every expert's two MpiAdam instances sum
 * their gradients over the ranks (train.py:65-121, mpi_adam.py:21-35) -- so between the halves the caller all-reduces
 * (SUM) the experts' gradients.  They are kept in ONE contiguous block [n_experts][grad_stride] (`grad` = expert 0's
 * vector), which makes that a single collective over n_experts * P floats (4.7 MB at 4 experts).
 *   curious_ddpg_grads_experts: curious_ddpg_grads for every expert (row-local pass + one weight-gradient launch),
 *     with `next` (expert e's sampler key = next->rng->seed + e * seed_stride) also every expert's next batch;
 *   curious_adam_update_and_sample_experts: curious_adam_update_and_sample for every expert (grid.y = expert): Adam
 *     from the summed gradients (+ the HER gather of every expert's next batch unless storage == NULL: it was part of
 *     the gradient call); `keep` = curious_ddpg_transposed() of expert 0's workspace (copies and fault word of expert
 *     e: expert_stride floats further).
 * Per expert the results are bit-identical to curious_ddpg_grads / curious_adam_update_and_sample on that expert alone,
 * and -- with one rank -- to curious_ddpg_update_experts. */
int curious_ddpg_grads_experts(const curious_net_cfg_t* cfg, int32_t n_experts, int64_t expert_stride,
                               int64_t grad_stride, const float* theta_main, const float* theta_target,
                               const float* batch, const curious_batch_layout_t* BL, int32_t B, const float* o_stats,
                               const float* g_stats, float* workspace, float* grad, float* out_losses, float* out_Q_pi,
                               int64_t* step_ctr, int32_t params_unchanged, uint64_t seed_stride,
                               const curious_next_batch_t* next, curious_stream_t stream);
int curious_adam_update_and_sample_experts(int32_t n_experts, int64_t expert_stride, int64_t grad_stride,
                                           uint64_t seed_stride, float* theta, float* m, float* v, const float* grad,
                                           int64_t n_Q, int64_t n_pi, const float* alpha_tab, const int64_t* step_ctr,
                                           int64_t tab_base, int32_t tab_len, float beta1, float one_minus_beta1,
                                           float beta2, float one_minus_beta2, float epsilon, const float* storage,
                                           int64_t buf_stride, const curious_layout_t* L, const curious_tasks_t* tasks,
                                           const curious_sample_params_t* P, const curious_sample_rng_t* rng, int32_t n,
                                           float* batch, const curious_batch_layout_t* BL,
                                           const curious_transposed_t* keep, curious_stream_t stream);

/* The gradient all-reduce of several ranks as ONE kernel per rank over peer-mapped buffers, fused with the optimiser
 * (reduce-scatter + Adam + all-gather; csrc/ipc.hip): replaces Allreduce(SUM) + Adam of MpiAdam.update
 * (mpi_adam.py:21-35) -- in this build: the RCCL all-reduce + curious_adam_update of the several-rank update.  Every rank
 * calls it after its gradient launches with
 *   peers   every rank's gradient vector, STAGING vector (n_Q + n_pi floats: where the peers deliver the new parameter
 *           slices) and flag block ([2][CURIOUS_IPC_MAX_RANKS] uint32, zeroed once) AS MAPPED INTO THIS PROCESS
 *           (hipIpcOpenMemHandle / the own rank's local pointers).  All three live in FINE-GRAINED device memory
 *           (curious_ipc_alloc): peers read, write and poll them while kernels of the owner run;
 *   theta   the LOCAL parameter vector, ordinary device memory no peer ever touches: the kernel copies the completed
 *           staging vector into it;
 *   m, v    the local moment vectors (full length; only the rank's slice [rank * n / world, (rank + 1) * n / world) is
 *           used: each rank runs the optimiser on its slice only and writes the new slice into every rank's staging vector);
 *   step_ctr  the device step counter the gradient launches advanced: indexes the step-size ring like curious_adam_update;
 *   epoch   a local uint32 device word (zeroed once), advanced by one per call: the hand-shake token.  NOT the step
 *           counter, which a caller may rewind to replay a run of updates -- tokens never repeat;
 *   done, err  two local uint32 / int32 device words (zeroed once): block bookkeeping / "a wait gave up after `spins`
 *              polls": the rank then skips the arithmetic and the copy of that epoch like a faulted update, keeps
 *              signalling so that no peer hangs, and the caller raises;
 *   keep    curious_ddpg_transposed() of the LOCAL workspace: fault word and flag index of the collective hand-off guard
 *           (every rank reads every rank's flag element and all skip alike), and the transposed copies, rebuilt from the
 *           new parameters at the end of the kernel (the next gradient call may say params_unchanged).
 * Sums are taken in rank order 0 .. world - 1 on every rank: the replicas stay bit-identical (mpi_adam.py:42-50).
 * n_Q + n_pi must divide by world.  The kernel's 64 workgroups wait for peers: every rank must call it for every update. */
#define CURIOUS_IPC_MAX_RANKS 8
typedef struct curious_ipc_peers {
  int32_t world, rank;
  const float* grad[CURIOUS_IPC_MAX_RANKS];
  float* stage[CURIOUS_IPC_MAX_RANKS];
  uint32_t* flags[CURIOUS_IPC_MAX_RANKS];
} curious_ipc_peers_t;
int curious_allreduce_adam_ipc(const curious_ipc_peers_t* peers, float* theta, float* m, float* v, int64_t n_Q,
                               int64_t n_pi, const float* alpha_tab, const int64_t* step_ctr, int64_t tab_base,
                               int32_t tab_len, float beta1, float one_minus_beta1, float beta2, float one_minus_beta2,
                               float epsilon, uint32_t* epoch, uint32_t* done, int32_t* err, int32_t spins,
                               const curious_transposed_t* keep, curious_stream_t stream);

/* Set-up / tear-down of the peer mappings (host side, synchronous: once per job, not per update).  The vectors a rank
 * shares come from curious_ipc_alloc (hipExtMallocWithFlags(hipDeviceMallocFinegrained), zeroed; CURIOUS_IPC_COARSE=1 in
 * the environment: plain hipMalloc, for A/B), are exported as 64-byte handles (hipIpcGetMemHandle), exchanged
 * by the caller (any host-side collective) and imported by the peers (hipIpcOpenMemHandle).  Every imported pointer is
 * closed before its owner frees the block. */
int curious_ipc_alloc(int64_t bytes, void** out);
int curious_ipc_free(void* ptr);
int curious_ipc_export(const void* ptr, unsigned char* handle64);
int curious_ipc_import(const unsigned char* handle64, void** out);
int curious_ipc_close(void* ptr);

/* target <- polyak*target + one_minus_polyak*main (ddpg.py:461-462); the two factors are the float32
 * roundings of the Python doubles `polyak` and `1. - polyak`.  polyak = 0, one_minus = 1 copies (ddpg.py:459-460). */
int curious_polyak_update(float* target, const float* main_, int64_t n, float polyak, float one_minus_polyak,
                          curious_stream_t stream);

/* Order-independent 64-bit checksum of the parameter bit patterns, for MpiAdam.check_synced
 * (mpi_adam.py:42-50) without broadcasting the vector.  out[2] = {sum, xor} of a per-element hash. */
int curious_param_checksum(const float* theta, int64_t n, uint64_t* out, curious_stream_t stream);

/* ---- synthetic batched environment (stand-in for gym_flowers MultiTaskFetchArm, rollout.py:107-143,
 * 256-284).  state/ record layout: see DESIGN.md "Synthetic env". ---------------------------------- */
typedef struct curious_env_cfg {
  int32_t ntasks, dimo, T;
  int32_t wrap;   /* 0: env i of a batch is env env_id0 + i.  > 0 (ABI 10; it sits in what was padding): a batch of SLOTS -- slot
                   * i is env env_id0 + i % wrap at an episode of its own (episode[i]): the n_test_rollouts evaluation rollouts
                   * of train.py:156-158, rollout k of env e = slot k * wrap + e, stepped by one launch instead of one per rollout */
  uint64_t seed;
} curious_env_cfg_t;

/* Reset n envs: o[n][dimo] from Philox stream (env_id0+i, episode[i]); episode[i] (episodes started so far) is
 * incremented afterwards, on the device; writes record row 0 of the
 * staging block (o, ag, g, td) and the working arrays o/ag/g/td. goals_raw[n][3] in [-1,1], tasks[n].
 * flags (optional, [n+1] floats, see curious_env_step): flags[n], the NaN word of the coming rollout, is cleared. */
int curious_env_reset(const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                      int32_t* episode, const int32_t* tasks, const float* goals_raw, int32_t n,
                      float* o, float* ag, float* g, float* td, float* staging, float* flags /* may be NULL */,
                      curious_stream_t stream);

/* curious_env_reset + curious_counter_add(counter, delta) in the one launch (counter may be NULL): the reset that heads
 * a captured rollout also advances the device-resident base of the policy's noise counter -- the rollout that follows
 * passes `counter = 1 - delta` so that it draws from base_before + 1 ... as if the add came behind it. */
int curious_env_reset_count(const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                            int32_t* episode, const int32_t* tasks, const float* goals_raw, int32_t n,
                            float* o, float* ag, float* g, float* td, float* staging, float* flags /* may be NULL */,
                            int64_t* counter, int64_t delta, curious_stream_t stream);

/* One step of n envs with actions u[n][dimu]: updates o/ag in place, writes u, g, td, change, is_success
 * into staging row t and o, ag into row t+1 (the episode record of rollout.py:273-303).
 * flags (optional, [n+1] floats): at the last step t = T-1, flags[i] = is_success of env i and flags[n] = 1 if any
 * observation is NaN (flags[n] is cleared by curious_env_reset) -- what rollout.py:268-271,306 reads, in one small D2H
 * copy.  dimo <= 128. */
int curious_env_step(const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                     const int32_t* episode, const int32_t* tasks, const float* u, int32_t ldu, int32_t t,
                     int32_t n, float* o, float* ag, const float* g, const float* td, float* staging,
                     int32_t off_change, int32_t off_success, double reward_eps, float* flags,
                     curious_stream_t stream);

/* *p += delta on the device, stream ordered (the noise-counter base of a captured rollout advances inside the graph). */
int curious_counter_add(int64_t* p, int64_t delta, curious_stream_t stream);

/* Fused acting step of the batched rollout: actor forward on the envs' current (o, g, td), output layer + noise +
 * clip + eps-greedy (device Philox, as curious_action_noise in throughput mode) and ONE env step, i.e.
 * policy.get_actions + env.step of rollout.py:226-263 for all n envs in ONE launch on the row-local route (n % 4 == 0,
 * hidden 256; 3-4 launches otherwise).  u_out receives the
 * actions.  Same results, bit for bit, as curious_policy_forward + curious_action_noise + curious_env_step.
 * The Philox noise counter is counter + *counter_base (counter_base: optional device int64, so that a T-step rollout
 * captured once in a hipGraph draws fresh noise on every replay). */
int curious_policy_act_env_step(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                const curious_env_cfg_t* E,
                                const curious_layout_t* L, int32_t env_id0, const int32_t* episode,
                                const int32_t* tasks, int32_t t, float* o, float* ag, const float* g, const float* td,
                                float* staging, int32_t off_change, int32_t off_success, double reward_eps,
                                float* flags, curious_stream_t stream);

/* The acting loop of a batched rollout (rollout.py:226-303 for every env): steps t0 .. t0 + nsteps - 1, i.e. nsteps x
 * curious_policy_act_env_step with noise counters counter, counter + 1, ... -- same results bit for bit.  The envs do
 * not depend on each other, so on the row-local route (n % 4 == 0, hidden 256) the whole loop is ONE launch: a
 * workgroup walks its 4 envs through all steps, the new observation goes from the env step straight into the
 * policy's input row in LDS.  Other shapes: the launches of the single-step entry point, nsteps times.
 * With nsteps >= 4, 2-3 layers and n <= the device's CU count (option "resident") the 4 envs get FOUR workgroups that
 * keep a quarter of every hidden matrix each in LDS for the whole launch and exchange activations through the first
 * n * 1024 floats of `workspace` -- which must then not be shared with a launch running concurrently, and whose
 * (seed, counter) pairs must not repeat (they tag the exchanged words).  The members wait for each other: should one
 * never be scheduled the others give up after "res_spins" polls and flags[n] is set to 2 (the rollout is void).  The
 * route assumes that the process has the device to itself (one process per GPU); several processes sharing one device
 * set option "resident" = 0. */
int curious_policy_rollout(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                           float* workspace, double noise_scale, double random_eps, uint64_t seed, uint64_t counter,
                           const int64_t* counter_base, float* u_out, int32_t ldu, const curious_env_cfg_t* E,
                           const curious_layout_t* L, int32_t env_id0, const int32_t* episode, const int32_t* tasks,
                           int32_t t0, int32_t nsteps, float* o, float* ag, const float* g, const float* td,
                           float* staging, int32_t off_change, int32_t off_success, double reward_eps, float* flags,
                           curious_stream_t stream);

/* The two entry points above for networks with input normalisation (actor_critic.py:76-83, --normalize_obs): o_stats /
 * g_stats = the state vectors of the observation / goal normalisers (layout as in curious_norm_recompute: mean at
 * 2 dim + 1, std at 3 dim + 1), applied to the clipped observation and goal on their way into the policy -- at the start
 * of the launch and, inside it, to every new observation the env step hands to the next acting step.  NULL statistics =
 * the entry points above.  relative_goals != 0 (ddpg.py:118-127): the policy sees g - ag, recomputed from the new
 * achieved goal after every env step (streaming and weights-resident kernel alike: every member of a resident group holds
 * the whole input row and runs the env step itself). */
int curious_policy_act_env_step_stats(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                      float* workspace, double noise_scale, double random_eps, uint64_t seed,
                                      uint64_t counter, const int64_t* counter_base, float* u_out, int32_t ldu,
                                      const curious_env_cfg_t* E, const curious_layout_t* L, int32_t env_id0,
                                      const int32_t* episode, const int32_t* tasks, int32_t t, float* o, float* ag,
                                      const float* g, const float* td, float* staging, int32_t off_change,
                                      int32_t off_success, double reward_eps, float* flags, int32_t relative_goals,
                                      const float* o_stats, const float* g_stats, curious_stream_t stream);
int curious_policy_rollout_stats(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                 float* workspace, double noise_scale, double random_eps, uint64_t seed, uint64_t counter,
                                 const int64_t* counter_base, float* u_out, int32_t ldu, const curious_env_cfg_t* E,
                                 const curious_layout_t* L, int32_t env_id0, const int32_t* episode,
                                 const int32_t* tasks, int32_t t0, int32_t nsteps, float* o, float* ag, const float* g,
                                 const float* td, float* staging, int32_t off_change, int32_t off_success,
                                 double reward_eps, float* flags, int32_t relative_goals, const float* o_stats,
                                 const float* g_stats, curious_stream_t stream);

/* curious_policy_rollout_stats for the envs of several VIRTUAL RANKS in one launch.  The reference runs one process per
 * rank, each with its own envs, its own exploration-noise stream (train.py:242-243) and its own decision whether a rollout
 * exploits (rollout.py:183-189: noise_eps = random_eps = 0 for that rank's envs).  Here the n envs are consecutive groups of
 * `group` envs, one group per rank: env i belongs to group i / group; its noise is drawn with the Philox key
 * seed + (i / group) * seed_stride at row index i % group -- exactly what a launch of its own with that key and n = group
 * would draw --, and exploit[i / group] != 0 (device int32 array of ceil(n / group) entries, may be NULL) switches the
 * noise of the whole group off.  groups == NULL or group == 0: curious_policy_rollout_stats. */
typedef struct curious_rank_groups {
  int32_t group;
  int32_t reserved;
  uint64_t seed_stride;
  const int32_t* exploit;
} curious_rank_groups_t;
int curious_policy_rollout_ranks(const curious_net_cfg_t* cfg, const float* theta, int32_t n, float clip_obs,
                                 float* workspace, double noise_scale, double random_eps, uint64_t seed, uint64_t counter,
                                 const int64_t* counter_base, float* u_out, int32_t ldu, const curious_env_cfg_t* E,
                                 const curious_layout_t* L, int32_t env_id0, const int32_t* episode,
                                 const int32_t* tasks, int32_t t0, int32_t nsteps, float* o, float* ag, const float* g,
                                 const float* td, float* staging, int32_t off_change, int32_t off_success,
                                 double reward_eps, float* flags, int32_t relative_goals, const float* o_stats,
                                 const float* g_stats, const curious_rank_groups_t* groups, curious_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CURIOUS_HIP_H */
