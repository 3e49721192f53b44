// K11: episode store (staging block -> replay slots) and the per-episode task activity test.
//
// Replaces (reference): ReplayBuffer.store_episode replay_buffer.py:57-72 and the routing loop of
// DDPG.store_episode ddpg.py:178-197.  An episode record is (T+1)*row_stride contiguous floats, so a store is
// a flat, fully coalesced 16-byte-per-lane copy; slot selection (replay_buffer.py:90-109, NumPy stream)
// stays on the host.
#include "common.h"
#include <algorithm>

// n_pairs_dev != NULL: the pair list was made on the device (route_episodes_kernel); the grid covers the largest
// possible list and the blocks beyond the real one leave
__global__ __launch_bounds__(256) void store_episodes_kernel(float* __restrict__ storage,
                                                            const float* __restrict__ staging,
                                                            const int32_t* __restrict__ pair_src,
                                                            const int64_t* __restrict__ pair_dst, int64_t rec_floats,
                                                            int32_t vec_ok, const int32_t* __restrict__ n_pairs_dev,
                                                            int32_t seg) {
  // (seg: pairs per segment of a list made rank by rank -- segment s holds n_pairs_dev[s] pairs; one segment otherwise)
  const int pair = blockIdx.y;
  if (n_pairs_dev && (pair % seg) >= n_pairs_dev[pair / seg]) return;
  if (pair_src[pair] < 0) return;                            // a later episode of the batch drew the same slot
  const float* src = staging + (int64_t)pair_src[pair] * rec_floats;
  float* dst = storage + pair_dst[pair] * rec_floats;
  if (vec_ok) {
    const int64_t n4 = rec_floats >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
      d4[i] = s4[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rec_floats;
         i += (int64_t)gridDim.x * blockDim.x)
      dst[i] = src[i];
  }
}

extern "C" int curious_store_episodes(float* storage, const float* staging, const curious_layout_t* L,
                                      const int32_t* pair_src, const int64_t* pair_dst, int32_t n_pairs,
                                      curious_stream_t stream) {
  CURIOUS_CHECK(storage && staging && L && pair_src && pair_dst, "curious_store_episodes: NULL argument");
  if (n_pairs <= 0) return 0;
  int64_t rec = (int64_t)(L->T + 1) * L->row_stride;
  int vec_ok = (rec % 4 == 0) && (((uintptr_t)storage | (uintptr_t)staging) % 16 == 0);
  int64_t work = vec_ok ? rec / 4 : rec;
  int bx = (int)((work + 255) / 256);
  if (bx > 16) bx = 16;
  if (bx < 1) bx = 1;
  { ProfScope ps__(CK_STORE, as_stream(stream)); hipLaunchKernelGGL(store_episodes_kernel, dim3(bx, n_pairs), dim3(256), 0, as_stream(stream), storage, staging,
                     pair_src, pair_dst, rec, vec_ok, (const int32_t*)nullptr, 1 << 30); }
  CURIOUS_LAUNCH_CHECK("store_episodes_kernel");
  return 0;
}

// Slot of a stored episode once its buffer is full (replay_buffer.py:101-102: np.random.randint(0, size)): in device
// RNG mode a Philox draw keyed by (seed, store call, episode, task), the same on the host (curious_store_slots_host)
// and on the device.
#define STREAM_STORE_SLOT 31u
__host__ __device__ inline int64_t store_random_slot(uint64_t seed, uint64_t call, int b, int j, int64_t size) {
  const Philox4 r = philox4x32((uint32_t)b, (uint32_t)j, (uint32_t)call, STREAM_STORE_SLOT, (uint32_t)seed,
                               (uint32_t)(seed >> 32));
  return (int64_t)(((uint64_t)r.x * (uint64_t)size) >> 32);
}

// Routing of DDPG.store_episode (ddpg.py:178-197) on the device: one block; task by task, the episodes whose activity
// flag is set get slots in ascending episode order (the order of the reference's loop): consecutive slots behind the
// buffer's current size while it has room (replay_buffer.py:94-95), a random slot each once it is full
// (replay_buffer.py:101-102); when two episodes of the batch end up on one slot the later one wins (sequential
// semantics) and the earlier pair is marked dead (src = -1).  Writes the (src, dst) pair list, its length, and the new
// sizes into the sampler's table.
// Virtual ranks (grid.x = rank): block v routes ITS n_episodes episodes -- staging records v * n_episodes .. -- into ITS
// buffers (size / alias tables tab_stride int32 further per rank, Philox key seed + v * seed_stride, pair-list segment and
// count of its own), exactly as a launch of its own would.
#define ROUTE_MAX_EPISODES 2048
#define ROUTE_HASH 4096u                                     // >= 2 x ROUTE_MAX_EPISODES, a power of two
__device__ inline unsigned route_hash(int slot) { return ((unsigned)slot * 2654435761u >> 16) & (ROUTE_HASH - 1u); }
__global__ __launch_bounds__(256) void route_episodes_kernel(const int32_t* active, int32_t ntasks,
                                                            int32_t n_route, int32_t n_episodes,
                                                            int32_t* __restrict__ cur_size,
                                                            const int32_t* __restrict__ buf_alias, int64_t capacity,
                                                            uint64_t seed, uint64_t call,
                                                            const float* __restrict__ skip,
                                                            int32_t* __restrict__ pair_src,
                                                            int64_t* __restrict__ pair_dst,
                                                            int32_t* __restrict__ n_pairs, int64_t tab_stride,
                                                            uint64_t seed_stride, const float* __restrict__ staging,
                                                            curious_layout_t L, curious_tasks_t TK, int32_t off_change,
                                                            int32_t* active_out) {
  const int vr = blockIdx.x;
  const int ep0 = vr * n_episodes;                           // this rank's first record in the staging block
  if (active_out) {
    // the task-activity test of ddpg.py:179-184 (episode_activity_kernel) done here: one launch less per cycle.  The flags
    // are written out as well -- the host mirrors the buffer sizes from them a cycle later (DDPG.settle)
    int32_t* mine = active_out + (int64_t)ep0 * ntasks;
    for (int i = threadIdx.x; i < n_episodes * ntasks; i += 256) {
      const int b = i / ntasks, j = i - b * ntasks;
      const float* row = staging + ((int64_t)(ep0 + b) * (L.T + 1) + (L.T - 1)) * L.row_stride + off_change;
      int any = 0;
      for (int k = 0; k < TK.len[j]; ++k) any |= (row[TK.ag_id[j][k]] != 0.0f);
      mine[i] = any;
    }
    __syncthreads();
  }
  active += (int64_t)ep0 * ntasks;
  cur_size += (int64_t)vr * tab_stride;
  buf_alias += (int64_t)vr * tab_stride;
  seed += (uint64_t)vr * seed_stride;
  pair_src += (int64_t)ep0 * n_route;
  pair_dst += (int64_t)ep0 * n_route;
  n_pairs += vr;
  __shared__ int wave_cnt[4];
  __shared__ int slot_of[ROUTE_MAX_EPISODES];                // this task's slot per episode, -1: not routed
  __shared__ int rank_of[ROUTE_MAX_EPISODES];                // its position in this task's part of the pair list
  __shared__ int hkey[ROUTE_HASH], hval[ROUTE_HASH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (skip && *skip != 0.0f) {                               // the rollout produced a NaN observation: keep nothing
    if (tid == 0) *n_pairs = 0;
    return;
  }
  int out = 0;                                               // pairs written so far (uniform)
  for (int j = 0; j < n_route; ++j) {
    const int cur0 = cur_size[1 + j];                        // logical buffer 1 + j belongs to task j (ddpg.py:185)
    const int64_t pool = (int64_t)buf_alias[1 + j] * capacity;
    const int out0 = out;
    int routed = 0;
    for (int b0 = 0; b0 < n_episodes; b0 += 256) {
      const int b = b0 + tid;
      const bool on = b < n_episodes && active[(int64_t)b * ntasks + j] != 0;
      const unsigned long long m = __ballot(on);
      const int before = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int off = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        off += (w < wave) ? wave_cnt[w] : 0;
        total += wave_cnt[w];
      }
      if (b < n_episodes) {
        int slot = -1;
        if (on) {
          const int rank = routed + off + before;
          slot = (cur0 + rank < capacity) ? cur0 + rank : (int)store_random_slot(seed, call, b, j, capacity);
          pair_src[out0 + rank] = ep0 + b;
          pair_dst[out0 + rank] = pool + slot;
          rank_of[b] = rank;
        }
        slot_of[b] = slot;
      }
      routed += total;
      __syncthreads();
    }
    if (cur0 + routed > capacity) {
      // random slots were drawn: an episode loses its slot to any later episode of the batch on the same slot.  The
      // last episode per slot is found through an open-addressed table in LDS (key: slot, value: max episode index),
      // O(episodes) instead of comparing every pair.
      for (int h = tid; h < ROUTE_HASH; h += 256) {
        hkey[h] = -1;
        hval[h] = -1;
      }
      __syncthreads();
      for (int b = tid; b < n_episodes; b += 256) {
        const int mine = slot_of[b];
        if (mine < 0) continue;
        unsigned h = route_hash(mine);
        for (;;) {
          const int prev = atomicCAS(&hkey[h], -1, mine);
          if (prev == -1 || prev == mine) break;
          h = (h + 1u) & (ROUTE_HASH - 1u);
        }
        atomicMax(&hval[h], b);
      }
      __syncthreads();
      for (int b = tid; b < n_episodes; b += 256) {
        const int mine = slot_of[b];
        if (mine < 0) continue;
        unsigned h = route_hash(mine);
        while (hkey[h] != mine) h = (h + 1u) & (ROUTE_HASH - 1u);
        if (hval[h] > b) pair_src[out0 + rank_of[b]] = -1;
      }
      __syncthreads();
    }
    out += routed;
    if (tid == 0) cur_size[1 + j] = (cur0 + routed < capacity) ? cur0 + routed : (int)capacity;
  }
  if (tid == 0) *n_pairs = out;
}

// the same slots for the host-routed form of a store in device RNG mode: out[i] for episode episodes[i] of task j
extern "C" int curious_store_slots_host(uint64_t seed, uint64_t call, int32_t task, int64_t size, int32_t n,
                                        const int32_t* episodes, int64_t* out) {
  CURIOUS_CHECK(size > 0 && (n == 0 || (episodes && out)), "curious_store_slots_host: bad argument");
  for (int i = 0; i < n; ++i) out[i] = store_random_slot(seed, call, episodes[i], task, size);
  return 0;
}

static int route_store_ranks(float* storage, const float* staging, const curious_layout_t* L, const int32_t* active,
                             int32_t ntasks, int32_t n_route, int32_t n_episodes, int32_t n_ranks, int32_t* cur_size,
                             const int32_t* buf_alias, int64_t tab_stride, int64_t capacity, uint64_t seed,
                             uint64_t seed_stride, uint64_t call, const float* skip, int32_t* pair_src,
                             int64_t* pair_dst, int32_t* n_pairs, curious_stream_t stream,
                             const curious_tasks_t* tasks = nullptr, int32_t off_change = 0);

extern "C" int curious_route_store_episodes(float* storage, const float* staging, const curious_layout_t* L,
                                            const int32_t* active, int32_t ntasks, int32_t n_route,
                                            int32_t n_episodes, int32_t* cur_size, const int32_t* buf_alias,
                                            int64_t capacity, uint64_t seed, uint64_t call, const float* skip,
                                            int32_t* pair_src, int64_t* pair_dst, int32_t* n_pairs,
                                            curious_stream_t stream) {
  return route_store_ranks(storage, staging, L, active, ntasks, n_route, n_episodes, 1, cur_size, buf_alias, 0, capacity,
                           seed, 0, call, skip, pair_src, pair_dst, n_pairs, stream);
}

extern "C" int curious_route_store_episodes_ranks(float* storage, const float* staging, const curious_layout_t* L,
                                                  const int32_t* active, int32_t ntasks, int32_t n_route,
                                                  int32_t n_episodes, int32_t n_ranks, int32_t* cur_size,
                                                  const int32_t* buf_alias, int64_t tab_stride, int64_t capacity,
                                                  uint64_t seed, uint64_t seed_stride, uint64_t call, const float* skip,
                                                  int32_t* pair_src, int64_t* pair_dst, int32_t* n_pairs,
                                                  curious_stream_t stream) {
  CURIOUS_CHECK(n_ranks >= 1 && n_ranks <= 4096 && tab_stride >= 0, "curious_route_store_episodes_ranks: bad rank arguments");
  return route_store_ranks(storage, staging, L, active, ntasks, n_route, n_episodes, n_ranks, cur_size, buf_alias,
                           tab_stride, capacity, seed, seed_stride, call, skip, pair_src, pair_dst, n_pairs, stream);
}

extern "C" int curious_activity_route_store_episodes(float* storage, const float* staging, const curious_layout_t* L,
                                                     const curious_tasks_t* tasks, int32_t off_change, int32_t* active,
                                                     int32_t n_route, int32_t n_episodes, int32_t n_ranks,
                                                     int32_t* cur_size, const int32_t* buf_alias, int64_t tab_stride,
                                                     int64_t capacity, uint64_t seed, uint64_t seed_stride, uint64_t call,
                                                     const float* skip, int32_t* pair_src, int64_t* pair_dst,
                                                     int32_t* n_pairs, curious_stream_t stream) {
  CURIOUS_CHECK(tasks && active, "curious_activity_route_store_episodes: NULL argument");
  CURIOUS_CHECK(n_ranks >= 1 && n_ranks <= 4096 && tab_stride >= 0, "curious_activity_route_store_episodes: bad rank arguments");
  return route_store_ranks(storage, staging, L, active, tasks->ntasks, n_route, n_episodes, n_ranks, cur_size, buf_alias,
                           tab_stride, capacity, seed, seed_stride, call, skip, pair_src, pair_dst, n_pairs, stream, tasks,
                           off_change);
}

static int route_store_ranks(float* storage, const float* staging, const curious_layout_t* L, const int32_t* active,
                             int32_t ntasks, int32_t n_route, int32_t n_episodes, int32_t n_ranks, int32_t* cur_size,
                             const int32_t* buf_alias, int64_t tab_stride, int64_t capacity, uint64_t seed,
                             uint64_t seed_stride, uint64_t call, const float* skip, int32_t* pair_src,
                             int64_t* pair_dst, int32_t* n_pairs, curious_stream_t stream, const curious_tasks_t* tasks,
                             int32_t off_change) {
  CURIOUS_CHECK(storage && staging && L && active && cur_size && buf_alias && pair_src && pair_dst && n_pairs,
                "curious_route_store_episodes: NULL argument");
  CURIOUS_CHECK(ntasks >= 1 && n_route >= 0 && n_route <= ntasks && capacity > 0 && capacity < (1ll << 31),
                "curious_route_store_episodes: bad task / capacity arguments");
  CURIOUS_CHECK(n_episodes <= ROUTE_MAX_EPISODES, "curious_route_store_episodes: at most %d episodes per call",
                ROUTE_MAX_EPISODES);
  // (the copy launch carries one (rank, episode, routed task) pair per grid.y index)
  CURIOUS_CHECK((int64_t)n_ranks * std::max(n_episodes, 0) * n_route <= 65535,
                "curious_route_store_episodes: %d ranks x %d episodes x %d routed tasks exceed the 65 535 pairs of one call",
                n_ranks, n_episodes, n_route);
  if (n_episodes <= 0 || n_route == 0) return 0;
  hipStream_t st = as_stream(stream);
  { ProfScope ps__(CK_ROUTE, st);
    curious_tasks_t tk;
    memset(&tk, 0, sizeof(tk));
    if (tasks) tk = *tasks;
    hipLaunchKernelGGL(route_episodes_kernel, dim3(n_ranks), dim3(256), 0, st, active, ntasks, n_route, n_episodes,
                       cur_size, buf_alias, capacity, seed, call, skip, pair_src, pair_dst, n_pairs, tab_stride,
                       seed_stride, staging, *L, tk, off_change, tasks ? const_cast<int32_t*>(active) : (int32_t*)nullptr); }
  CURIOUS_LAUNCH_CHECK("route_episodes_kernel");
  int64_t rec = (int64_t)(L->T + 1) * L->row_stride;
  int vec_ok = (rec % 4 == 0) && (((uintptr_t)storage | (uintptr_t)staging) % 16 == 0);
  int64_t work = vec_ok ? rec / 4 : rec;
  int bx = (int)((work + 255) / 256);
  if (bx > 16) bx = 16;
  if (bx < 1) bx = 1;
  { ProfScope ps__(CK_STORE, st);
    hipLaunchKernelGGL(store_episodes_kernel, dim3(bx, n_ranks * n_episodes * n_route), dim3(256), 0, st, storage, staging,
                       (const int32_t*)pair_src, (const int64_t*)pair_dst, rec, vec_ok, (const int32_t*)n_pairs,
                       n_episodes * n_route); }
  CURIOUS_LAUNCH_CHECK("store_episodes_kernel");
  return 0;
}

// any(change[b, -1, tasks_ag_id[j][:len(tasks_g_id[j])]])   (ddpg.py:181); change of step T-1 lives in row T-1.
__global__ void episode_activity_kernel(const float* __restrict__ staging, curious_layout_t L, curious_tasks_t T,
                                        int32_t off_change, int32_t n_episodes, int32_t* __restrict__ active) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_episodes * T.ntasks) return;
  int b = i / T.ntasks, j = i % T.ntasks;
  const float* row = staging + ((int64_t)b * (L.T + 1) + (L.T - 1)) * L.row_stride + off_change;
  int any = 0;
  for (int k = 0; k < T.len[j]; ++k) any |= (row[T.ag_id[j][k]] != 0.0f);
  active[i] = any;
}

extern "C" int curious_episode_activity(const float* staging, const curious_layout_t* L, const curious_tasks_t* tasks,
                                        int32_t off_change, int32_t n_episodes, int32_t* active,
                                        curious_stream_t stream) {
  CURIOUS_CHECK(staging && L && tasks && active, "curious_episode_activity: NULL argument");
  if (n_episodes <= 0) return 0;
  int n = n_episodes * tasks->ntasks;
  { ProfScope ps__(CK_ACTIVITY, as_stream(stream)); hipLaunchKernelGGL(episode_activity_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), staging, *L,
                     *tasks, off_change, n_episodes, active); }
  CURIOUS_LAUNCH_CHECK("episode_activity_kernel");
  return 0;
}
