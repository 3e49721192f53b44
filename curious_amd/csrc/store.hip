// K11: episode store (staging block -> replay slots) and the per-episode task activity test.
//
// Replaces (reference): ReplayBuffer.store_episode replay_buffer.py:57-72 and the routing loop of
// DDPG.store_episode ddpg.py:178-197.  An episode record is (T+1)*row_stride contiguous floats, so a store is
// a flat, fully coalesced 16-byte-per-lane copy; slot selection (replay_buffer.py:90-109, NumPy stream)
// stays on the host.
#include "common.h"

__global__ __launch_bounds__(256) void store_episodes_kernel(float* __restrict__ storage,
                                                            const float* __restrict__ staging,
                                                            const int32_t* __restrict__ pair_src,
                                                            const int64_t* __restrict__ pair_dst, int64_t rec_floats,
                                                            int32_t vec_ok) {
  const int pair = blockIdx.y;
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
                     pair_src, pair_dst, rec, vec_ok); }
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
