// K1-K3: HER transition sampling as one coalesced, LDS-staged gather kernel.
//
// Replaces (reference): ReplayBuffer.sample replay_buffer.py:37-55, _sample_her_transitions
// her.py:99-183 (multi-task) / her.py:20-66 (flat), the multi-buffer concat+shuffle ddpg.py:326-345 and
// the clip of ddpg.py:350-353.  Index math (HER mask, future offset) is float64/int like NumPy's; goal/task
// relabel and the reward threshold are bit-exact with oracle/her.py + oracle/reward.py.
//
// Mapping: one 256-thread workgroup = 4 waves = 4 sampled transitions per pass; each wave pulls the
// transition's record row t, the (o, ag) head of row t+1 and the future achieved goal into its LDS slot with
// lane-contiguous dword loads (a record row is a few hundred contiguous bytes), relabels in LDS, and
// streams the staged batch row out lane-contiguously.  Algorithmic traffic: SURVEY 8d (1 034 B per
// transition at Arm4 dims).
#include "her_body.h"

__global__ __launch_bounds__(256) void her_sample_kernel(HerArgs a) {
  extern __shared__ float lds[];
  her_sample_body(a, blockIdx.x, lds);
}

extern "C" int curious_her_sample(const float* storage, int64_t buf_stride, const curious_layout_t* L,
                                  const curious_tasks_t* tasks, const curious_sample_params_t* P,
                                  const curious_sample_plan_t* plan, const curious_sample_rng_t* rng, int32_t n,
                                  float* batch, const curious_batch_layout_t* BL, curious_stream_t stream) {
  HerArgs a;
  if (her_fill_args(a, storage, buf_stride, L, tasks, P, plan, rng, n, batch, BL)) return -1;
  if (n == 0) return 0;
  int blocks = (n + SPB - 1) / SPB;
  { ProfScope ps__(CK_HER_SAMPLE, as_stream(stream));
    hipLaunchKernelGGL(her_sample_kernel, dim3(blocks), dim3(256), her_lds_bytes(L), as_stream(stream), a); }
  CURIOUS_LAUNCH_CHECK("her_sample_kernel");
  return 0;
}
