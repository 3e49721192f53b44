// TORCH_LIBRARY(curious_hip, ...): the dispatcher face of the C ABI ("what the C++/HIP layer must export under
// that Python surface ... device tensors in, device tensors out, errors via TORCH_CHECK, current-stream semantics").
//
// Host-only C++ (no device code): every op checks its tensors, takes the CURRENT stream of the calling context and calls
// the entry point of libcurious_hip.so that include/curious_hip.h declares -- there is no second implementation.  Built by
// curious_amd/build.py into curious_amd/lib/libcurious_torch.so (g++ against the torch headers; linked to libcurious_hip.so
// next to it) and loaded by curious_amd/torch_ops.py with torch.ops.load_library.
//
//   torch.ops.curious_hip.polyak_update(target, main, polyak)                                   ddpg.py:459-462
//   torch.ops.curious_hip.adam_update(theta, m, v, grad, n_Q, n_pi, alpha_Q, alpha_pi)          mpi_adam.py:29-35
//   torch.ops.curious_hip.param_checksum(theta) -> int64[2]                                     mpi_adam.py:42-50
//   torch.ops.curious_hip.norm_update(rows, col_off, dim, acc)                                  normalizer.py:64-70
//   torch.ops.curious_hip.norm_recompute(acc, state, world_size, eps)                           normalizer.py:96-118
//   torch.ops.curious_hip.policy_forward(cfg_i, cfg_f, theta, o, g, td, clip_obs, compute_Q) -> (pi, Q)      ddpg.py:129-146
//   torch.ops.curious_hip.ddpg_grads(cfg_i, cfg_f, theta, theta_target, batch, batch_layout, grad) -> (losses, Q_pi)
//                                                                                               ddpg.py:235-243, 419-449
// The three hot entry points take the reference's table-shaped arguments (record / batch layouts, task tables, sampler and
// env descriptions), for which the dispatcher's schema language has no type: the caller files them ONCE as a descriptor --
// a curious_torch_desc_t of pointers to the C ABI's own structs, registered here under an int64 handle
// (desc_register / desc_release; curious_amd/torch_ops.py desc_create builds it and keeps what it points to alive) --
// and the ops take the handle + tensors:
//   torch.ops.curious_hip.her_sample(desc, storage, batch)                                      her.py:99-183, ddpg.py:326-353
//   torch.ops.curious_hip.ddpg_update(desc, theta, theta_target, batch, workspace, grad, losses, Q_pi, m, v, step_ctr,
//                                     alpha_tab, next_batch, storage, params_unchanged)         ddpg.py:235-248, mpi_adam.py:29-35
//   torch.ops.curious_hip.policy_rollout(desc, theta, workspace, u_out, counter_base, episode, tasks, o, ag, g, td, staging,
//                                        flags)                                                 rollout.py:226-303 x T
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "../../include/curious_hip.h"

namespace {

curious_stream_t current_stream() { return reinterpret_cast<curious_stream_t>(c10::hip::getCurrentHIPStream().stream()); }

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", curious_last_error()); }

// a float32 / given-dtype tensor on the GPU whose elements are contiguous
void dev(const at::Tensor& t, const char* name, at::ScalarType dtype = at::kFloat) {
  TORCH_CHECK(t.defined() && t.is_cuda(), "curious_hip: `", name, "` must be a GPU tensor");
  TORCH_CHECK(t.scalar_type() == dtype, "curious_hip: `", name, "` must be ", dtype, ", got ", t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), "curious_hip: `", name, "` must be contiguous");
}
// a row matrix: unit stride inside a row, any stride between rows
void rows(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.defined() && t.is_cuda() && t.scalar_type() == at::kFloat && t.dim() == 2 && (t.size(1) <= 1 || t.stride(1) == 1),
              "curious_hip: `", name, "` must be a float32 GPU matrix with contiguous rows");
}
float* f32(const at::Tensor& t) { return t.data_ptr<float>(); }

// cfg_i = [dimo, dimg, dimu, dimtd, hidden, layers, modular, clip_pos_returns, normalize_obs],
// cfg_f = [max_u, gamma, clip_return, action_l2, norm_clip]                                   (curious_net_cfg_t)
curious_net_cfg_t net_cfg(at::IntArrayRef ci, at::ArrayRef<double> cf) {
  TORCH_CHECK(ci.size() == 9 && cf.size() == 5, "curious_hip: cfg_i has 9 entries and cfg_f 5 (curious_net_cfg_t)");
  curious_net_cfg_t c;
  std::memset(&c, 0, sizeof(c));
  c.dimo = (int32_t)ci[0]; c.dimg = (int32_t)ci[1]; c.dimu = (int32_t)ci[2]; c.dimtd = (int32_t)ci[3];
  c.hidden = (int32_t)ci[4]; c.layers = (int32_t)ci[5]; c.modular = ci[6] != 0;
  c.clip_pos_returns = ci[7] != 0; c.normalize_obs = ci[8] != 0;
  c.max_u = (float)cf[0]; c.gamma = (float)cf[1]; c.clip_return = (float)std::fmin(cf[2], 3.0e38);
  c.action_l2 = (float)cf[3]; c.norm_clip = (float)std::fmin(cf[4], 3.0e38);
  return c;
}

void polyak_update(at::Tensor target, const at::Tensor& main, double polyak) {
  dev(target, "target"); dev(main, "main");
  TORCH_CHECK(target.numel() == main.numel(), "curious_hip::polyak_update: sizes differ");
  const float p = (float)polyak;                            // (the reference's float32 constants: ddpg.py:459-462)
  check(curious_polyak_update(f32(target), f32(main), target.numel(), p, (float)(1.0 - polyak), current_stream()),
        "curious_polyak_update");
}

void adam_update(at::Tensor theta, at::Tensor m, at::Tensor v, const at::Tensor& grad, int64_t n_Q, int64_t n_pi,
                 double alpha_Q, double alpha_pi) {
  dev(theta, "theta"); dev(m, "m"); dev(v, "v"); dev(grad, "grad");
  TORCH_CHECK(n_Q >= 0 && n_pi >= 0 && theta.numel() >= n_Q + n_pi && m.numel() >= n_Q + n_pi && v.numel() >= n_Q + n_pi &&
                  grad.numel() >= n_Q + n_pi, "curious_hip::adam_update: vectors shorter than n_Q + n_pi");
  const float alpha[2] = {(float)alpha_Q, (float)alpha_pi};
  check(curious_adam_update(f32(theta), f32(m), f32(v), f32(grad), n_Q, n_pi, nullptr, nullptr, 0, 0, alpha, 0.9f,
                            (float)(1 - 0.9), 0.999f, (float)(1 - 0.999), 1e-08f, nullptr, current_stream()),
        "curious_adam_update");
}

at::Tensor param_checksum(const at::Tensor& theta) {
  dev(theta, "theta");
  at::Tensor out = at::zeros({2}, theta.options().dtype(at::kLong));
  check(curious_param_checksum(f32(theta), theta.numel(), reinterpret_cast<uint64_t*>(out.data_ptr<int64_t>()),
                               current_stream()), "curious_param_checksum");
  return out;
}

void norm_update(const at::Tensor& rws, int64_t col_off, int64_t dim, at::Tensor acc) {
  rows(rws, "rows"); dev(acc, "acc");
  const int64_t n = rws.size(0);
  TORCH_CHECK(col_off >= 0 && dim > 0 && col_off + dim <= rws.size(1) && acc.numel() >= 2 * dim + 1,
              "curious_hip::norm_update: columns outside the rows / accumulator too short");
  at::Tensor scratch = at::empty({curious_norm_scratch_doubles((int32_t)n, (int32_t)dim)}, rws.options().dtype(at::kDouble));
  check(curious_norm_update(f32(rws), (int32_t)n, (int32_t)rws.stride(0), (int32_t)col_off, (int32_t)dim, f32(acc),
                            scratch.data_ptr<double>(), current_stream()), "curious_norm_update");
}

void norm_recompute(at::Tensor acc, at::Tensor state, double world_size, double eps) {
  dev(acc, "acc"); dev(state, "state");
  const int64_t dim = (state.numel() - 1) / 4;
  TORCH_CHECK(dim > 0 && state.numel() == 4 * dim + 1 && acc.numel() >= 2 * dim + 1,
              "curious_hip::norm_recompute: state is [sum | sumsq | count | mean | std]");
  check(curious_norm_recompute(f32(acc), f32(state), (int32_t)dim, (float)world_size, (float)eps, current_stream()),
        "curious_norm_recompute");
}

std::tuple<at::Tensor, at::Tensor> policy_forward(at::IntArrayRef cfg_i, at::ArrayRef<double> cfg_f, const at::Tensor& theta,
                                                  const at::Tensor& o, const at::Tensor& g, const at::Tensor& td,
                                                  double clip_obs, bool compute_Q) {
  const curious_net_cfg_t cfg = net_cfg(cfg_i, cfg_f);
  dev(theta, "theta"); rows(o, "o"); rows(g, "g");
  const int64_t n = o.size(0);
  TORCH_CHECK(o.size(1) >= cfg.dimo && g.size(0) == n && g.size(1) >= cfg.dimg, "curious_hip::policy_forward: o / g shapes");
  if (cfg.dimtd > 0) {
    rows(td, "td");
    TORCH_CHECK(td.size(0) == n && td.size(1) >= cfg.dimtd, "curious_hip::policy_forward: td shape");
  }
  TORCH_CHECK(theta.numel() >= curious_param_total(&cfg), "curious_hip::policy_forward: theta shorter than the networks");
  at::Tensor ws = at::empty({curious_workspace_floats(&cfg, (int32_t)n)}, o.options());
  at::Tensor pi = at::empty({n, cfg.dimu}, o.options());
  at::Tensor Q = at::empty({n, 1}, o.options());
  check(curious_policy_forward(&cfg, f32(theta), f32(o), (int32_t)o.stride(0), nullptr, 0, f32(g), (int32_t)g.stride(0),
                               cfg.dimtd > 0 ? f32(td) : nullptr, cfg.dimtd > 0 ? (int32_t)td.stride(0) : 0, (int32_t)n,
                               (float)clip_obs, 0, nullptr, nullptr, f32(ws), f32(pi), compute_Q ? f32(Q) : nullptr,
                               current_stream()), "curious_policy_forward");
  return std::make_tuple(pi, Q);
}

// batch_layout = [off_o, off_td, off_u, off_g, off_o2, off_g2, off_r, off_ag, off_ag2, off_extra, stride]
std::tuple<at::Tensor, at::Tensor> ddpg_grads(at::IntArrayRef cfg_i, at::ArrayRef<double> cfg_f, const at::Tensor& theta,
                                              const at::Tensor& theta_target, const at::Tensor& batch,
                                              at::IntArrayRef batch_layout, at::Tensor grad) {
  const curious_net_cfg_t cfg = net_cfg(cfg_i, cfg_f);
  dev(theta, "theta"); dev(theta_target, "theta_target"); dev(batch, "batch"); dev(grad, "grad");
  TORCH_CHECK(batch.dim() == 2 && batch_layout.size() == 11, "curious_hip::ddpg_grads: batch [B, stride], batch_layout of 11");
  curious_batch_layout_t BL;
  int32_t* f = &BL.off_o;
  for (int i = 0; i < 11; ++i) f[i] = (int32_t)batch_layout[i];
  const int64_t B = batch.size(0), total = curious_param_total(&cfg);
  TORCH_CHECK(BL.stride == batch.size(1), "curious_hip::ddpg_grads: batch_layout.stride != batch.shape[1]");
  TORCH_CHECK(theta.numel() >= total && theta_target.numel() >= total && grad.numel() >= total,
              "curious_hip::ddpg_grads: parameter vectors shorter than the networks");
  at::Tensor ws = at::zeros({curious_workspace_floats(&cfg, (int32_t)B)}, batch.options());   // (holds the fault word)
  at::Tensor losses = at::zeros({2}, batch.options());
  at::Tensor Q_pi = at::zeros({B, 1}, batch.options());
  check(curious_ddpg_grads(&cfg, f32(theta), f32(theta_target), f32(batch), &BL, (int32_t)B, nullptr, nullptr, f32(ws),
                           f32(grad), f32(losses), f32(Q_pi), nullptr, 0, nullptr, current_stream()), "curious_ddpg_grads");
  return std::make_tuple(losses, Q_pi);
}

// ------------------------------------------------------------------ descriptors of the three hot entry points
// (mirrored field by field by curious_amd/torch_ops.py TorchDesc; pointers a given op does not need stay NULL)
struct curious_torch_desc_t {
  const curious_net_cfg_t* cfg; const curious_layout_t* L; const curious_batch_layout_t* BL;
  const curious_tasks_t* tasks; const curious_sample_params_t* P; const curious_sample_rng_t* rng;
  const curious_sample_plan_t* plan; const curious_env_cfg_t* E;
  const float* o_stats; const float* g_stats;
  int64_t buf_stride, tab_base;
  uint64_t seed, counter;
  double noise_scale, random_eps, reward_eps;
  float clip_obs;
  int32_t n, B, env_id0, t0, nsteps, off_change, off_success, relative_goals;
};
std::mutex g_desc_mutex;
std::unordered_map<int64_t, const curious_torch_desc_t*> g_desc;
int64_t g_desc_next = 1;

int64_t desc_register(int64_t address) {
  TORCH_CHECK(address != 0, "curious_hip::desc_register: NULL descriptor");
  std::lock_guard<std::mutex> lock(g_desc_mutex);
  g_desc[g_desc_next] = reinterpret_cast<const curious_torch_desc_t*>(address);
  return g_desc_next++;
}
void desc_release(int64_t handle) {
  std::lock_guard<std::mutex> lock(g_desc_mutex);
  g_desc.erase(handle);
}
const curious_torch_desc_t& desc(int64_t handle) {
  std::lock_guard<std::mutex> lock(g_desc_mutex);
  auto it = g_desc.find(handle);
  TORCH_CHECK(it != g_desc.end(), "curious_hip: unknown descriptor handle ", handle, " (torch_ops.desc_create)");
  return *it->second;
}

void her_sample(int64_t handle, const at::Tensor& storage, at::Tensor batch) {
  const curious_torch_desc_t& d = desc(handle);
  TORCH_CHECK(d.L && d.BL && d.tasks && d.P && (d.rng || d.plan), "curious_hip::her_sample: the descriptor needs layout, tasks, "
              "params and rng or plan");
  dev(storage, "storage"); rows(batch, "batch");
  TORCH_CHECK(batch.stride(0) == d.BL->stride && batch.size(0) >= d.n, "curious_hip::her_sample: batch is [>= n, batch stride]");
  check(curious_her_sample(f32(storage), d.buf_stride, d.L, d.tasks, d.P, d.plan, d.plan ? nullptr : d.rng, d.n, f32(batch), d.BL,
                           current_stream()), "curious_her_sample");
}

void ddpg_update(int64_t handle, at::Tensor theta, const at::Tensor& theta_target, const at::Tensor& batch, at::Tensor workspace,
                 at::Tensor grad, at::Tensor losses, at::Tensor Q_pi, at::Tensor m, at::Tensor v, at::Tensor step_ctr,
                 const at::Tensor& alpha_tab, at::Tensor next_batch, const at::Tensor& storage, bool params_unchanged) {
  const curious_torch_desc_t& d = desc(handle);
  TORCH_CHECK(d.cfg && d.L && d.BL && d.tasks && d.P && d.rng, "curious_hip::ddpg_update: the descriptor needs cfg, layout, tasks, "
              "params and rng");
  dev(theta, "theta"); dev(theta_target, "theta_target"); rows(batch, "batch"); dev(workspace, "workspace"); dev(grad, "grad");
  dev(losses, "losses"); dev(Q_pi, "Q_pi"); dev(m, "m"); dev(v, "v"); dev(step_ctr, "step_ctr", at::kLong);
  dev(alpha_tab, "alpha_tab"); rows(next_batch, "next_batch"); dev(storage, "storage");
  const int64_t total = curious_param_total(d.cfg);
  TORCH_CHECK(theta.numel() >= total && theta_target.numel() >= total && grad.numel() >= total && m.numel() >= total &&
                  v.numel() >= total, "curious_hip::ddpg_update: parameter vectors shorter than the networks");
  TORCH_CHECK(batch.size(0) >= d.B && batch.stride(0) == d.BL->stride && next_batch.size(0) >= d.B &&
                  next_batch.stride(0) == d.BL->stride && next_batch.data_ptr() != batch.data_ptr(),
              "curious_hip::ddpg_update: batch / next_batch are two staging tensors [B, batch stride]");
  TORCH_CHECK(workspace.numel() >= curious_workspace_floats(d.cfg, d.B) && Q_pi.numel() >= d.B && alpha_tab.dim() == 2 &&
                  alpha_tab.size(1) == 2, "curious_hip::ddpg_update: workspace / Q_pi / alpha_tab [len, 2] too small");
  curious_adam_state_t A;
  std::memset(&A, 0, sizeof(A));
  A.m = f32(m); A.v = f32(v); A.alpha_tab = f32(alpha_tab); A.tab_base = d.tab_base; A.tab_len = (int32_t)alpha_tab.size(0);
  A.beta1 = 0.9f; A.one_minus_beta1 = (float)(1 - 0.9); A.beta2 = 0.999f; A.one_minus_beta2 = (float)(1 - 0.999);
  A.epsilon = 1e-08f; A.params_unchanged = params_unchanged ? 1 : 0;
  curious_next_batch_t N;
  N.storage = f32(storage); N.buf_stride = d.buf_stride; N.L = d.L; N.tasks = d.tasks; N.P = d.P; N.rng = d.rng;
  N.batch = f32(next_batch);
  check(curious_ddpg_update(d.cfg, f32(theta), f32(theta_target), f32(batch), d.BL, d.B, d.o_stats, d.g_stats, f32(workspace),
                            f32(grad), f32(losses), f32(Q_pi), step_ctr.data_ptr<int64_t>(), &A, &N, current_stream()),
        "curious_ddpg_update");
}

void policy_rollout(int64_t handle, const at::Tensor& theta, at::Tensor workspace, at::Tensor u_out, const at::Tensor& counter_base,
                    at::Tensor episode, const at::Tensor& tasks, at::Tensor o, at::Tensor ag, const at::Tensor& g,
                    const at::Tensor& td, at::Tensor staging, at::Tensor flags) {
  const curious_torch_desc_t& d = desc(handle);
  TORCH_CHECK(d.cfg && d.E && d.L && d.n > 0 && d.nsteps > 0, "curious_hip::policy_rollout: the descriptor needs cfg, ecfg, layout, n, "
              "nsteps");
  dev(theta, "theta"); dev(workspace, "workspace"); rows(u_out, "u_out"); dev(counter_base, "counter_base", at::kLong);
  dev(episode, "episode", at::kInt); dev(tasks, "tasks", at::kInt); dev(o, "o"); dev(ag, "ag"); dev(g, "g"); dev(td, "td");
  dev(staging, "staging"); dev(flags, "flags");
  TORCH_CHECK(u_out.size(0) >= d.n && episode.numel() >= d.n && tasks.numel() >= d.n && flags.numel() > d.n,
              "curious_hip::policy_rollout: per-env tensors shorter than n");
  if (d.o_stats || d.g_stats || d.relative_goals)
    check(curious_policy_rollout_stats(d.cfg, f32(theta), d.n, d.clip_obs, f32(workspace), d.noise_scale, d.random_eps, d.seed,
                                       d.counter, counter_base.data_ptr<int64_t>(), f32(u_out), (int32_t)u_out.stride(0), d.E,
                                       d.L, d.env_id0, episode.data_ptr<int32_t>(), tasks.data_ptr<int32_t>(), d.t0, d.nsteps,
                                       f32(o), f32(ag), f32(g), f32(td), f32(staging), d.off_change, d.off_success,
                                       d.reward_eps, f32(flags), d.relative_goals, d.o_stats, d.g_stats, current_stream()),
          "curious_policy_rollout_stats");
  else
    check(curious_policy_rollout(d.cfg, f32(theta), d.n, d.clip_obs, f32(workspace), d.noise_scale, d.random_eps, d.seed,
                                 d.counter, counter_base.data_ptr<int64_t>(), f32(u_out), (int32_t)u_out.stride(0), d.E, d.L,
                                 d.env_id0, episode.data_ptr<int32_t>(), tasks.data_ptr<int32_t>(), d.t0, d.nsteps, f32(o),
                                 f32(ag), f32(g), f32(td), f32(staging), d.off_change, d.off_success, d.reward_eps, f32(flags),
                                 current_stream()), "curious_policy_rollout");
}

}  // namespace

TORCH_LIBRARY_FRAGMENT(curious_hip, m) {
  m.def("desc_register(int address) -> int", &desc_register);
  m.def("desc_release(int handle) -> ()", &desc_release);
  m.def("her_sample(int desc, Tensor storage, Tensor(a!) batch) -> ()");
  m.def("ddpg_update(int desc, Tensor(a!) theta, Tensor theta_target, Tensor batch, Tensor(b!) workspace, Tensor(c!) grad, "
        "Tensor(d!) losses, Tensor(e!) Q_pi, Tensor(f!) m, Tensor(g!) v, Tensor(h!) step_ctr, Tensor alpha_tab, "
        "Tensor(i!) next_batch, Tensor storage, bool params_unchanged) -> ()");
  m.def("policy_rollout(int desc, Tensor theta, Tensor(a!) workspace, Tensor(b!) u_out, Tensor counter_base, Tensor(c!) episode, "
        "Tensor tasks, Tensor(d!) o, Tensor(e!) ag, Tensor g, Tensor td, Tensor(f!) staging, Tensor(g!) flags) -> ()");
  m.def("polyak_update(Tensor(a!) target, Tensor main, float polyak) -> ()");
  m.def("adam_update(Tensor(a!) theta, Tensor(b!) m, Tensor(c!) v, Tensor grad, int n_Q, int n_pi, float alpha_Q, "
        "float alpha_pi) -> ()");
  m.def("param_checksum(Tensor theta) -> Tensor");
  m.def("norm_update(Tensor rows, int col_off, int dim, Tensor(a!) acc) -> ()");
  m.def("norm_recompute(Tensor(a!) acc, Tensor(b!) state, float world_size, float eps) -> ()");
  m.def("policy_forward(int[] cfg_i, float[] cfg_f, Tensor theta, Tensor o, Tensor g, Tensor td, float clip_obs, "
        "bool compute_Q) -> (Tensor, Tensor)");
  m.def("ddpg_grads(int[] cfg_i, float[] cfg_f, Tensor theta, Tensor theta_target, Tensor batch, int[] batch_layout, "
        "Tensor(a!) grad) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(curious_hip, CUDA, m) {
  m.impl("polyak_update", &polyak_update);
  m.impl("adam_update", &adam_update);
  m.impl("param_checksum", &param_checksum);
  m.impl("norm_update", &norm_update);
  m.impl("norm_recompute", &norm_recompute);
  m.impl("policy_forward", &policy_forward);
  m.impl("ddpg_grads", &ddpg_grads);
  m.impl("her_sample", &her_sample);
  m.impl("ddpg_update", &ddpg_update);
  m.impl("policy_rollout", &policy_rollout);
}
