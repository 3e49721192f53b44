// Error reporting and device probing for libcurious_hip.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void curious_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* curious_last_error(void) { return g_err; }

extern "C" int curious_abi_version(void) { return CURIOUS_ABI_VERSION; }

#ifndef CURIOUS_BUILD_DIGEST
#define CURIOUS_BUILD_DIGEST "unknown"
#endif
extern "C" const char* curious_build_digest(void) { return CURIOUS_BUILD_DIGEST; }

extern "C" int curious_device_info(char* name_host, int name_len, int* cu_count_host) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  CURIOUS_CHECK(e == hipSuccess, "curious_device_info: no HIP device: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  CURIOUS_CHECK(e == hipSuccess, "curious_device_info: hipGetDeviceProperties: %s", hipGetErrorString(e));
  CURIOUS_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "curious_device_info: libcurious_hip is built for gfx950 (MI355X), found %s", prop.gcnArchName);
  if (name_host && name_len > 0) {
    strncpy(name_host, prop.name, name_len - 1);
    name_host[name_len - 1] = 0;
  }
  if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
  return 0;
}

// ------------------------------------------------------------------ per-kernel event timing
#include <vector>
int g_curious_prof_on = 0;
namespace {
struct Pair { int kid; hipEvent_t a, b; };
std::vector<Pair> g_pairs;
std::vector<hipEvent_t> g_free;
hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace

void curious_prof_push(int kid, hipStream_t st, bool start) {
  if (start) {
    Pair p; p.kid = kid; p.a = get_event(); p.b = nullptr;
    (void)hipEventRecord(p.a, st);
    g_pairs.push_back(p);
  } else {
    Pair& p = g_pairs.back();
    p.b = get_event();
    (void)hipEventRecord(p.b, st);
  }
}

static const char* k_names[CK_COUNT] = {
    "her_sample_kernel", "store_episodes_kernel", "episode_activity_kernel", "norm_partial_kernel",
    "norm_final_kernel", "norm_recompute_kernel", "norm_pair_partial_kernel", "norm_pair_final_kernel",
    "fwd_l0_kernel", "fwd_layer_kernel", "fwd_hot_kernel", "dx_hot_kernel", "dx_kernel", "dw_all_kernel", "dw_kernel",
    "head_fwd_kernel", "dx_crit_kernel", "critic_head_kernel", "dx_actor_kernel", "actor_dz_kernel", "adam_kernel",
    "adam_her_kernel", "polyak_kernel", "checksum_kernel", "action_noise_kernel", "env_reset_kernel", "env_step_kernel",
    "counter_add_kernel", "fwd_pi_kernel", "dw_adam_her_kernel", "act_step_kernel", "fwd_l01_kernel",
    "ddpg_rows_kernel", "policy_rows_kernel", "rows_transpose_kernel", "route_episodes_kernel",
    "policy_resident_kernel", "ddpg_rows_her_kernel", "allreduce_adam_ipc_kernel"};
int64_t g_curious_launches[CK_COUNT] = {0};

// launches per kernel id since the library was loaded (counted whether or not event timing is enabled, also during
// hipGraph capture -- a captured launch counts once, its replays do not)
extern "C" int curious_prof_launch_counts(int64_t* counts_host) {
  CURIOUS_CHECK(counts_host, "curious_prof_launch_counts: NULL argument");
  for (int k = 0; k < CK_COUNT; ++k) counts_host[k] = g_curious_launches[k];
  return 0;
}

// ------------------------------------------------------------------ run-time options
#include <stdlib.h>
CuriousOptions& curious_options() {
  static CuriousOptions o;
  static bool init = false;
  if (!init) {
    auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    o.rows = env_int("CURIOUS_ROWS", 1) != 0;
    o.rows_xcd = env_int("CURIOUS_ROWS_XCD", 1) != 0;
    o.xcd_map = env_int("CURIOUS_XCD_MAP", 0);
    if (o.xcd_map != 4 && o.xcd_map != 8) o.xcd_map = 0;
    o.fault_inject = 0;
    o.qt_spins = 1 << 22;
    o.lab_no_target = 0;
    o.dw_xcd = env_int("CURIOUS_DW_XCD", 1) != 0;
    o.rows_pre = env_int("CURIOUS_ROWS_PRE", 1) != 0;
    o.rows8 = env_int("CURIOUS_ROWS8", 1) != 0;
    o.rows16 = env_int("CURIOUS_ROWS16", ROWS16_DEFAULT_MIN);
    o.dw64 = env_int("CURIOUS_DW64", 0);
    o.dw_bal = env_int("CURIOUS_DW_BAL", DW_BAL_MIN_B);
    o.fwd16 = 0;
    o.gather_dw = env_int("CURIOUS_GATHER_DW", ROWS16_DEFAULT_MIN);
    o.dw_split = env_int("CURIOUS_DW_SPLIT", 0);
    o.lab_dw_stamps = 0;
    o.lab_rows_stamps = 0;
    o.lab_res_stamps = 0;
    o.resident = env_int("CURIOUS_RESIDENT", 1) != 0;
    o.res_spins = 1 << 20;
    init = true;
  }
  return o;
}
static int* option_slot(const char* name) {
  CuriousOptions& o = curious_options();
  if (!name) return nullptr;
  if (!strcmp(name, "rows")) return &o.rows;
  if (!strcmp(name, "rows_xcd")) return &o.rows_xcd;
  if (!strcmp(name, "xcd_map")) return &o.xcd_map;
  if (!strcmp(name, "fault_inject")) return &o.fault_inject;
  if (!strcmp(name, "qt_spins")) return &o.qt_spins;
  if (!strcmp(name, "lab_no_target")) return &o.lab_no_target;
  if (!strcmp(name, "dw_xcd")) return &o.dw_xcd;
  if (!strcmp(name, "rows_pre")) return &o.rows_pre;
  if (!strcmp(name, "rows8")) return &o.rows8;
  if (!strcmp(name, "rows16")) return &o.rows16;
  if (!strcmp(name, "dw64")) return &o.dw64;
  if (!strcmp(name, "dw_bal")) return &o.dw_bal;
  if (!strcmp(name, "fwd16")) return &o.fwd16;
  if (!strcmp(name, "gather_dw")) return &o.gather_dw;
  if (!strcmp(name, "dw_split")) return &o.dw_split;
  if (!strcmp(name, "lab_dw_stamps")) return &o.lab_dw_stamps;
  if (!strcmp(name, "lab_rows_stamps")) return &o.lab_rows_stamps;
  if (!strcmp(name, "lab_res_stamps")) return &o.lab_res_stamps;
  if (!strcmp(name, "resident")) return &o.resident;
  if (!strcmp(name, "res_spins")) return &o.res_spins;
  return nullptr;
}
extern "C" int curious_set_option(const char* name, int64_t value) {
  int* s = option_slot(name);
  CURIOUS_CHECK(s, "curious_set_option: unknown option '%s'", name ? name : "(null)");
  if (s == &curious_options().xcd_map) CURIOUS_CHECK(value == 0 || value == 4 || value == 8, "xcd_map must be 0, 4 or 8");
  if (s == &curious_options().qt_spins) CURIOUS_CHECK(value >= 1 && value <= (1 << 30), "qt_spins out of range");
  *s = (int)value;
  return 0;
}
extern "C" int64_t curious_get_option(const char* name) {
  int* s = option_slot(name);
  if (!s) { curious_set_error("curious_get_option: unknown option '%s'", name ? name : "(null)"); return -1; }
  return *s;
}

extern "C" int curious_prof_kernel_count(void) { return CK_COUNT; }
extern "C" const char* curious_prof_kernel_name(int kid) { return (kid >= 0 && kid < CK_COUNT) ? k_names[kid] : ""; }

extern "C" int curious_prof_enable(int on) {
  g_curious_prof_on = on ? 1 : 0;
  return 0;
}

// Synchronises the device, then returns per-kernel launch counts and summed durations (ms) since the last collect.
extern "C" int curious_prof_collect(int64_t* counts_host, double* total_ms_host) {
  hipError_t e = hipDeviceSynchronize();
  CURIOUS_CHECK(e == hipSuccess, "curious_prof_collect: %s", hipGetErrorString(e));
  for (int k = 0; k < CK_COUNT; ++k) { counts_host[k] = 0; total_ms_host[k] = 0.0; }
  for (auto& p : g_pairs) {
    float ms = 0.f;
    if (p.b && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      counts_host[p.kid] += 1;
      total_ms_host[p.kid] += ms;
    }
    g_free.push_back(p.a);
    if (p.b) g_free.push_back(p.b);
  }
  g_pairs.clear();
  return 0;
}
