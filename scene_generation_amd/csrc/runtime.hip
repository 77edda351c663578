// Error string + opt-in HIP-event profiler (include/sg2im_hip.h: sg_last_error_string, sg_prof_*).
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

static thread_local char t_err[512] = "ok";

void sg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_err, sizeof(t_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sg_last_error_string(void) { return t_err; }
extern "C" int sg_version(void) { return 100; }

int g_sg_prof_on = 0;

// ---- option table ------------------------------------------------------------------------------------------------
std::atomic<int> g_sg_opt[SG_OPT_COUNT];
FILE* g_sg_launch_log = nullptr;
namespace {
struct OptDef { const char* name; int def; };
const OptDef kOpts[SG_OPT_COUNT] = {
    {"tile", -1}, {"t128_min", 384}, {"tile3", 1}, {"tile3_min", 768}, {"split_target", 1536}, {"split_kmin", 2048}, {"splits", -1}, {"fixedtap", 1}, {"wino_wt", 1},
    {"w24_small", 1}, {"w24_s", -1}, {"w24_pmin", 256}, {"wino_adjoint", 1}, {"wino24", 1}, {"linear_nsub", 2},
    {"linear_skinny", 2048}, {"wgrad_rowsum", 1}, {"layout_reg", 1}, {"layout_dsplit", 1}, {"bn_blocks", 4096},
    {"instnorm_reg", 2}, {"wgrad_xcd", 1}, {"wino_reuse", 1}, {"wino_fold_cells", 1}, {"wino_pipe", 2},
    {"check_indices", 0}, {"last_block", 0}, {"wino_gemm_tile", 0}, {"wino43", 1}, {"gconv_fused_gather", 1}, {"w24_gemm_tile", 2}, {"wino_in_fuse", 1}, {"w43_nsub", 1}, {"w43_kfold", 256}, {"wave_prio", 0}, {"par_xcd_chunk", 16}, {"w43_tail_split", 2}, {"tail_smax", 4}, {"tail_ktmin", 8}, {"w43_wgrad_tile", 0}, {"par_split", 1}, {"tail_capture", 1}};
// runs when the shared library is loaded, before any entry point can be called: the ONLY place the environment is read
struct OptInit {
  OptInit() {
    for (int i = 0; i < SG_OPT_COUNT; ++i) {
      char env[64] = "SG_";
      size_t n = 3;
      for (const char* c = kOpts[i].name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
      env[n] = 0;
      const char* e = getenv(env);
      g_sg_opt[i].store((e && *e) ? atoi(e) : kOpts[i].def, std::memory_order_relaxed);
    }
    const char* lf = getenv("SG_LAUNCH_LOG");
    if (lf && *lf) g_sg_launch_log = fopen(lf, "a");
  }
} g_opt_init;
int opt_index(const char* key) {
  if (!key) return -1;
  for (int i = 0; i < SG_OPT_COUNT; ++i)
    if (strcmp(key, kOpts[i].name) == 0) return i;
  return -1;
}
}  // namespace

extern "C" int sg_num_options(void) { return SG_OPT_COUNT; }
extern "C" const char* sg_option_name(int i) { return (i >= 0 && i < SG_OPT_COUNT) ? kOpts[i].name : nullptr; }
extern "C" int sg_option_default(int i) { return (i >= 0 && i < SG_OPT_COUNT) ? kOpts[i].def : 0; }
extern "C" int sg_get_option(const char* key, int* value) {
  const int i = opt_index(key);
  if (i < 0 || !value) { sg_set_error("sg_get_option: unknown option '%s'", key ? key : "(null)"); return -1; }
  *value = sg_opt(i);
  return 0;
}
extern "C" int sg_set_option(const char* key, int value) {
  const int i = opt_index(key);
  if (i < 0) { sg_set_error("sg_set_option: unknown option '%s'", key ? key : "(null)"); return -1; }
  g_sg_opt[i].store(value, std::memory_order_relaxed);
  return 0;
}

// ---- ticket counters of the in-launch finalisation (common.h: sg_arrive_last) ---------------------------------------------------
// One zero-initialised pool per device.  Launches outside a stream capture take counters from a RING (a counter is busy only
// while its launch runs and is left at zero by the last arriver: 65536 slots cannot wrap onto a launch still in flight);
// launches recorded into a hipGraph keep their counter for the life of the graph, so they come from a second region that is
// only ever handed out once (exhausted -> nullptr -> the caller's two-kernel form).
namespace {
constexpr int CNT_RING = 65536, CNT_CAPTURED = 262144, CNT_DEVICES = 16;
std::mutex g_cnt_mu;
int* g_cnt_pool[CNT_DEVICES] = {};
unsigned g_cnt_ring[CNT_DEVICES] = {};
int g_cnt_cap[CNT_DEVICES] = {};
}  // namespace

int* sg_counter_alloc(hipStream_t s, int n, bool always, int family) {
  if ((!always && !(sg_opt(SG_OPT_LAST_BLOCK) & family)) || n < 1 || n > 4096) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CNT_DEVICES) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lk(g_cnt_mu);
  if (!g_cnt_pool[dev]) {
    if (capturing) return nullptr;                        // (no allocation / memset inside a capture)
    int* p = nullptr;
    const size_t bytes = (size_t)(CNT_RING + CNT_CAPTURED) * sizeof(int);
    if (hipMalloc((void**)&p, bytes) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, bytes) != hipSuccess) { hipFree(p); return nullptr; }      // synchronous: visible to every later launch
    g_cnt_pool[dev] = p;
  }
  if (capturing) {
    if (g_cnt_cap[dev] + n > CNT_CAPTURED) return nullptr;
    int* r = g_cnt_pool[dev] + CNT_RING + g_cnt_cap[dev];
    g_cnt_cap[dev] += n;
    return r;
  }
  unsigned at = g_cnt_ring[dev];
  if (at + (unsigned)n > (unsigned)CNT_RING) at = 0;
  g_cnt_ring[dev] = at + (unsigned)n;
  return g_cnt_pool[dev] + at;
}

namespace {
struct TailScratch { int dev; hipStream_t s; float* p; size_t bytes; };
std::vector<TailScratch> g_tail;
}  // namespace

// Launches that are being CAPTURED into a hipGraph cannot allocate, and the stream they are captured on is not the stream the
// graph will be replayed on: they share one buffer per device (g_tail_cap), grown whenever an EAGER launch on any stream asks for
// more than it holds -- the warm-up iterations that precede every capture (graphs.py) run the same shapes eagerly, so the buffer is
// large enough by the time the capture asks.  Replays of captured graphs are ordered with each other on the replaying stream
// (graphs.py replays every segment on the current stream); eager launches use their own stream's buffer.  Until round 6 a capture
// got nullptr here, i.e. every tail-split schedule was silently OFF inside the graphed generator segments -- the headline pass
// never ran it (the event-profiled pass, which runs eagerly, did: 0.15-0.3 ms per step).
namespace {
struct TailCap { float* p; size_t bytes; };
TailCap g_tail_cap[CNT_DEVICES] = {};
void tail_cap_reserve(int dev, size_t bytes) {          // g_cnt_mu held; outside captures only
  if (dev < 0 || dev >= CNT_DEVICES || g_tail_cap[dev].bytes >= bytes) return;
  float* q = nullptr;
  if (hipMalloc((void**)&q, bytes) != hipSuccess) return;
  g_tail_cap[dev].p = q; g_tail_cap[dev].bytes = bytes;   // (the old buffer may be baked into a captured graph: kept)
}
}  // namespace

float* sg_tail_scratch(hipStream_t s, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lk(g_cnt_mu);
  if (capturing)
    return (sg_opt(SG_OPT_TAIL_CAPTURE) && dev >= 0 && dev < CNT_DEVICES && g_tail_cap[dev].bytes >= bytes) ? g_tail_cap[dev].p : nullptr;
  tail_cap_reserve(dev, bytes);
  for (auto& t : g_tail)
    if (t.dev == dev && t.s == s) {
      if (t.bytes >= bytes) return t.p;
      // a larger request: the old buffer may still be read by a launch in flight -- it is kept (a few MB), a new one takes its place
      float* q = nullptr;
      if (hipMalloc((void**)&q, bytes) != hipSuccess) return nullptr;
      t.p = q; t.bytes = bytes;
      return q;
    }
  float* q = nullptr;
  if (hipMalloc((void**)&q, bytes) != hipSuccess) return nullptr;
  g_tail.push_back(TailScratch{dev, s, q, bytes});
  return q;
}

namespace {
struct Rec { int kind; hipEvent_t e0, e1; double flops, bytes; };
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;
hipEvent_t g_open[SG_K_COUNT];
double g_ms[SG_K_COUNT], g_flops[SG_K_COUNT], g_bytes[SG_K_COUNT];
int64_t g_cnt[SG_K_COUNT];
const char* kTail[SG_K_COUNT - SG_K_IGEMM_COUNT] = {"linear", "layout_fwd", "layout_bwd", "instnorm", "batchnorm", "adam",
                                                    "segsum", "crop", "other", "wino_bgemm_t128", "wino_bgemm_t64",
                                                    "wino_transforms", "head_conv", "instnorm_bwd", "wino24_bgemm_t128", "wino43_bgemm_t64"};
char g_names[SG_K_COUNT][32];
bool g_names_init = false;
void init_names() {
  if (g_names_init) return;
  const char* fam[3] = {"kn0", "kn1", "nk"};
  const int ks[4] = {1, 3, 4, 7};
  const char* tl[4] = {"128", "64", "32", "64x128"};
  for (int f = 0; f < 3; ++f)
    for (int k = 0; k < 4; ++k)
      for (int t = 0; t < 4; ++t)
        snprintf(g_names[f * 16 + k * 4 + t], 32, "igemm_%s_k%d_t%s", fam[f], ks[k], tl[t]);
  for (int i = SG_K_IGEMM_COUNT; i < SG_K_COUNT; ++i) snprintf(g_names[i], 32, "%s", kTail[i - SG_K_IGEMM_COUNT]);
  g_names_init = true;
}

hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

void drain() {   // caller holds g_mu
  for (auto& r : g_recs) {
    hipEventSynchronize(r.e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    g_ms[r.kind] += ms; g_flops[r.kind] += r.flops; g_bytes[r.kind] += r.bytes; g_cnt[r.kind] += 1;
    g_free.push_back(r.e0); g_free.push_back(r.e1);
  }
  g_recs.clear();
}
}  // namespace

void sg_prof_begin(int kind, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e = get_event();
  hipEventRecord(e, s);
  g_open[kind] = e;
}

void sg_prof_end(int kind, hipStream_t s, double flops, double bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e = get_event();
  hipEventRecord(e, s);
  g_recs.push_back(Rec{kind, g_open[kind], e, flops, bytes});
  if (g_recs.size() > 8192) drain();
}

extern "C" int sg_prof_enable(int on) { g_sg_prof_on = on; return 0; }
extern "C" int sg_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  drain();
  for (int i = 0; i < SG_K_COUNT; ++i) { g_ms[i] = g_flops[i] = g_bytes[i] = 0; g_cnt[i] = 0; }
  return 0;
}
extern "C" int sg_prof_num_kinds(void) { return SG_K_COUNT; }
extern "C" const char* sg_prof_kind_name(int kind) {
  init_names();
  return (kind >= 0 && kind < SG_K_COUNT) ? g_names[kind] : "?";
}
extern "C" int sg_prof_read(int kind, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  if (kind < 0 || kind >= SG_K_COUNT) { sg_set_error("sg_prof_read: bad kind %d", kind); return -1; }
  std::lock_guard<std::mutex> lk(g_mu);
  drain();
  if (total_ms) *total_ms = g_ms[kind];
  if (launches) *launches = g_cnt[kind];
  if (flops) *flops = g_flops[kind];
  if (bytes) *bytes = g_bytes[kind];
  return 0;
}
