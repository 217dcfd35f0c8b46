// abrk_host.cpp - the C ABI of libabrk.so (include/abrk.h): arm registry, parameter
// conversion, host<->device staging and kernel dispatch.  No arithmetic of the hot path
// happens here and there is no CPU fallback: without a HIP device every compute entry
// point fails with ABRK_ENODEV.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/abrk.h"
#include "abrk_coop.h"
#include "abrk_plugin.h"
#include "abrk_kernels.h"
#include "abrk_params.h"

namespace abrk {
const ArmOps* ops_ur5();
const ArmOps* ops_jaco2();
const ArmOps* ops_twojoint();
const ArmOps* ops_threejoint();
const ArmOps* ops_onejoint();
const ArmOps* ops_rt(int n_joints);
size_t rt_table_size(int n_joints, int dtype);
void rt_table_fill(int n_joints, int dtype, const abrk_arm_desc* d, void* dst);
}  // namespace abrk

using namespace abrk;

// ------------------------------------------------------------------------------- errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (void)hipGetLastError();                                                             \
      return fail(e_ == hipErrorOutOfMemory ? ABRK_ENOMEM : ABRK_ENODEV, "%s: %s", #expr,  \
                  hipGetErrorString(e_));                                                  \
    }                                                                                      \
  } while (0)

extern "C" const char* abrk_last_error(void) { return g_err.c_str(); }
extern "C" int abrk_version(void) { return ABRK_VERSION; }

// ------------------------------------------------------------------------------- arm registry
namespace {
struct ArmEntry {
  bool live = false;
  bool builtin = false;
  int gen = 0;  // user arms: bumped when the slot is freed - an id is slot | gen << kArmSlotBits, so a stale id (a copy of a
                // closed config, a handle cached past abrk_arm_destroy) never matches the slot's next tenant
  abrk_arm_desc desc;
  const ArmOps* ops = nullptr;
  std::vector<unsigned char> rt64, rt32;  // RtArm<N,double> / RtArm<N,float> images (user arms on the runtime-table
                                          // kernels; empty for built-in and compiled arms)
};
// plugins stay loaded for the life of the process (their code objects are registered with the HIP runtime)
struct PluginLib {
  std::string path;
  const ArmOps* ops;
  abrk_arm_desc desc;
};
std::vector<PluginLib> g_plugins;
std::mutex g_mu;
std::vector<ArmEntry> g_arms;
constexpr int kArmSlotBits = 12, kArmSlots = 1 << kArmSlotBits, kArmGenMask = (1 << (31 - kArmSlotBits)) - 1;
int arm_id_of(int slot) { return slot | (g_arms[slot].gen << kArmSlotBits); }
// slot of a live arm id, or -1 (caller holds g_mu)
int arm_slot(int id) {
  if (id < 0) return -1;
  const int slot = id & (kArmSlots - 1);
  if (slot >= (int)g_arms.size() || !g_arms[slot].live || g_arms[slot].gen != (id >> kArmSlotBits)) return -1;
  return slot;
}

void init_builtins() {
  if (!g_arms.empty()) return;
  g_arms.reserve(kArmSlots);  // entries are handed out by address: never reallocate
  g_arms.resize(5);
  desc_from_tab<Tab_ur5>(&g_arms[0].desc);
  g_arms[0].ops = ops_ur5();
  desc_from_tab<Tab_jaco2>(&g_arms[1].desc);
  g_arms[1].ops = ops_jaco2();
  desc_from_tab<Tab_twojoint>(&g_arms[2].desc);
  g_arms[2].ops = ops_twojoint();
  desc_from_tab<Tab_threejoint>(&g_arms[3].desc);
  g_arms[3].ops = ops_threejoint();
  desc_from_tab<Tab_onejoint>(&g_arms[4].desc);
  g_arms[4].ops = ops_onejoint();
  for (auto& a : g_arms) a.live = a.builtin = true;
}

// returns a COPY-safe pointer valid while the registry lock is not needed (entries are
// never moved after creation: vector of stable size is avoided by reserving)
ArmEntry* get_arm(int id) {
  std::lock_guard<std::mutex> lk(g_mu);
  init_builtins();
  const int slot = arm_slot(id);
  return slot < 0 ? nullptr : &g_arms[slot];
}
}  // namespace

extern "C" int abrk_arm_builtin(const char* name) {
  std::lock_guard<std::mutex> lk(g_mu);
  init_builtins();
  if (!name) return fail(ABRK_EINVAL, "abrk_arm_builtin: NULL name");
  for (int i = 0; i < 5; i++)
    if (!strcmp(name, g_arms[i].desc.name)) return i;
  return fail(ABRK_ENOARM, "unknown built-in arm '%s'", name);
}

extern "C" int abrk_arm_create(const abrk_arm_desc* d) {
  if (!d) return fail(ABRK_EINVAL, "abrk_arm_create: NULL desc");
  if (d->n_joints < 1 || d->n_joints > ABRK_MAX_JOINTS)
    return fail(ABRK_EINVAL, "n_joints=%d outside 1..%d", d->n_joints, ABRK_MAX_JOINTS);
  if (d->n_links_dyn < 0 || d->n_links_dyn > d->n_joints + 1)
    return fail(ABRK_EINVAL, "n_links_dyn=%d outside 0..n_joints+1", d->n_links_dyn);
  std::lock_guard<std::mutex> lk(g_mu);
  init_builtins();
  // a destroyed user arm's slot is taken again (ids are handles, like file descriptors): a long-running process
  // may register and drop any number of arms, 4096 at a time
  int slot = -1;
  for (int i = 5; i < (int)g_arms.size() && slot < 0; i++)
    if (!g_arms[i].live) slot = i;
  if (slot < 0 && g_arms.size() >= (size_t)kArmSlots) return fail(ABRK_ENOMEM, "too many arms (%d live at once)", kArmSlots);
  ArmEntry e;
  e.live = true;
  e.desc = *d;
  e.desc.name[sizeof e.desc.name - 1] = 0;
  e.ops = ops_rt(d->n_joints);
  e.rt64.resize(rt_table_size(d->n_joints, ABRK_F64));
  e.rt32.resize(rt_table_size(d->n_joints, ABRK_F32));
  rt_table_fill(d->n_joints, ABRK_F64, d, e.rt64.data());
  rt_table_fill(d->n_joints, ABRK_F32, d, e.rt32.data());
  if (slot >= 0) {
    e.gen = g_arms[slot].gen;
    g_arms[slot] = std::move(e);
    return arm_id_of(slot);
  }
  g_arms.push_back(std::move(e));
  return arm_id_of((int)g_arms.size() - 1);
}

namespace {
bool same_table(const abrk_arm_desc& a, const abrk_arm_desc& b) {
  if (a.n_joints != b.n_joints || a.n_links_dyn != b.n_links_dyn || (a.has_ee != 0) != (b.has_ee != 0)) return false;
  const int n = a.n_joints;
  if (memcmp(a.A0, b.A0, sizeof a.A0)) return false;
  for (int i = 0; i < n; i++)
    if (memcmp(a.AJ[i], b.AJ[i], sizeof a.AJ[i]) || memcmp(a.B[i], b.B[i], sizeof a.B[i])) return false;
  if (a.has_ee && memcmp(a.E, b.E, sizeof a.E)) return false;
  for (int l = 0; l <= n; l++)
    if (memcmp(a.mdiag[l], b.mdiag[l], sizeof a.mdiag[l])) return false;
  return true;
}
}  // namespace

extern "C" const char* abrk_plugin_abi(void) { return ABRK_PLUGIN_ABI; }

extern "C" int abrk_arm_create_compiled(const abrk_arm_desc* d, const char* plugin_path) {
  if (!d || !plugin_path) return fail(ABRK_EINVAL, "abrk_arm_create_compiled: NULL argument");
  if (d->n_joints < 1 || d->n_joints > ABRK_MAX_JOINTS)
    return fail(ABRK_EINVAL, "n_joints=%d outside 1..%d", d->n_joints, ABRK_MAX_JOINTS);
  if (d->n_links_dyn < 0 || d->n_links_dyn > d->n_joints + 1)
    return fail(ABRK_EINVAL, "n_links_dyn=%d outside 0..n_joints+1", d->n_links_dyn);
  std::lock_guard<std::mutex> lk(g_mu);
  init_builtins();
  const PluginLib* pl = nullptr;
  for (const auto& p : g_plugins)
    if (p.path == plugin_path) pl = &p;
  if (!pl) {
    void* h = dlopen(plugin_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(ABRK_EINVAL, "cannot load arm plugin: %s", dlerror());
    auto f_abi = reinterpret_cast<abrk_plugin_abi_fn>(dlsym(h, "abrk_plugin_abi_tag"));
    auto f_ops = reinterpret_cast<abrk_plugin_ops_fn>(dlsym(h, "abrk_plugin_ops"));
    auto f_desc = reinterpret_cast<abrk_plugin_desc_fn>(dlsym(h, "abrk_plugin_desc"));
    if (!f_abi || !f_ops || !f_desc) {
      dlclose(h);
      return fail(ABRK_EINVAL, "%s is not an arm plugin (entry points missing)", plugin_path);
    }
    if (strcmp(f_abi(), ABRK_PLUGIN_ABI) != 0) {
      int rc = fail(ABRK_EINVAL, "arm plugin %s was built for kernel headers %s, this library is %s: rebuild it",
                    plugin_path, f_abi(), ABRK_PLUGIN_ABI);
      dlclose(h);
      return rc;
    }
    PluginLib p;
    p.path = plugin_path;
    p.ops = static_cast<const ArmOps*>(f_ops());
    f_desc(&p.desc);
    g_plugins.push_back(std::move(p));
    pl = &g_plugins.back();
  }
  if (!same_table(pl->desc, *d))
    return fail(ABRK_EINVAL, "arm plugin %s was compiled for a different arm table than the one given", plugin_path);
  int slot = -1;
  for (int i = 5; i < (int)g_arms.size() && slot < 0; i++)
    if (!g_arms[i].live) slot = i;
  if (slot < 0 && g_arms.size() >= (size_t)kArmSlots) return fail(ABRK_ENOMEM, "too many arms (%d live at once)", kArmSlots);
  ArmEntry e;
  e.live = true;
  e.desc = *d;
  e.desc.name[sizeof e.desc.name - 1] = 0;
  e.ops = pl->ops;
  if (slot >= 0) {
    e.gen = g_arms[slot].gen;
    g_arms[slot] = std::move(e);
    return arm_id_of(slot);
  }
  g_arms.push_back(std::move(e));
  return arm_id_of((int)g_arms.size() - 1);
}

extern "C" int abrk_arm_get_desc(int arm_id, abrk_arm_desc* out) {
  ArmEntry* a = get_arm(arm_id);
  if (!a || !out) return fail(ABRK_ENOARM, "unknown arm id %d", arm_id);
  *out = a->desc;
  return 0;
}

extern "C" int abrk_arm_destroy(int arm_id) {
  std::lock_guard<std::mutex> lk(g_mu);
  init_builtins();
  const int slot = arm_slot(arm_id);
  if (slot < 5) return fail(ABRK_ENOARM, "arm id %d is not a (live) user arm", arm_id);
  g_arms[slot].live = false;
  g_arms[slot].gen = (g_arms[slot].gen + 1) & kArmGenMask;
  g_arms[slot].rt64 = {};
  g_arms[slot].rt32 = {};
  return 0;
}

// ------------------------------------------------------------------------------- device plumbing
static thread_local int t_current_device = -1;  // what this thread last hipSetDevice'd (plan launches skip the call)
static int use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(ABRK_ENODEV, "no HIP device available (%s); libabrk has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  }
  if (device < 0 || device >= n) return fail(ABRK_EINVAL, "device %d outside 0..%d", device, n - 1);
  HIPCHK(hipSetDevice(device));
  t_current_device = device;
  return 0;
}

// ---- the "M is not positive definite" flag (ABRK_ESINGULAR; the reference raises LinAlgError at osc.py:136).  The OSC
// kernels store 1 through OscP::status where a row's Cholesky factor of M meets a non-positive pivot.  A word lives in
// pinned, device-mapped host memory, so reading it costs the host nothing once the stream is drained:
//   * a call that hands over HOST arrays is synchronous - it uses the calling thread's own word and returns the code;
//   * a call on DEVICE pointers is asynchronous - it uses the word of its (device, stream), which is reported ONCE by
//     whatever drains that stream next: abrk_stream_sync, abrk_memcpy_d2h on it, abrk_device_sync (every stream of the
//     device).  Several control loops on one GPU each hear of their own singular batch and of nobody else's.
// Words come from a pool (64 of them per pinned 4 KiB block; blocks live as long as the process, words are recycled): a
// thread's word goes back when the thread exits, a stream's word when abrk_stream_destroy is called.
struct StatusWord {
  volatile int* host = nullptr;
  int* dev = nullptr;
  bool take() {  // -> was it raised?  (cleared)
    if (!host || !*host) return false;
    *host = 0;
    return true;
  }
};
namespace {
struct StatusPool {
  std::mutex mu;
  std::vector<StatusWord> free_;
  int64_t blocks = 0, handed_out = 0;
  static constexpr size_t kStride = 64, kBlock = 4096;
  StatusWord get() {  // {nullptr, nullptr}: no pinned memory to be had - nobody listens, as before round 5
    std::lock_guard<std::mutex> lk(mu);
    if (free_.empty()) {
      void *h = nullptr, *d = nullptr;
      if (hipHostMalloc(&h, kBlock, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess ||
          hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (h) (void)hipHostFree(h);
        return {};
      }
      memset(h, 0, kBlock);
      blocks++;
      for (size_t o = 0; o < kBlock; o += kStride)
        free_.push_back(StatusWord{(volatile int*)((char*)h + o), (int*)((char*)d + o)});
    }
    StatusWord w = free_.back();
    free_.pop_back();
    *w.host = 0;
    handed_out++;
    return w;
  }
  void put(StatusWord w) {
    if (!w.host) return;
    std::lock_guard<std::mutex> lk(mu);
    free_.push_back(w);
    handed_out--;
  }
};
// (never destroyed: thread_local destructors of late-exiting threads may still hand words back during process exit)
StatusPool& status_pool() {
  static StatusPool* p = new StatusPool;
  return *p;
}
struct ThreadStatus {
  StatusWord w;
  ~ThreadStatus() { status_pool().put(w); }
};
thread_local ThreadStatus t_status;
std::mutex g_status_mu;
std::map<std::pair<int, void*>, StatusWord> g_stream_status;  // (device, stream) -> the word of its asynchronous calls
}  // namespace
// the word a call's kernels report to (nullptr: the allocation failed)
static StatusWord* status_word(bool synchronous, int device, void* stream = nullptr) {
  if (synchronous) {
    if (!t_status.w.host) t_status.w = status_pool().get();
    return t_status.w.host ? &t_status.w : nullptr;
  }
  std::lock_guard<std::mutex> lk(g_status_mu);
  StatusWord& w = g_stream_status[{device, stream}];  // (node addresses of a std::map are stable)
  if (!w.host) w = status_pool().get();
  return w.host ? &w : nullptr;
}
// was the flag of (device, stream) raised since it was last reported?  all_streams: any stream of the device
static bool status_take(int device, void* stream, bool all_streams) {
  std::lock_guard<std::mutex> lk(g_status_mu);
  bool raised = false;
  if (all_streams) {
    for (auto it = g_stream_status.lower_bound({device, nullptr}); it != g_stream_status.end() && it->first.first == device; ++it)
      raised = it->second.take() || raised;
  } else {
    auto it = g_stream_status.find({device, stream});
    if (it != g_stream_status.end()) raised = it->second.take();
  }
  return raised;
}
static void status_forget_stream(int device, void* stream) {
  std::lock_guard<std::mutex> lk(g_status_mu);
  auto it = g_stream_status.find({device, stream});
  if (it == g_stream_status.end()) return;
  status_pool().put(it->second);
  g_stream_status.erase(it);
}
static int singular_error() {
  return fail(ABRK_ESINGULAR, "Singular matrix: the joint-space inertia matrix M of at least one row is not positive "
                              "definite (numpy.linalg.inv(M) raises LinAlgError there, osc.py:136); the outputs of "
                              "those rows are unspecified");
}

extern "C" int abrk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
extern "C" int abrk_device_name(int device, char* buf, size_t len) {
  if (int rc = use_device(device)) return rc;
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, device));
  // the marketing name comes from libdrm's amdgpu.ids, which minimal images lack: fall back to the architecture
  snprintf(buf, len, "%s (%s, %d CUs)", p.name[0] ? p.name : "AMD GPU", p.gcnArchName, p.multiProcessorCount);
  return 0;
}
extern "C" void* abrk_malloc(int device, size_t bytes) {
  if (use_device(device)) return nullptr;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    fail(ABRK_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}
extern "C" int abrk_free(int device, void* p) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipFree(p));
  return 0;
}
extern "C" int abrk_memcpy_h2d(int device, void* dst, const void* src, size_t bytes, void* stream) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int abrk_memcpy_d2h(int device, void* dst, const void* src, size_t bytes, void* stream) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  // the stream is drained: a singular batch enqueued on it earlier is reported here (the bytes ARE copied; the torques of
  // the offending rows are unspecified - a caller that reads results back this way must not get them silently)
  if (status_take(device, stream, false)) return singular_error();
  return 0;
}
extern "C" int abrk_memset(int device, void* dst, int value, size_t bytes, void* stream) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return 0;
}
extern "C" void* abrk_stream_create(int device) {
  if (use_device(device)) return nullptr;
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    fail(ABRK_ENODEV, "hipStreamCreate failed");
    return nullptr;
  }
  return s;
}
namespace {
void wl_forget_stream(int device, hipStream_t stream);  // the six-row kernels' (device, stream) scratch: below
}
extern "C" int abrk_stream_destroy(int device, void* stream) {
  if (int rc = use_device(device)) return rc;
  if (stream) wl_forget_stream(device, (hipStream_t)stream);
  HIPCHK(hipStreamDestroy((hipStream_t)stream));  // (waits for the stream's work: nothing writes its word any more)
  if (stream) status_forget_stream(device, stream);
  return 0;
}
extern "C" int abrk_stream_sync(int device, void* stream) {
  if (int rc = use_device(device)) return rc;
  // ABRK_SYNC_SPIN=1 (measurement switch): poll the stream instead of blocking in hipStreamSynchronize - whether the
  // wake-up latency of the blocking wait shows in a 20-step replay (it does not: profiles/round3/launch_latency.txt)
  static const bool spin = measurement_env("ABRK_SYNC_SPIN") != nullptr;
  if (spin) {
    hipError_t e;
    while ((e = hipStreamQuery((hipStream_t)stream)) == hipErrorNotReady) {
    }
    (void)hipGetLastError();
    HIPCHK(e);
  } else {
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  }
  // the "M not positive definite" flag of THIS stream's asynchronous (device-pointer) OSC calls: reported once
  if (status_take(device, stream, false)) return singular_error();
  return 0;
}
extern "C" int abrk_device_sync(int device) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipDeviceSynchronize());
  if (status_take(device, nullptr, true)) return singular_error();  // every stream of the device is drained
  return 0;
}
extern "C" void* abrk_event_create(int device) {
  if (use_device(device)) return nullptr;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) {
    (void)hipGetLastError();
    fail(ABRK_ENODEV, "hipEventCreate failed");
    return nullptr;
  }
  return e;
}
extern "C" int abrk_event_destroy(int device, void* ev) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipEventDestroy((hipEvent_t)ev));
  return 0;
}
extern "C" int abrk_event_record(int device, void* ev, void* stream) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return 0;
}
extern "C" int abrk_event_elapsed_ms(int device, void* start, void* stop, float* ms) {
  if (int rc = use_device(device)) return rc;
  HIPCHK(hipEventSynchronize((hipEvent_t)stop));
  HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}

// ------------------------------------------------------------------------------- staging
namespace {
// Per-thread device scratch arena for host-pointer arguments (grown on demand, reused).
struct Arena {
  int device = -1;
  char* base = nullptr;
  size_t cap = 0, used = 0;
};
thread_local Arena t_arena;

// Small host batches (a control loop calling with B = 1 .. a few thousand rows) skip the copy engine: the
// arguments are packed by the CPU into a pinned, device-mapped arena that the kernel reads and writes directly
// over PCIe - one launch and one stream sync instead of 3-4 hipMemcpyAsync calls of a few hundred bytes each.
// Above ABRK_ZEROCOPY_MAX bytes (default 1 MiB) the copy engine + device scratch arena wins.
struct PinArena {
  char* base = nullptr;  // host address
  char* dev = nullptr;   // the same memory as the device sees it
  size_t cap = 0;
};
thread_local PinArena t_pin;
size_t zero_copy_max() {
  static const size_t v = [] {
    const char* e = getenv("ABRK_ZEROCOPY_MAX");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 20;
  }();
  return v;
}

// device-usable address of p, or nullptr for pageable host memory.  Pinned host memory (hipHostMalloc /
// hipHostRegister) is device-visible: it is passed through and read in place.
void* device_view(const void* p) {
  hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'ed host memory is "invalid value" to HIP
    return nullptr;
  }
  if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) return const_cast<void*>(p);
  if (at.type == hipMemoryTypeHost && at.devicePointer) return at.devicePointer;
  return nullptr;
}

bool recording();  // is this thread inside abrk_plan_begin .. abrk_plan_end?
struct Stager {
  int device;
  hipStream_t stream;
  struct Item {
    void* host;
    void* dev;
    size_t bytes;
    bool out, in;
  };
  std::vector<Item> items;
  size_t need = 0;
  bool staged = false, pinned = false;

  // first pass: register; second pass (after reserve()) resolve
  void* add(const void* p, size_t bytes, bool in, bool out) {
    if (!p) return nullptr;
    if (void* d = device_view(p)) return d;
    items.push_back({const_cast<void*>(p), nullptr, bytes, out, in});
    need += (bytes + 255) & ~size_t(255);
    return (void*)(uintptr_t)(items.size());  // placeholder index+1, resolved by fix()
  }
  int reserve() {
    if (items.empty()) return 0;
    staged = true;
    if (recording()) return fail(ABRK_EINVAL, "plans take device pointers only (got a host pointer)");
    if (need <= zero_copy_max()) {
      PinArena& pa = t_pin;
      if (pa.cap < need) {
        if (pa.base) (void)hipHostFree(pa.base);
        pa.base = pa.dev = nullptr;
        pa.cap = 0;
        size_t cap = need < (64u << 10) ? (64u << 10) : need;
        void* h = nullptr;
        hipError_t e = hipHostMalloc(&h, cap, hipHostMallocMapped | hipHostMallocPortable);
        void* d = nullptr;
        if (e == hipSuccess) e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
          (void)hipGetLastError();
          if (h) (void)hipHostFree(h);
          return fail(ABRK_ENOMEM, "pinned staging hipHostMalloc(%zu): %s", cap, hipGetErrorString(e));
        }
        pa.base = (char*)h;
        pa.dev = (char*)d;
        pa.cap = cap;
      }
      pinned = true;
      size_t off = 0;
      for (auto& it : items) {
        it.dev = pa.dev + off;
        if (it.in) memcpy(pa.base + off, it.host, it.bytes);
        off += (it.bytes + 255) & ~size_t(255);
      }
      return 0;
    }
    Arena& a = t_arena;
    if (a.device != device || a.cap < need) {
      if (a.base) {
        (void)hipSetDevice(a.device);
        (void)hipFree(a.base);
        (void)hipSetDevice(device);
      }
      a.base = nullptr;
      a.cap = 0;
      size_t cap = need < (1u << 20) ? (1u << 20) : need + need / 4;
      hipError_t e = hipMalloc((void**)&a.base, cap);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ABRK_ENOMEM, "staging hipMalloc(%zu): %s", cap, hipGetErrorString(e));
      }
      a.cap = cap;
      a.device = device;
    }
    size_t off = 0;
    for (auto& it : items) {
      it.dev = a.base + off;
      off += (it.bytes + 255) & ~size_t(255);
      if (it.in) {
        hipError_t e = hipMemcpyAsync(it.dev, it.host, it.bytes, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return fail(ABRK_ENODEV, "H2D staging: %s", hipGetErrorString(e));
      }
    }
    return 0;
  }
  // map a value returned by add() to the real device pointer
  template <class P>
  P fix(P p, const void* orig) const {
    if (!orig) return nullptr;
    for (size_t i = 0; i < items.size(); i++)
      if (items[i].host == orig && (uintptr_t)p == i + 1) return (P)items[i].dev;
    return p;
  }
  int finish() {
    if (!staged) return 0;
    if (pinned) {
      hipError_t e = hipStreamSynchronize(stream);
      if (e != hipSuccess) return fail(ABRK_ENODEV, "stream sync: %s", hipGetErrorString(e));
      for (auto& it : items)
        if (it.out) memcpy(it.host, t_pin.base + ((char*)it.dev - t_pin.dev), it.bytes);
      return 0;
    }
    for (auto& it : items)
      if (it.out) {
        hipError_t e = hipMemcpyAsync(it.host, it.dev, it.bytes, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return fail(ABRK_ENODEV, "D2H staging: %s", hipGetErrorString(e));
      }
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(ABRK_ENODEV, "stream sync: %s", hipGetErrorString(e));
    return 0;
  }
};

size_t esz(int dtype) { return dtype == ABRK_F64 ? 8 : 4; }

// ---- launch recording (abrk_plan_begin .. abrk_plan_end).  While a thread records, every abrk_*_batch call on it
// is validated and converted as usual but its kernel launch is kept as a closure instead of being enqueued; the
// closures own their parameter blocks and (for user arms) a private copy of the arm table.
struct Recorder {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<std::function<hipError_t()>> steps;
  std::vector<std::unique_ptr<std::vector<unsigned char>>> tables;  // stable addresses
  std::vector<void*> dev_bufs;  // device scratch owned by the plan (worklists of the six-row OSC kernels)
};
thread_local Recorder* t_rec = nullptr;
bool recording() { return t_rec != nullptr; }

const void* arm_table(const ArmEntry* a, int dtype) {
  if (!a || a->rt64.empty()) return nullptr;  // built-in and compiled arms carry their table in the kernels
  return dtype == ABRK_F64 ? (const void*)a->rt64.data() : (const void*)a->rt32.data();
}

// A caller's grip on its (device, stream) scratch slot of the six-row kernels (worklist cache, below): the slot's mutex
// AND a reference to the slot - a slot evicted or forgotten between the look-up and the lock stays alive (and simply
// leaves the cache) until the last holder lets go.
struct WorklistSlot;
struct WlHold {
  std::shared_ptr<WorklistSlot> slot;
  std::unique_lock<std::mutex> lk;
  void release() {
    if (lk.owns_lock()) lk.unlock();
    slot.reset();
  }
};

// enqueue now (and finish the staging), or keep the launch for the plan being recorded.  fn(arm_rt) launches the
// kernel; it captures its arguments by value.
template <class F>
int dispatch(Stager& st, const ArmEntry* a, int dtype, F&& fn, WlHold* held = nullptr) {
  if (Recorder* r = t_rec) {
    if (st.staged) return fail(ABRK_EINVAL, "plans take device pointers only (got a host pointer)");
    if (st.device != r->device || st.stream != r->stream)
      return fail(ABRK_EINVAL, "a recorded call must use the device and stream given to abrk_plan_begin");
    const void* rt = nullptr;
    if (a && !a->rt64.empty()) {
      r->tables.emplace_back(new std::vector<unsigned char>(dtype == ABRK_F64 ? a->rt64 : a->rt32));
      rt = r->tables.back()->data();
    }
    r->steps.emplace_back([fn, rt]() { return fn(rt); });
    return 0;
  }
  const hipError_t le = fn(arm_table(a, dtype));
  if (held) held->release();  // everything that uses the shared scratch is enqueued
  HIPCHK(le);
  return st.finish();
}

// Worklist of the six-row OSC kernels (rows whose law needs the truncating pseudo-inverse are deferred to a second pass,
// abrk_kernels.h): wl_ints(B) ~ B + 20 k ints of device scratch and - hand-over mode, batches of up to
// kHandoverMaxRows - a record of rec_len(n) values per list slot.  Immediate calls take both from a cache keyed by
// (device, stream) - calls on one stream are ordered, so the buffers can be reused - PROVIDED the launches that use
// them (counters' memset, pass 1, pass 2) enter the stream as a unit: several host threads may share a stream (every
// Python call without an explicit stream is on the NULL stream, and ctypes releases the GIL), and memset A, pass-1 A,
// memset B, pass-2 A would lose A's deferred rows (or overrun a list sized for one call).  So the caller keeps `hold` -
// the SLOT's mutex: calls on other streams and other devices do not wait for it - from here until its launches are
// enqueued (dispatch() releases it before any stream sync).  A grow frees the old buffers only after draining the
// slot's own stream, which - every earlier user having enqueued under the same lock - covers all of them; only callers
// of that one stream wait meanwhile.  The registry lock (g_wl_mu) is held for the look-up alone.  abrk_stream_destroy
// hands a stream's slot back; streams the caller destroyed behind the library's back are evicted once the cache is
// full (least recently used slot whose stream is idle or gone).  A recorded plan owns its worklist.
struct WorklistSlot {
  int device = 0;
  hipStream_t stream = nullptr;
  int* buf = nullptr;      // counters + row indices
  size_t cap = 0;          // bytes
  void* rec = nullptr;     // hand-over records
  size_t rec_cap = 0;      // bytes
  uint64_t last_use = 0;
  bool retired = false;    // left the cache (evicted / its stream destroyed): buffers released, never handed out again
  std::mutex mu;
};
std::mutex g_wl_mu;
// shared ownership: a caller copies the pointer under g_wl_mu and locks the slot afterwards - an eviction or
// abrk_stream_destroy in that window removes the slot from the cache but cannot free the mutex under the caller
std::vector<std::shared_ptr<WorklistSlot>> g_wl_cache;
uint64_t g_wl_clock = 0;
std::atomic<int64_t> g_wl_inline_fallbacks{0}, g_wl_evictions{0};
// batches up to here take the hand-over form of the six-row law (kHandoverMaxRows is what the finish kernel can scan).
// Measured on one box (UR5, all six rows, us per step, hand-over / recompute form): 4096 rows 17.1 / 20.9 (the recompute
// form runs one-pass below 16 k rows), 16 k 18.3 / 32.6, 32 k 23.0 / 33.8, 64 k 35.9 / 37.0, 128 k 47.2 / 38.8, 256 k 67.8 /
// 48.6 - beyond 64 k rows the deferred rows fill wavefronts, and re-running the row program on coalesced inputs beats
// reading 656-byte records one lane each.
constexpr int64_t kHandoverRows = 65536;
constexpr size_t kWlCacheSlots = 64;
// batches up to kHandoverRows (below) hand the deferred rows' intermediate results over to the second pass (43 MB of
// records for a six-joint arm in fp64); beyond, the second pass recomputes its rows

void wl_release(WorklistSlot& s) {  // caller holds s.mu; the slot's stream is drained
  (void)hipSetDevice(s.device);
  if (s.buf) (void)hipFree(s.buf);
  if (s.rec) (void)hipFree(s.rec);
  s.buf = nullptr;
  s.rec = nullptr;
  s.cap = s.rec_cap = 0;
}
// a slot that left the cache: wait for whoever is enqueueing on it, drain its stream if that still exists, free
void wl_retire(const std::shared_ptr<WorklistSlot>& s, bool stream_alive) {
  std::lock_guard<std::mutex> sl(s->mu);
  s->retired = true;
  if (stream_alive) {
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->stream);
    (void)hipGetLastError();
  }
  wl_release(*s);
}
// abrk_stream_destroy: the stream's slot leaves the cache (its launches are drained by the caller's hipStreamDestroy,
// which waits for the stream's work; the buffers are freed after an explicit drain here)
void wl_forget_stream(int device, hipStream_t stream) {
  std::shared_ptr<WorklistSlot> mine;
  {
    std::lock_guard<std::mutex> lk(g_wl_mu);
    for (size_t i = 0; i < g_wl_cache.size(); i++)
      if (g_wl_cache[i]->device == device && g_wl_cache[i]->stream == stream) {
        mine = std::move(g_wl_cache[i]);
        g_wl_cache.erase(g_wl_cache.begin() + i);
        break;
      }
  }
  if (mine) wl_retire(mine, true);  // (a caller that found the slot just before may still be enqueueing: s->mu)
}
// Beyond kHandoverRows and up to here the hand-over form continues with the DENSE finish kernel (abrk_law.hip
// osc6_finish_dense_kernel, round 6: 64 chunks' records numbered through, one record per lane) instead of the recompute
// pass.  Same box, UR5 six rows, us per step recompute / dense: 131 072 rows 36.9 / 29.5, 1 M rows 98.9 / 98.0, 8 M rows
// 708 / 758 (profiles/round6/dense_finish_ab.txt): while the records (656 B per deferred row, written 16 bytes at a time
// by ~3 lanes of a wavefront) stay in the caches they beat re-running the 1800-instruction kinematics; from HBM their
// traffic - 2 x 253 MB at 8 M rows - costs more than the recomputation.
constexpr int64_t kDenseFinishRows = 1 << 20;
int64_t dense_finish_max() {
  static const int64_t v = [] {  // measurement switch (0: recompute form beyond kHandoverRows, as until round 5)
    const char* e = measurement_env("ABRK_DENSE_MAX");
    return e ? (int64_t)atoll(e) : kDenseFinishRows;
  }();
  return v;
}
int64_t handover_max() {  // the largest batch whose records go to the per-chunk / grouped finish kernels
  static const int64_t v = [] {  // measurement switch
    const char* e = measurement_env("ABRK_HANDOVER_MAX");
    const int64_t x = e ? atoll(e) : kHandoverRows;
    return x < kHandoverMaxRows ? x : (int64_t)kHandoverMaxRows;
  }();
  return v;
}
bool handover_enabled() {
  static const bool off = measurement_env("ABRK_NO_HANDOVER") != nullptr;  // measurement switch: the round-3 scheme
  return !off;
}
// one of the slot's two buffers grown to `need` bytes (+ 25 %); the other one is left alone.  The slot's stream is
// drained first if the old buffer may still be in use (every earlier user enqueued under s->mu).  false: no memory
bool wl_grow(hipStream_t stream, void** buf, size_t* cap, size_t need, bool* drained) {
  if (*cap >= need) return true;
  if (*buf) {
    if (!*drained) (void)hipStreamSynchronize(stream);
    *drained = true;
    (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
  }
  void* p = nullptr;
  if (hipMalloc(&p, need + need / 4) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  *buf = p;
  *cap = need + need / 4;
  return true;
}
// -> 0 and *wl = the worklist (nullptr: run the sweeps inline), *rec = the record store (nullptr: recompute form), or
// an error code.  n / dtype: the arm's joint count and the arithmetic type (record size).
int worklist_for(int device, hipStream_t stream, int64_t B, int n, int dtype, int** wl, void** rec, WlHold& hold) {
  *wl = nullptr;
  *rec = nullptr;
  static const bool off = measurement_env("ABRK_NO_DEFER") != nullptr;  // measurement switch: sweeps inline, as before round 2
  // (row indices are parked as 32-bit ints: batches beyond 2^31 rows - they fit the 288 GB for fp32 arms - run inline)
  if (off || B > 0x7fffffffLL) return 0;
  const int64_t ho_max = handover_max();
  // below one wavefront of rows the second launch costs more than the eigen-decomposition it takes off the critical path
  // (a single state truncates in 4.6 % of the calls: 0.7 us expected, against ~4 us of launch + the finish kernel's scan)
  const bool handover = handover_enabled() && B >= 64 && (B <= ho_max || B <= dense_finish_max());
  // the recompute form below ~16 k rows: the second launch costs more than the divergence it removes (round 2)
  if (!handover && B < 16384) return 0;
  const size_t need = (size_t)wl_ints(B) * sizeof(int);
  // (hand-over mode keeps one 64-bit mask per 64-row chunk in `wl` - far less than the recompute form's lists)
  // (whole chunks: the finish kernel asks for a chunk's slot before it knows whether a record is there)
  const size_t need_rec = handover ? (size_t)((B + kBlock - 1) / kBlock * kBlock) * rec_len(n) * esz(dtype) : 0;
  if (Recorder* r = t_rec) {
    void *p = nullptr, *q = nullptr;
    hipError_t e = hipMalloc(&p, need);
    if (e == hipSuccess && need_rec) e = hipMalloc(&q, need_rec);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      if (p) (void)hipFree(p);
      return fail(ABRK_ENOMEM, "worklist of the recorded six-row OSC call, hipMalloc(%zu + %zu): %s", need, need_rec,
                  hipGetErrorString(e));
    }
    r->dev_bufs.push_back(p);
    if (q) r->dev_bufs.push_back(q);
    *wl = (int*)p;
    *rec = q;
    return 0;
  }
  for (int attempt = 0;; attempt++) {
    std::shared_ptr<WorklistSlot> s, evicted;
    bool evicted_stream_alive = false;
    {
      std::lock_guard<std::mutex> lk(g_wl_mu);
      for (auto& e : g_wl_cache)
        if (e->device == device && e->stream == stream) s = e;
      if (!s) {
        if (g_wl_cache.size() >= kWlCacheSlots) {
          // full: the least recently used slot whose stream has nothing in flight (or no longer exists) makes room
          size_t pick = g_wl_cache.size();
          bool pick_alive = false;
          for (size_t i = 0; i < g_wl_cache.size(); i++) {
            WorklistSlot& c = *g_wl_cache[i];
            if (pick < g_wl_cache.size() && c.last_use >= g_wl_cache[pick]->last_use) continue;
            if (g_wl_cache[i].use_count() > 1) continue;  // a caller holds (or is about to lock) it
            if (!c.mu.try_lock()) continue;               // somebody is enqueueing on it
            (void)hipSetDevice(c.device);
            const hipError_t q = c.stream ? hipStreamQuery(c.stream) : hipErrorNotReady;  // the NULL stream stays
            (void)hipGetLastError();
            c.mu.unlock();
            if (q != hipErrorNotReady) {
              pick = i;
              pick_alive = q == hipSuccess;
            }
          }
          (void)hipSetDevice(device);
          if (pick == g_wl_cache.size()) {  // every cached stream is busy: inline sweeps for this call
            g_wl_inline_fallbacks++;
            return 0;
          }
          g_wl_evictions++;
          evicted = std::move(g_wl_cache[pick]);
          evicted_stream_alive = pick_alive;
          g_wl_cache.erase(g_wl_cache.begin() + pick);
        }
        s = std::make_shared<WorklistSlot>();
        s->device = device;
        s->stream = stream;
        g_wl_cache.push_back(s);
      }
      s->last_use = ++g_wl_clock;
    }
    if (evicted) {
      wl_retire(evicted, evicted_stream_alive);
      (void)hipSetDevice(device);
    }
    hold.slot = s;
    hold.lk = std::unique_lock<std::mutex>(s->mu);
    if (s->retired) {  // evicted / forgotten between the look-up and the lock: look again (a fresh slot)
      hold.release();
      if (attempt < 4) continue;
      g_wl_inline_fallbacks++;
      return 0;
    }
    bool drained = false;
    void* b = s->buf;
    // the two buffers grow independently, and a recompute-form call (need_rec = 0) leaves the record store alone: a
    // stream alternating hand-over batches with large ones keeps both
    const bool ok = wl_grow(stream, &b, &s->cap, need, &drained) &&
                    (!need_rec || wl_grow(stream, &s->rec, &s->rec_cap, need_rec, &drained));
    s->buf = (int*)b;
    if (!ok) {
      hold.release();
      g_wl_inline_fallbacks++;
      return 0;  // no scratch on an immediate call: inline sweeps (same results)
    }
    *wl = s->buf;
    if (handover) *rec = s->rec;
    return 0;
  }
}

// finish kernel of the hand-over form: measurement switches (read once)
int env_int(const char* name, int dflt) {
  const char* e = measurement_env(name);
  return e ? atoi(e) : dflt;
}
// wavefronts per chunk and records per wavefront of the finish kernel (abrk_kernels.h finish_slots / finish_rounds);
// measurement switches, read once
int finish_slots_for(int64_t B) {
  static const int forced = env_int("ABRK_FINISH_SLOTS", 0);
  return forced >= 1 && forced <= kBlock ? forced : finish_slots((long)B);
}
// chunks per group of the grouped finish kernel (abrk_law.hip osc6_finish_group_kernel), 0 = the per-chunk kernel: the
// 16384-row band, where the per-chunk grid doubles working wavefronts up on SIMDs
int finish_group_for(int64_t B) {
  if (B > handover_max()) return -1;  // (beyond that the hand-over form only exists with the dense finish kernel: worklist_for)
  static const int forced = env_int("ABRK_FINISH_GROUP", -1);  // measurement switch: 0 = never grouped, 1..16 = always
  if (forced >= 0 && forced <= 16) return forced;
  const int64_t nchunk = (B + kBlock - 1) / kBlock;
  return nchunk > 128 && nchunk <= 256 ? 16 : 0;
}
int finish_rounds_for(int64_t B) {
  static const int forced = env_int("ABRK_FINISH_ROUNDS", -1);  // (0: every chunk goes one record per lane)
  return forced >= 0 ? (forced > kBlock ? kBlock : forced) : finish_rounds((long)B);
}

int check_common(int arm_id, int dtype, int64_t B, ArmEntry** a) {
  *a = get_arm(arm_id);
  if (!*a) return fail(ABRK_ENOARM, "unknown arm id %d", arm_id);
  if (dtype != ABRK_F64 && dtype != ABRK_F32) return fail(ABRK_EINVAL, "dtype %d is not ABRK_F64/ABRK_F32", dtype);
  if (B < 0) return fail(ABRK_EINVAL, "negative batch %lld", (long long)B);
  return 0;
}

}  // namespace

extern "C" int abrk_scratch_stats(int device, abrk_scratch_info* out) {
  if (!out) return fail(ABRK_EINVAL, "out is NULL");
  memset(out, 0, sizeof *out);
  {
    std::lock_guard<std::mutex> lk(g_wl_mu);
    out->worklist_slots = (int64_t)g_wl_cache.size();
    for (auto& e : g_wl_cache) out->worklist_bytes += (int64_t)(e->cap + e->rec_cap);  // (racy read of sizes: diagnostics)
  }
  out->inline_fallbacks = g_wl_inline_fallbacks.load();
  out->evictions = g_wl_evictions.load();
  {
    StatusPool& sp = status_pool();
    std::lock_guard<std::mutex> lk(sp.mu);
    out->status_words_out = sp.handed_out;
    out->status_blocks = sp.blocks;
  }
  if (int rc = use_device(device)) return rc;
  size_t fr = 0, tot = 0;
  HIPCHK(hipMemGetInfo(&fr, &tot));
  out->device_free_bytes = (int64_t)fr;
  out->device_total_bytes = (int64_t)tot;
  return 0;
}

// ------------------------------------------------------------------------------- dynamics
extern "C" int abrk_dynamics_batch(int arm_id, int dtype, int64_t B, const void* q, const void* dq, int frame,
                                   const double* x_off, uint32_t want, const abrk_dyn_out* out, int device,
                                   void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (frame < 0 || frame > 2 * n + 1) return fail(ABRK_EFRAME, "Invalid transformation name: frame id %d", frame);
  if (!out || !want) return fail(ABRK_EINVAL, "nothing requested");
  if (want >> 10) return fail(ABRK_EINVAL, "unknown want bits 0x%x", want);
  if (!q) return fail(ABRK_EINVAL, "q is NULL");
  if ((want & (ABRK_WANT_C | ABRK_WANT_DJ)) && !dq) return fail(ABRK_EINVAL, "C / dJ need dq");
  void* const* outs = reinterpret_cast<void* const*>(out);
  const size_t per[10] = {3, (size_t)6 * n, (size_t)n * n, (size_t)n, (size_t)n * n, (size_t)6 * n, 9, 16, 16, 4};
  for (int i = 0; i < 10; i++)
    if ((want >> i & 1) && !outs[i]) return fail(ABRK_EINVAL, "output %d requested but its pointer is NULL", i);
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  DynArgs da;
  memset(&da, 0, sizeof da);
  const void* q_ = st.add(q, B * n * s, true, false);
  const void* dq_ = (want & (ABRK_WANT_C | ABRK_WANT_DJ)) ? st.add(dq, B * n * s, true, false) : nullptr;
  void* o_[10];
  for (int i = 0; i < 10; i++) o_[i] = (want >> i & 1) ? st.add(outs[i], B * per[i] * s, false, true) : nullptr;
  if (int rc = st.reserve()) return rc;
  da.q = st.fix(q_, q);
  da.dq = dq_ ? st.fix(dq_, dq) : nullptr;
  for (int i = 0; i < 10; i++) da.out[i] = (want >> i & 1) ? st.fix(o_[i], outs[i]) : nullptr;
  da.frame = frame;
  da.m = frame_m(frame, n);
  for (int r = 0; r < 3; r++) da.off[r] = x_off ? x_off[r] : 0.0;
  da.want = want;
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, a, dtype, [=](const void* rt) { return ops->dyn(dtype, LaunchArgs{rt, (long)B, hs}, da); });
}

// ------------------------------------------------------------------------------- OSC
static int osc_generate_impl(int arm_id, int dtype, const abrk_osc_params* P, int64_t B, const void* q,
                             const void* dq, const void* target, const void* target_velocity,
                             void* integrated_error, const void* u_null_ext, void* u, void* training_signal,
                             uint32_t want, const abrk_dyn_out* out, int device, void* stream);

extern "C" int abrk_osc_generate_batch(int arm_id, int dtype, const abrk_osc_params* P, int64_t B, const void* q,
                                       const void* dq, const void* target, const void* target_velocity,
                                       void* integrated_error, const void* u_null_ext, void* u,
                                       void* training_signal, int device, void* stream) {
  return osc_generate_impl(arm_id, dtype, P, B, q, dq, target, target_velocity, integrated_error, u_null_ext, u,
                           training_signal, 0, nullptr, device, stream);
}

extern "C" int abrk_osc_generate_full_batch(int arm_id, int dtype, const abrk_osc_params* P, int64_t B, const void* q,
                                            const void* dq, const void* target, const void* target_velocity,
                                            void* integrated_error, const void* u_null_ext, void* u,
                                            void* training_signal, uint32_t want, const abrk_dyn_out* out, int device,
                                            void* stream) {
  const uint32_t ok = ABRK_WANT_TX | ABRK_WANT_J | ABRK_WANT_M | ABRK_WANT_G | ABRK_WANT_C | ABRK_WANT_DJ;
  if (!want || (want & ~ok))
    return fail(ABRK_EINVAL, "want must be a non-empty subset of Tx | J | M | g | C | dJ (0x%x)", want);
  if (!out) return fail(ABRK_EINVAL, "out is NULL");
  return osc_generate_impl(arm_id, dtype, P, B, q, dq, target, target_velocity, integrated_error, u_null_ext, u,
                           training_signal, want, out, device, stream);
}

static int osc_generate_impl(int arm_id, int dtype, const abrk_osc_params* P, int64_t B, const void* q,
                             const void* dq, const void* target, const void* target_velocity,
                             void* integrated_error, const void* u_null_ext, void* u, void* training_signal,
                             uint32_t want, const abrk_dyn_out* out, int device, void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (!P) return fail(ABRK_EINVAL, "params is NULL");
  if (P->ref_frame < 0 || P->ref_frame > 2 * n + 1)
    return fail(ABRK_EFRAME, "Invalid transformation name: frame id %d", P->ref_frame);
  if (P->n_null < 0 || P->n_null > ABRK_MAX_NULL) return fail(ABRK_EINVAL, "n_null=%d outside 0..%d", P->n_null, ABRK_MAX_NULL);
  for (int c = 0; c < P->n_null; c++)
    if (P->null_ctrl[c].kind != ABRK_NULL_DAMPING && P->null_ctrl[c].kind != ABRK_NULL_RESTING)
      return fail(ABRK_EINVAL, "null controller %d has unknown kind %d", c, P->null_ctrl[c].kind);
  if (P->orientation_algorithm != 0 && P->orientation_algorithm != 1)
    return fail(ABRK_EINVAL, "Invalid algorithm number %d for calculating orientation error", P->orientation_algorithm);
  int k = 0;
  for (int r = 0; r < 6; r++) k += P->ctrlr_dof[r] ? 1 : 0;
  if (k == 0) return fail(ABRK_EINVAL, "ctrlr_dof selects no task-space dimension");
  if (!q || !dq || !target || !u) return fail(ABRK_EINVAL, "q, dq, target and u are required");
  if (P->ki != 0 && !integrated_error) return fail(ABRK_EINVAL, "ki != 0 needs the integrated_error state array");
  void* const wout[6] = {out ? out->Tx : nullptr, out ? out->J : nullptr, out ? out->M : nullptr, out ? out->g : nullptr,
                         out ? out->C : nullptr,  out ? out->dJ : nullptr};
  for (int i = 0; i < 6; i++)
    if ((want >> i & 1) && !wout[i]) return fail(ABRK_EINVAL, "output %d requested but its pointer is NULL", i);
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  void* ie = (P->ki != 0) ? integrated_error : nullptr;
  const void* q_ = st.add(q, B * n * s, true, false);
  const void* dq_ = st.add(dq, B * n * s, true, false);
  const void* t_ = st.add(target, B * 6 * s, true, false);
  const void* tv_ = st.add(target_velocity, B * 6 * s, true, false);
  void* ie_ = st.add(ie, B * 6 * s, true, true);
  const void* une_ = st.add(u_null_ext, B * n * s, true, false);
  void* u_ = st.add(u, B * n * s, false, true);
  void* ts_ = st.add(training_signal, B * n * s, false, true);
  const size_t per[6] = {3, (size_t)6 * n, (size_t)n * n, (size_t)n, (size_t)n * n, (size_t)6 * n};
  void* o_[6];
  for (int i = 0; i < 6; i++) o_[i] = (want >> i & 1) ? st.add(wout[i], B * per[i] * s, false, true) : nullptr;
  if (int rc = st.reserve()) return rc;
  OscArgs oa;
  oa.want = want;
  for (int i = 0; i < 6; i++) oa.out[i] = (want >> i & 1) ? st.fix(o_[i], wout[i]) : nullptr;
  oa.q = st.fix(q_, q);
  oa.dq = st.fix(dq_, dq);
  oa.target = st.fix(t_, target);
  oa.tv = st.fix(tv_, target_velocity);
  oa.ierr = st.fix(ie_, ie);
  oa.une = st.fix(une_, u_null_ext);
  oa.u = st.fix(u_, u);
  oa.ts = st.fix(ts_, training_signal);
  oa.use_C = P->use_C ? 1 : 0;
  oa.fast = osc_fast_rows(*P, n, u_null_ext != nullptr);
  WlHold wl_hold;  // the (device, stream) worklist stays ours until the launches are enqueued
  if (oa.fast == 0 && !want)
    if (int rc = worklist_for(device, (hipStream_t)stream, B, n, dtype, &oa.wl, &oa.rec, wl_hold)) return rc;
  OscP<double> p64 = make_oscp<double>(*P, n);
  OscP<float> p32 = make_oscp<float>(*P, n);
  StatusWord* sw = status_word(st.staged, device, stream);
  p64.status = p32.status = sw ? sw->dev : nullptr;
  if (sw && st.staged) *sw->host = 0;  // (this thread's word: nothing of an earlier, failed call is left in it)
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  // hand-over mode: the arm's first pass, then the arm-independent finish kernel on the records it left
  FinishArgs fa{oa.wl, oa.rec, (P->n_null > 0 || u_null_ext) ? 1 : 0, finish_slots_for(B), finish_rounds_for(B),
                oa.u, oa.ts, finish_group_for(B)};
  const int rc = dispatch(st, a, dtype, [=](const void* rt) {
    OscArgs o = oa;
    o.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    const LaunchArgs la{rt, (long)B, hs};
    const hipError_t e = ops->osc(dtype, la, o);
    if (e != hipSuccess || !o.rec) return e;
    return launch_osc6_finish(n, dtype, la, fa);
  }, &wl_hold);
  // host arrays: the call is complete (outputs copied back) - report a non-positive-definite M now
  if (rc == 0 && st.staged && sw && sw->take()) return singular_error();
  return rc;
}

// ------------------------------------------------------------------------------- OSC, wave-cooperative mapping
namespace abrk {
hipError_t launch_osc_coop_ur5(const LaunchArgs& la, const CoopArgs& a);
}
extern "C" int abrk_osc_generate_coop_batch(int arm_id, int dtype, const abrk_osc_params* P, int64_t B, const void* q,
                                            const void* dq, const void* target, void* u, void* training_signal,
                                            int lanes_per_arm, int device, void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (!(a->builtin && strcmp(a->desc.name, "ur5") == 0) || dtype != ABRK_F64)
    return fail(ABRK_EINVAL, "the wave-cooperative kernels are built for the built-in ur5 arm in fp64 (measurement variant)");
  if (lanes_per_arm != 4 && lanes_per_arm != 8 && lanes_per_arm != 16)
    return fail(ABRK_EINVAL, "lanes_per_arm must be 4, 8 or 16");
  if (!P) return fail(ABRK_EINVAL, "params is NULL");
  if (osc_fast_rows(*P, n, false) != 3 || P->n_null != 0 || P->use_C || P->ki != 0 || P->xyz_offset[0] != 0 ||
      P->xyz_offset[1] != 0 || P->xyz_offset[2] != 0)
    return fail(ABRK_EINVAL, "the wave-cooperative kernels cover the plain law: x,y,z of the EE, no offset, no null "
                             "controllers, no Coriolis term, ki = 0");
  if (!q || !dq || !target || !u) return fail(ABRK_EINVAL, "q, dq, target and u are required");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* q_ = st.add(q, B * n * s, true, false);
  const void* dq_ = st.add(dq, B * n * s, true, false);
  const void* t_ = st.add(target, B * 6 * s, true, false);
  void* u_ = st.add(u, B * n * s, false, true);
  void* ts_ = st.add(training_signal, B * n * s, false, true);
  if (int rc = st.reserve()) return rc;
  CoopArgs ca;
  ca.P = nullptr;
  ca.lanes = lanes_per_arm;
  ca.q = st.fix(q_, q);
  ca.dq = st.fix(dq_, dq);
  ca.target = st.fix(t_, target);
  ca.u = st.fix(u_, u);
  ca.ts = st.fix(ts_, training_signal);
  const OscP<double> p64 = make_oscp<double>(*P, n);
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, a, dtype, [=](const void*) {
    CoopArgs o = ca;
    o.P = &p64;
    return launch_osc_coop_ur5(LaunchArgs{nullptr, (long)B, hs}, o);
  });
}

// ------------------------------------------------------------------------------- OSC over several devices
// One call, host arrays, all the devices the caller names: the batch is cut into contiguous row shards
// (sizes differing by at most one row; BASELINE config 4: 2^20 rows over 8 GPUs), each shard staged to its device's scratch
// arena, evaluated by the same kernel on that device's own stream, and copied back.  Rows are independent: there is no
// exchange step and therefore no collective.  Round 6: the shards of one DEVICE are one unit of work - staged in,
// launched and collected by one host thread per device (the calling thread takes the first device, short-lived workers
// the others), under that device's lock only: the staging copies (pageable host memory: synchronous on the thread that
// issues them) of different devices run side by side, and calls on disjoint device sets do not wait for each other.
// Callers that keep their shards RESIDENT use the abrk_*_resident entry points below (device pointer tables, enqueue
// only) or one process per GPU (bench.py --gpus N).
namespace {
struct ShardCtx {
  hipStream_t stream = nullptr;
  char* base = nullptr;
  size_t cap = 0;
};
constexpr int kMaxShardDevices = 64;
std::mutex g_shard_mu;                       // the registry below (look-ups only)
std::mutex g_shard_dev_mu[kMaxShardDevices];  // the scratch arenas of one device's contexts (held while a call uses them)
// (device, slot on that device) -> context; heap-allocated: the pointers handed out stay valid as the table grows
std::vector<std::pair<std::pair<int, int>, std::unique_ptr<ShardCtx>>> g_shard_ctx;

// the context of (device, slot), its stream created; the calling thread's current device must be `device`
ShardCtx* shard_ctx_find(int device, int slot) {
  std::lock_guard<std::mutex> lk(g_shard_mu);
  ShardCtx* c = nullptr;
  for (auto& e : g_shard_ctx)
    if (e.first.first == device && e.first.second == slot) c = e.second.get();
  if (!c) {
    g_shard_ctx.emplace_back(std::make_pair(device, slot), std::unique_ptr<ShardCtx>(new ShardCtx));
    c = g_shard_ctx.back().second.get();
  }
  if (!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return c;
}
// ... with at least `need` bytes of device scratch (caller holds g_shard_dev_mu[device])
ShardCtx* shard_ctx(int device, int slot, size_t need) {
  ShardCtx* c = shard_ctx_find(device, slot);
  if (!c) return nullptr;
  if (c->cap < need) {
    if (c->base) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipFree(c->base);
    }
    c->base = nullptr;
    c->cap = 0;
    const size_t cap = need + need / 4;
    if (hipMalloc((void**)&c->base, cap) != hipSuccess) return nullptr;
    c->cap = cap;
  }
  return c;
}
}  // namespace

namespace {
// One host batch over several devices: [0, B) is cut into n_shards contiguous row ranges (sizes differing by at most one
// row - abr_control_amd/sharding.py shard_range), the pieces of shard g are staged to devices[g] on a stream of its own,
// `launch(dev, rows, stream)` enqueues the kernel(s) on it (dev[k]: device address of piece k, null for an absent one),
// and only when every shard of a device is in flight are that device's results collected.  No collective: rows are
// independent.
struct ShardPiece {
  const void* host_in;
  void* host_out;
  size_t per_row;  // bytes
};
template <class Launch>
int run_sharded(int64_t B, int n_shards, const int* devices, const ShardPiece* pieces, int n_pieces, Launch&& launch) {
  if (n_shards < 1 || !devices) return fail(ABRK_EINVAL, "n_shards must be >= 1 and devices non-NULL");
  if (recording()) return fail(ABRK_EINVAL, "a sharded entry point cannot be recorded into a plan");
  if (B < 0) return fail(ABRK_EINVAL, "negative batch %lld", (long long)B);
  if (B == 0) return 0;
  for (int k = 0; k < n_pieces; k++) {
    const void* ptrs[2] = {pieces[k].host_in, pieces[k].host_out};
    for (const void* p : ptrs) {
      hipPointerAttribute_t at;
      if (p && hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice)
        return fail(ABRK_EINVAL, "the *_sharded entry points take host arrays (a device pointer lives on one device); "
                                 "shards that live on the devices go through the *_resident entry points");
      (void)hipGetLastError();  // plain malloc'ed memory is "invalid value" to HIP
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(ABRK_ENODEV, "no HIP device available; libabrk has no CPU fallback");
  }
  for (int g = 0; g < n_shards; g++)
    if (devices[g] < 0 || devices[g] >= ndev || devices[g] >= kMaxShardDevices)
      return fail(ABRK_EINVAL, "device %d outside 0..%d", devices[g], ndev - 1);
  constexpr int kMaxPieces = 16;
  if (n_pieces > kMaxPieces) return fail(ABRK_EINVAL, "too many arrays for one sharded call");
  struct Shard {
    ShardCtx* c;
    int device, slot, index;
    int64_t r0, rows;
    char* dev[kMaxPieces];
  };
  std::vector<Shard> shards;
  std::vector<int> per_dev(ndev, 0), order;  // order: the distinct devices, as first named
  for (int g = 0; g < n_shards; g++) {
    const int64_t base = B / n_shards, extra = B % n_shards;
    const int64_t r0 = g * base + (g < extra ? g : extra), r1 = r0 + base + (g < extra ? 1 : 0);
    if (r1 == r0) continue;
    Shard sh{};
    sh.device = devices[g];
    sh.index = g;
    sh.r0 = r0;
    sh.rows = r1 - r0;
    if (per_dev[sh.device] == 0) order.push_back(sh.device);
    sh.slot = per_dev[sh.device]++;
    shards.push_back(sh);
  }
  // everything one device has to do (its lock held throughout: the contexts' scratch arenas are per (device, slot))
  auto device_work = [&](int device) -> int {
    std::lock_guard<std::mutex> lk(g_shard_dev_mu[device]);
    HIPCHK(hipSetDevice(device));
    t_current_device = device;
    for (Shard& sh : shards) {
      if (sh.device != device) continue;
      size_t need = 0;
      for (int k = 0; k < n_pieces; k++)
        if (pieces[k].host_in || pieces[k].host_out) need += (sh.rows * pieces[k].per_row + 255) & ~size_t(255);
      sh.c = shard_ctx(device, sh.slot, need);
      if (!sh.c) {
        (void)hipGetLastError();
        return fail(ABRK_ENOMEM, "shard %d: stream / %zu bytes of scratch on device %d", sh.index, need, device);
      }
      size_t off = 0;
      for (int k = 0; k < n_pieces; k++) {
        const ShardPiece& pc = pieces[k];
        sh.dev[k] = nullptr;
        if (!pc.host_in && !pc.host_out) continue;
        sh.dev[k] = sh.c->base + off;
        off += (sh.rows * pc.per_row + 255) & ~size_t(255);
        if (pc.host_in)
          HIPCHK(hipMemcpyAsync(sh.dev[k], (const char*)pc.host_in + sh.r0 * pc.per_row, sh.rows * pc.per_row,
                                hipMemcpyHostToDevice, sh.c->stream));
      }
      HIPCHK(launch(sh.dev, sh.rows, sh.c->stream));
    }
    // every kernel of this device is enqueued; now collect (a device-to-host copy into pageable memory waits for its shard)
    for (Shard& sh : shards) {
      if (sh.device != device) continue;
      for (int k = 0; k < n_pieces; k++)
        if (pieces[k].host_out && sh.dev[k])
          HIPCHK(hipMemcpyAsync((char*)pieces[k].host_out + sh.r0 * pieces[k].per_row, sh.dev[k],
                                sh.rows * pieces[k].per_row, hipMemcpyDeviceToHost, sh.c->stream));
      HIPCHK(hipStreamSynchronize(sh.c->stream));
    }
    return 0;
  };
  if (order.size() == 1) return device_work(order[0]);
  // several devices: one host thread each (the caller's own for the first); a worker's error text travels back with its code
  std::vector<int> rcs(order.size(), 0);
  std::vector<std::string> msgs(order.size());
  std::vector<std::thread> workers;
  for (size_t i = 1; i < order.size(); i++)
    workers.emplace_back([&, i] {
      rcs[i] = device_work(order[i]);
      if (rcs[i]) msgs[i] = g_err;
    });
  rcs[0] = device_work(order[0]);
  if (rcs[0]) msgs[0] = g_err;
  for (auto& w : workers) w.join();
  (void)hipSetDevice(order[0]);
  t_current_device = order[0];
  for (size_t i = 0; i < order.size(); i++)
    if (rcs[i]) {
      g_err = msgs[i];
      return rcs[i];
    }
  return 0;
}
}  // namespace

extern "C" int abrk_osc_generate_sharded(int arm_id, int dtype, const abrk_osc_params* P, int64_t B, const void* q,
                                         const void* dq, const void* target, const void* target_velocity,
                                         void* integrated_error, const void* u_null_ext, void* u,
                                         void* training_signal, int n_shards, const int* devices) {
  if (n_shards < 1 || !devices) return fail(ABRK_EINVAL, "n_shards must be >= 1 and devices non-NULL");
  // argument checks of the single-device entry point (B = 0 returns right after them)
  if (int rc = osc_generate_impl(arm_id, dtype, P, 0, q, dq, target, target_velocity, integrated_error, u_null_ext, u,
                                 training_signal, 0, nullptr, devices[0], nullptr))
    return rc;
  ArmEntry* a = get_arm(arm_id);
  const int n = a->desc.n_joints;
  const size_t s = esz(dtype);
  void* ie = (P->ki != 0) ? integrated_error : nullptr;
  OscP<double> p64 = make_oscp<double>(*P, n);
  OscP<float> p32 = make_oscp<float>(*P, n);
  // (a synchronous call: the calling thread's word, shared by every shard - the word is portable pinned host memory)
  StatusWord* sw = status_word(true, devices[0]);
  p64.status = p32.status = sw ? sw->dev : nullptr;
  if (sw) *sw->host = 0;
  const ShardPiece pieces[8] = {{q, nullptr, n * s},           {dq, nullptr, n * s},        {target, nullptr, 6 * s},
                                {target_velocity, nullptr, 6 * s}, {ie, ie, 6 * s},          {u_null_ext, nullptr, n * s},
                                {nullptr, u, n * s},           {nullptr, training_signal, n * s}};
  const int use_C = P->use_C ? 1 : 0, fast = osc_fast_rows(*P, n, u_null_ext != nullptr);
  const int rc = run_sharded(B, n_shards, devices, pieces, 8, [&](char* const* dev, int64_t rows, hipStream_t st) {
    OscArgs oa;
    oa.q = dev[0];
    oa.dq = dev[1];
    oa.target = dev[2];
    oa.tv = dev[3];
    oa.ierr = dev[4];
    oa.une = dev[5];
    oa.u = dev[6];
    oa.ts = dev[7];
    oa.use_C = use_C;
    oa.fast = fast;
    oa.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return a->ops->osc(dtype, LaunchArgs{arm_table(a, dtype), (long)rows, st}, oa);
  });
  if (rc == 0 && sw && sw->take()) return singular_error();
  return rc;
}

extern "C" int abrk_sliding_generate_sharded(int arm_id, int dtype, const abrk_sliding_params* P, int64_t B,
                                             const void* q, const void* dq, const void* target,
                                             const void* target_velocity, const void* target_acc, void* u, void* s_out,
                                             int n_shards, const int* devices) {
  if (n_shards < 1 || !devices) return fail(ABRK_EINVAL, "n_shards must be >= 1 and devices non-NULL");
  if (int rc = abrk_sliding_generate_batch(arm_id, dtype, P, 0, q, dq, target, target_velocity, target_acc, u, s_out,
                                           devices[0], nullptr))
    return rc;
  ArmEntry* a = get_arm(arm_id);
  const int n = a->desc.n_joints, nt = P->cartesian ? 3 : n;
  const size_t s = esz(dtype);
  const SlidingP<double> p64 = make_slidingp<double>(*P, n);
  const SlidingP<float> p32 = make_slidingp<float>(*P, n);
  const ShardPiece pieces[7] = {{q, nullptr, n * s},  {dq, nullptr, n * s},         {target, nullptr, nt * s},
                                {target_velocity, nullptr, nt * s}, {target_acc, nullptr, nt * s},
                                {nullptr, u, n * s},  {nullptr, s_out, n * s}};
  return run_sharded(B, n_shards, devices, pieces, 7, [&](char* const* dev, int64_t rows, hipStream_t st) {
    SlidingArgs sa;
    sa.q = dev[0];
    sa.dq = dev[1];
    sa.target = dev[2];
    sa.tv = dev[3];
    sa.ta = dev[4];
    sa.u = dev[5];
    sa.s = dev[6];
    sa.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return a->ops->sliding(dtype, LaunchArgs{arm_table(a, dtype), (long)rows, st}, sa);
  });
}

extern "C" int abrk_joint_generate_sharded(int arm_id, int dtype, const abrk_null_ctrl* ctrl, int account_for_gravity,
                                           int64_t B, const void* q, const void* dq, const void* target,
                                           const void* target_velocity, void* u, int n_shards, const int* devices) {
  if (n_shards < 1 || !devices) return fail(ABRK_EINVAL, "n_shards must be >= 1 and devices non-NULL");
  if (int rc = abrk_joint_generate_batch(arm_id, dtype, ctrl, account_for_gravity, 0, q, dq, target, target_velocity, u,
                                         devices[0], nullptr))
    return rc;
  ArmEntry* a = get_arm(arm_id);
  const int n = a->desc.n_joints;
  const size_t s = esz(dtype);
  const JointP<double> p64 = make_jointp<double>(*ctrl, account_for_gravity);
  const JointP<float> p32 = make_jointp<float>(*ctrl, account_for_gravity);
  const ShardPiece pieces[5] = {{q, nullptr, n * s}, {dq, nullptr, n * s}, {target, nullptr, n * s},
                                {target_velocity, nullptr, n * s}, {nullptr, u, n * s}};
  return run_sharded(B, n_shards, devices, pieces, 5, [&](char* const* dev, int64_t rows, hipStream_t st) {
    JointArgs ja;
    ja.q = dev[0];
    ja.dq = dev[1];
    ja.target = dev[2];
    ja.tv = dev[3];
    ja.u = dev[4];
    ja.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return a->ops->joint(dtype, LaunchArgs{arm_table(a, dtype), (long)rows, st}, ja);
  });
}

extern "C" int abrk_dynamics_sharded(int arm_id, int dtype, int64_t B, const void* q, const void* dq, int frame,
                                     const double* x_off, uint32_t want, const abrk_dyn_out* out, int n_shards,
                                     const int* devices) {
  if (n_shards < 1 || !devices) return fail(ABRK_EINVAL, "n_shards must be >= 1 and devices non-NULL");
  if (int rc = abrk_dynamics_batch(arm_id, dtype, 0, q, dq, frame, x_off, want, out, devices[0], nullptr)) return rc;
  ArmEntry* a = get_arm(arm_id);
  const int n = a->desc.n_joints;
  const size_t s = esz(dtype);
  void* const* outs = reinterpret_cast<void* const*>(out);
  const size_t per[10] = {3, (size_t)6 * n, (size_t)n * n, (size_t)n, (size_t)n * n, (size_t)6 * n, 9, 16, 16, 4};
  const bool vel = (want & (ABRK_WANT_C | ABRK_WANT_DJ)) != 0;
  ShardPiece pieces[12] = {{q, nullptr, n * s}, {vel ? dq : nullptr, nullptr, n * s}};
  for (int i = 0; i < 10; i++) pieces[2 + i] = {nullptr, (want >> i & 1) ? outs[i] : nullptr, per[i] * s};
  DynArgs da0;
  memset(&da0, 0, sizeof da0);
  da0.frame = frame;
  da0.m = frame_m(frame, n);
  for (int r = 0; r < 3; r++) da0.off[r] = x_off ? x_off[r] : 0.0;
  da0.want = want;
  return run_sharded(B, n_shards, devices, pieces, 12, [&](char* const* dev, int64_t rows, hipStream_t st) {
    DynArgs da = da0;
    da.q = dev[0];
    da.dq = dev[1];
    for (int i = 0; i < 10; i++) da.out[i] = dev[2 + i];
    return a->ops->dyn(dtype, LaunchArgs{arm_table(a, dtype), (long)rows, st}, da);
  });
}

// ------------------------------------------------------------------------------- resident shards
// "Results remain in per-device buffers unless the caller asks for host arrays": a batch whose shards LIVE on the
// devices.  Every array argument is a table of n_shards device pointers (shard g: rows[g] rows on devices[g]); a call
// only ENQUEUES - shard g's kernels on shard g's stream, nothing is staged, nothing is waited for - by going through the
// single-device entry point of the same name once per shard (so everything that holds there holds here: the six-row
// law's (device, stream) scratch, the per-(device, stream) ABRK_ESINGULAR word, per-row state staying with its shard).
// abrk_shards_sync drains the shards' streams.  One host thread, N GPUs, no collective.
namespace {
int check_cut(const abrk_shard_cut* cut) {
  if (!cut || cut->n_shards < 1 || !cut->devices || !cut->rows)
    return fail(ABRK_EINVAL, "shard cut: n_shards >= 1, devices and rows are required");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(ABRK_ENODEV, "no HIP device available; libabrk has no CPU fallback");
  }
  for (int g = 0; g < cut->n_shards; g++) {
    if (cut->devices[g] < 0 || cut->devices[g] >= ndev || cut->devices[g] >= kMaxShardDevices)
      return fail(ABRK_EINVAL, "shard %d: device %d outside 0..%d", g, cut->devices[g], ndev - 1);
    if (cut->rows[g] < 0) return fail(ABRK_EINVAL, "shard %d: negative row count", g);
  }
  return 0;
}
// the stream shard g runs on: the caller's, or the library's stream of (device, k) for the k-th shard named on that device
int cut_stream(const abrk_shard_cut* cut, int g, void** out) {
  if (cut->streams) {
    *out = cut->streams[g];
    return 0;
  }
  int slot = 0;
  for (int h = 0; h < g; h++) slot += cut->devices[h] == cut->devices[g] ? 1 : 0;
  if (int rc = use_device(cut->devices[g])) return rc;
  ShardCtx* c = shard_ctx_find(cut->devices[g], slot);
  if (!c) {
    (void)hipGetLastError();
    return fail(ABRK_ENODEV, "shard %d: no stream on device %d", g, cut->devices[g]);
  }
  *out = c->stream;
  return 0;
}
// entry g of a pointer table (a NULL table: the array is absent in every shard)
template <class P>
P tab_at(P const* tab, int g) {
  return tab ? tab[g] : nullptr;
}
// a resident array must be memory the device reads in place: device memory or pinned host memory
int check_resident(int g, const char* name, const void* p) {
  if (!p) return 0;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ABRK_EINVAL, "shard %d: %s is a pageable host pointer - the *_resident entry points take device "
                             "pointers (host arrays go through the *_sharded entry points)", g, name);
  }
  return 0;
}
template <class Call>
int for_each_shard(const abrk_shard_cut* cut, Call&& call) {
  if (int rc = check_cut(cut)) return rc;
  for (int g = 0; g < cut->n_shards; g++) {
    if (cut->rows[g] == 0) continue;
    void* stream = nullptr;
    if (int rc = cut_stream(cut, g, &stream)) return rc;
    if (int rc = call(g, cut->devices[g], cut->rows[g], stream)) return rc;
  }
  return 0;
}
}  // namespace

extern "C" void* abrk_shard_stream(int device, int slot) {
  if (slot < 0 || device >= kMaxShardDevices || use_device(device)) return nullptr;
  ShardCtx* c = shard_ctx_find(device, slot);
  if (!c) {
    (void)hipGetLastError();
    fail(ABRK_ENODEV, "no stream for (device %d, slot %d)", device, slot);
    return nullptr;
  }
  return c->stream;
}

extern "C" int abrk_osc_generate_resident(int arm_id, int dtype, const abrk_osc_params* P, const abrk_shard_cut* cut,
                                          const void* const* q, const void* const* dq, const void* const* target,
                                          const void* const* target_velocity, void* const* integrated_error,
                                          const void* const* u_null_ext, void* const* u, void* const* training_signal) {
  if (!q || !dq || !target || !u) return fail(ABRK_EINVAL, "q, dq, target and u are required");
  return for_each_shard(cut, [&](int g, int device, int64_t rows, void* stream) -> int {
    const struct { const char* name; const void* p; } arrs[] = {
        {"q", q[g]}, {"dq", dq[g]}, {"target", target[g]}, {"target_velocity", tab_at(target_velocity, g)},
        {"integrated_error", tab_at(integrated_error, g)}, {"u_null_ext", tab_at(u_null_ext, g)}, {"u", u[g]},
        {"training_signal", tab_at(training_signal, g)}};
    for (auto& a : arrs)
      if (int rc = check_resident(g, a.name, a.p)) return rc;
    return osc_generate_impl(arm_id, dtype, P, rows, q[g], dq[g], target[g], tab_at(target_velocity, g),
                             tab_at(integrated_error, g), tab_at(u_null_ext, g), u[g], tab_at(training_signal, g), 0,
                             nullptr, device, stream);
  });
}

extern "C" int abrk_sliding_generate_resident(int arm_id, int dtype, const abrk_sliding_params* P,
                                              const abrk_shard_cut* cut, const void* const* q, const void* const* dq,
                                              const void* const* target, const void* const* target_velocity,
                                              const void* const* target_acc, void* const* u, void* const* s_out) {
  if (!q || !dq || !target || !u) return fail(ABRK_EINVAL, "q, dq, target and u are required");
  return for_each_shard(cut, [&](int g, int device, int64_t rows, void* stream) -> int {
    const struct { const char* name; const void* p; } arrs[] = {
        {"q", q[g]}, {"dq", dq[g]}, {"target", target[g]}, {"target_velocity", tab_at(target_velocity, g)},
        {"target_acc", tab_at(target_acc, g)}, {"u", u[g]}, {"s", tab_at(s_out, g)}};
    for (auto& a : arrs)
      if (int rc = check_resident(g, a.name, a.p)) return rc;
    return abrk_sliding_generate_batch(arm_id, dtype, P, rows, q[g], dq[g], target[g], tab_at(target_velocity, g),
                                       tab_at(target_acc, g), u[g], tab_at(s_out, g), device, stream);
  });
}

extern "C" int abrk_joint_generate_resident(int arm_id, int dtype, const abrk_null_ctrl* ctrl, int account_for_gravity,
                                            const abrk_shard_cut* cut, const void* const* q, const void* const* dq,
                                            const void* const* target, const void* const* target_velocity,
                                            void* const* u) {
  if (!q || !dq || !u) return fail(ABRK_EINVAL, "q, dq and u are required");
  return for_each_shard(cut, [&](int g, int device, int64_t rows, void* stream) -> int {
    const struct { const char* name; const void* p; } arrs[] = {
        {"q", q[g]}, {"dq", dq[g]}, {"target", tab_at(target, g)}, {"target_velocity", tab_at(target_velocity, g)},
        {"u", u[g]}};
    for (auto& a : arrs)
      if (int rc = check_resident(g, a.name, a.p)) return rc;
    return abrk_joint_generate_batch(arm_id, dtype, ctrl, account_for_gravity, rows, q[g], dq[g], tab_at(target, g),
                                     tab_at(target_velocity, g), u[g], device, stream);
  });
}

extern "C" int abrk_dynamics_resident(int arm_id, int dtype, const abrk_shard_cut* cut, const void* const* q,
                                      const void* const* dq, int frame, const double* x_off, uint32_t want,
                                      const abrk_dyn_out* out) {
  if (!q || !out) return fail(ABRK_EINVAL, "q and out (one abrk_dyn_out per shard) are required");
  return for_each_shard(cut, [&](int g, int device, int64_t rows, void* stream) -> int {
    if (int rc = check_resident(g, "q", q[g])) return rc;
    if (int rc = check_resident(g, "dq", tab_at(dq, g))) return rc;
    void* const* outs = reinterpret_cast<void* const*>(&out[g]);
    for (int i = 0; i < 10; i++)
      if (want >> i & 1)
        if (int rc = check_resident(g, "an output array", outs[i])) return rc;
    return abrk_dynamics_batch(arm_id, dtype, rows, q[g], tab_at(dq, g), frame, x_off, want, &out[g], device, stream);
  });
}

extern "C" int abrk_shards_sync(const abrk_shard_cut* cut) {
  // every stream is drained before anything is reported: a singular shard does not leave its neighbours running
  bool singular = false;
  int first_rc = 0;
  std::string first_msg;
  const int rc = for_each_shard(cut, [&](int, int device, int64_t, void* stream) -> int {
    const int r = abrk_stream_sync(device, stream);
    if (r == ABRK_ESINGULAR) singular = true;
    else if (r && !first_rc) {
      first_rc = r;
      first_msg = g_err;
    }
    return 0;
  });
  if (rc) return rc;
  if (first_rc) {
    g_err = first_msg;
    return first_rc;
  }
  return singular ? singular_error() : 0;
}

// ------------------------------------------------------------------------------- Sliding
extern "C" int abrk_sliding_generate_batch(int arm_id, int dtype, const abrk_sliding_params* P, int64_t B,
                                           const void* q, const void* dq, const void* target,
                                           const void* target_velocity, const void* target_acc, void* u, void* s_out,
                                           int device, void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (!P) return fail(ABRK_EINVAL, "params is NULL");
  if (P->ref_frame < 0 || P->ref_frame > 2 * n + 1)
    return fail(ABRK_EFRAME, "Invalid transformation name: frame id %d", P->ref_frame);
  if (!q || !dq || !target || !u) return fail(ABRK_EINVAL, "q, dq, target and u are required");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  const int nt = P->cartesian ? 3 : n;
  Stager st{device, (hipStream_t)stream};
  const void* q_ = st.add(q, B * n * s, true, false);
  const void* dq_ = st.add(dq, B * n * s, true, false);
  const void* t_ = st.add(target, B * nt * s, true, false);
  const void* tv_ = st.add(target_velocity, B * nt * s, true, false);
  const void* ta_ = st.add(target_acc, B * nt * s, true, false);
  void* u_ = st.add(u, B * n * s, false, true);
  void* s_ = st.add(s_out, B * n * s, false, true);
  if (int rc = st.reserve()) return rc;
  SlidingArgs sa;
  sa.q = st.fix(q_, q);
  sa.dq = st.fix(dq_, dq);
  sa.target = st.fix(t_, target);
  sa.tv = st.fix(tv_, target_velocity);
  sa.ta = st.fix(ta_, target_acc);
  sa.u = st.fix(u_, u);
  sa.s = st.fix(s_, s_out);
  const SlidingP<double> p64 = make_slidingp<double>(*P, n);
  const SlidingP<float> p32 = make_slidingp<float>(*P, n);
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, a, dtype, [=](const void* rt) {
    SlidingArgs o = sa;
    o.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return ops->sliding(dtype, LaunchArgs{rt, (long)B, hs}, o);
  });
}

// ------------------------------------------------------------------------------- Joint / Damping / RestingConfig
extern "C" int abrk_joint_generate_batch(int arm_id, int dtype, const abrk_null_ctrl* ctrl, int account_for_gravity,
                                         int64_t B, const void* q, const void* dq, const void* target,
                                         const void* target_velocity, void* u, int device, void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (!ctrl) return fail(ABRK_EINVAL, "ctrl is NULL");
  if (ctrl->kind < 0 || ctrl->kind > 2) return fail(ABRK_EINVAL, "unknown controller kind %d", ctrl->kind);
  if (!q || !dq || !u) return fail(ABRK_EINVAL, "q, dq and u are required");
  if (ctrl->kind == 0 && !target) return fail(ABRK_EINVAL, "Joint.generate needs target");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* q_ = st.add(q, B * n * s, true, false);
  const void* dq_ = st.add(dq, B * n * s, true, false);
  const void* t_ = st.add(target, B * n * s, true, false);
  const void* tv_ = st.add(target_velocity, B * n * s, true, false);
  void* u_ = st.add(u, B * n * s, false, true);
  if (int rc = st.reserve()) return rc;
  JointArgs ja;
  ja.q = st.fix(q_, q);
  ja.dq = st.fix(dq_, dq);
  ja.target = st.fix(t_, target);
  ja.tv = st.fix(tv_, target_velocity);
  ja.u = st.fix(u_, u);
  const JointP<double> p64 = make_jointp<double>(*ctrl, account_for_gravity);
  const JointP<float> p32 = make_jointp<float>(*ctrl, account_for_gravity);
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, a, dtype, [=](const void* rt) {
    JointArgs o = ja;
    o.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return ops->joint(dtype, LaunchArgs{rt, (long)B, hs}, o);
  });
}

// ------------------------------------------------------------------------------- AvoidJointLimits / Floating / AvoidObstacles
extern "C" int abrk_avoid_joint_limits_generate_batch(int n_joints, int dtype, const abrk_limits_params* params,
                                                      int64_t B, const void* q, void* u, int accumulate, int device,
                                                      void* stream) {
  if (n_joints < 1 || n_joints > ABRK_MAX_JOINTS) return fail(ABRK_EINVAL, "n_joints=%d outside 1..7", n_joints);
  if (dtype != ABRK_F64 && dtype != ABRK_F32) return fail(ABRK_EINVAL, "unknown dtype %d", dtype);
  if (B < 0) return fail(ABRK_EINVAL, "negative batch size");
  if (!params) return fail(ABRK_EINVAL, "params is NULL");
  if (!q || !u) return fail(ABRK_EINVAL, "q and u are required");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const int n = n_joints;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* q_ = st.add(q, B * n * s, true, false);
  void* u_ = st.add(u, B * n * s, accumulate != 0, true);
  if (int rc = st.reserve()) return rc;
  const LimitsP<double> p64 = make_limitsp<double>(*params);
  const LimitsP<float> p32 = make_limitsp<float>(*params);
  const void* qd = st.fix(q_, q);
  void* ud = st.fix(u_, u);
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, nullptr, dtype, [=](const void*) {
    return launch_limits(n, dtype, LaunchArgs{nullptr, (long)B, hs},
                         dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32, qd, ud, accumulate != 0);
  });
}

extern "C" int abrk_floating_generate_batch(int arm_id, int dtype, int dynamic, int task_space, int64_t B,
                                            const void* q, const void* dq, void* u, int accumulate, int device,
                                            void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (!q || !u) return fail(ABRK_EINVAL, "q and u are required");
  if (dynamic && !dq) return fail(ABRK_EINVAL, "Floating(dynamic=True) needs dq");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* q_ = st.add(q, B * n * s, true, false);
  const void* dq_ = dynamic ? st.add(dq, B * n * s, true, false) : nullptr;
  void* u_ = st.add(u, B * n * s, accumulate != 0, true);
  if (int rc = st.reserve()) return rc;
  FloatingArgs fa;
  fa.dynamic = dynamic != 0;
  fa.task_space = task_space != 0;
  fa.acc = accumulate != 0;
  fa.q = st.fix(q_, q);
  fa.dq = dynamic ? st.fix(dq_, dq) : nullptr;
  fa.u = st.fix(u_, u);
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, a, dtype, [=](const void* rt) { return ops->floating(dtype, LaunchArgs{rt, (long)B, hs}, fa); });
}

extern "C" int abrk_avoid_obstacles_generate_batch(int arm_id, int dtype, const abrk_obstacles_params* params,
                                                   int64_t B, const void* q, void* u, int accumulate, int device,
                                                   void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (!params) return fail(ABRK_EINVAL, "params is NULL");
  if (params->n_obstacles < 0 || params->n_obstacles > ABRK_MAX_OBSTACLES)
    return fail(ABRK_EINVAL, "n_obstacles=%d outside 0..%d", params->n_obstacles, ABRK_MAX_OBSTACLES);
  if (!(params->threshold > 0)) return fail(ABRK_EINVAL, "threshold must be positive");
  if (!q || !u) return fail(ABRK_EINVAL, "q and u are required");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* q_ = st.add(q, B * n * s, true, false);
  void* u_ = st.add(u, B * n * s, accumulate != 0, true);
  if (int rc = st.reserve()) return rc;
  ObstaclesArgs oa;
  const ObsP<double> p64 = make_obsp<double>(*params);
  const ObsP<float> p32 = make_obsp<float>(*params);
  oa.P = nullptr;
  oa.acc = accumulate != 0;
  oa.q = st.fix(q_, q);
  oa.u = st.fix(u_, u);
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, a, dtype, [=](const void* rt) {
    ObstaclesArgs o = oa;
    o.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return ops->obstacles(dtype, LaunchArgs{rt, (long)B, hs}, o);
  });
}

// ------------------------------------------------------------------------------- OSC law on supplied dynamics
extern "C" int abrk_osc_law_batch(int n_joints, int dtype, const abrk_osc_params* P, int64_t B, const void* J,
                                  const void* M, const void* g, const void* Cdq, const void* xyz, const void* R,
                                  const void* q, const void* dq, const void* target, const void* target_velocity,
                                  void* integrated_error, const void* u_null_ext, void* u, void* training_signal,
                                  int device, void* stream) {
  const int n = n_joints;
  if (n < 1 || n > ABRK_MAX_JOINTS) return fail(ABRK_EINVAL, "n_joints=%d outside 1..%d", n, ABRK_MAX_JOINTS);
  if (dtype != ABRK_F64 && dtype != ABRK_F32) return fail(ABRK_EINVAL, "dtype %d is not ABRK_F64/ABRK_F32", dtype);
  if (B < 0) return fail(ABRK_EINVAL, "negative batch %lld", (long long)B);
  if (!P) return fail(ABRK_EINVAL, "params is NULL");
  if (P->n_null < 0 || P->n_null > ABRK_MAX_NULL) return fail(ABRK_EINVAL, "n_null=%d outside 0..%d", P->n_null, ABRK_MAX_NULL);
  if (P->orientation_algorithm != 0 && P->orientation_algorithm != 1)
    return fail(ABRK_EINVAL, "Invalid algorithm number %d for calculating orientation error", P->orientation_algorithm);
  int k = 0, pos = 0, ori = 0;
  for (int r = 0; r < 6; r++) {
    k += P->ctrlr_dof[r] ? 1 : 0;
    (r < 3 ? pos : ori) += P->ctrlr_dof[r] ? 1 : 0;
  }
  if (k == 0) return fail(ABRK_EINVAL, "ctrlr_dof selects no task-space dimension");
  if (!J || !M || !dq || !target || !u) return fail(ABRK_EINVAL, "J, M, dq, target and u are required");
  if (pos && !xyz) return fail(ABRK_EINVAL, "position control needs xyz (robot_config.Tx)");
  if (ori && !R) return fail(ABRK_EINVAL, "orientation control needs R (robot_config.R)");
  if (P->use_g && !g) return fail(ABRK_EINVAL, "use_g needs g (robot_config.g)");
  if (P->use_C && !Cdq) return fail(ABRK_EINVAL, "use_C needs Cdq (robot_config.C(q,dq) @ dq)");
  if (P->ki != 0 && !integrated_error) return fail(ABRK_EINVAL, "ki != 0 needs the integrated_error state array");
  for (int c = 0; c < P->n_null; c++)
    if (P->null_ctrl[c].kind == ABRK_NULL_RESTING && !q) return fail(ABRK_EINVAL, "RestingConfig needs q");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  void* ie = (P->ki != 0) ? integrated_error : nullptr;
  const void* c_in = P->use_C ? Cdq : nullptr;
  const void* g_in = P->use_g ? g : nullptr;
  const void* J_ = st.add(J, B * 6 * n * s, true, false);
  const void* M_ = st.add(M, B * n * n * s, true, false);
  const void* g_ = st.add(g_in, B * n * s, true, false);
  const void* c_ = st.add(c_in, B * n * s, true, false);
  const void* x_ = st.add(xyz, B * 3 * s, true, false);
  const void* R_ = st.add(R, B * 9 * s, true, false);
  const void* q_ = st.add(q, B * n * s, true, false);
  const void* dq_ = st.add(dq, B * n * s, true, false);
  const void* t_ = st.add(target, B * 6 * s, true, false);
  const void* tv_ = st.add(target_velocity, B * 6 * s, true, false);
  void* ie_ = st.add(ie, B * 6 * s, true, true);
  const void* une_ = st.add(u_null_ext, B * n * s, true, false);
  void* u_ = st.add(u, B * n * s, false, true);
  void* ts_ = st.add(training_signal, B * n * s, false, true);
  if (int rc = st.reserve()) return rc;
  LawArgs a;
  a.J = st.fix(J_, J);
  a.M = st.fix(M_, M);
  a.g = st.fix(g_, g_in);
  a.c = st.fix(c_, c_in);
  a.xyz = st.fix(x_, xyz);
  a.R = st.fix(R_, R);
  a.q = st.fix(q_, q);
  a.dq = st.fix(dq_, dq);
  a.target = st.fix(t_, target);
  a.tv = st.fix(tv_, target_velocity);
  a.ierr = st.fix(ie_, ie);
  a.une = st.fix(une_, u_null_ext);
  a.u = st.fix(u_, u);
  a.ts = st.fix(ts_, training_signal);
  OscP<double> p64 = make_oscp<double>(*P, n);
  OscP<float> p32 = make_oscp<float>(*P, n);
  StatusWord* sw = status_word(st.staged, device, stream);
  p64.status = p32.status = sw ? sw->dev : nullptr;
  if (sw && st.staged) *sw->host = 0;  // (this thread's word: nothing of an earlier, failed call is left in it)
  a.P = nullptr;
  const hipStream_t hs = (hipStream_t)stream;
  const int rc = dispatch(st, nullptr, dtype, [=](const void*) {
    LawArgs o = a;
    o.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return launch_osc_law(n, dtype, LaunchArgs{nullptr, (long)B, hs}, o);
  });
  if (rc == 0 && st.staged && sw && sw->take()) return singular_error();
  return rc;
}

// ------------------------------------------------------------------------------- helper methods of OSC
static int check_helper(int dtype, int64_t B) {
  if (recording()) return fail(ABRK_EINVAL, "the OSC helper / transformations entry points cannot be recorded into a plan");
  if (dtype != ABRK_F64 && dtype != ABRK_F32) return fail(ABRK_EINVAL, "dtype %d is not ABRK_F64/ABRK_F32", dtype);
  if (B < 0) return fail(ABRK_EINVAL, "negative batch %lld", (long long)B);
  return 0;
}

extern "C" int abrk_osc_mx_batch(int n_joints, int k, int dtype, int64_t B, const void* M, const void* J,
                                 double threshold, void* Mx, void* M_inv, int device, void* stream) {
  const int n = n_joints;
  if (n < 1 || n > ABRK_MAX_JOINTS) return fail(ABRK_EINVAL, "n_joints=%d outside 1..%d", n, ABRK_MAX_JOINTS);
  if (k < 1 || k > 6) return fail(ABRK_EINVAL, "k=%d task rows outside 1..6", k);
  if (int rc = check_helper(dtype, B)) return rc;
  if (!M || !J || !Mx) return fail(ABRK_EINVAL, "M, J and Mx are required");
  if (!(threshold >= 0)) return fail(ABRK_EINVAL, "threshold must be >= 0");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* M_ = st.add(M, B * n * n * s, true, false);
  const void* J_ = st.add(J, B * k * n * s, true, false);
  void* X_ = st.add(Mx, B * k * k * s, false, true);
  void* I_ = st.add(M_inv, B * n * n * s, false, true);
  if (int rc = st.reserve()) return rc;
  LaunchArgs la{nullptr, (long)B, (hipStream_t)stream};
  HIPCHK(launch_osc_mx(n, dtype, la, k, threshold, st.fix(M_, M), st.fix(J_, J), st.fix(X_, Mx), st.fix(I_, M_inv)));
  return st.finish();
}

extern "C" int abrk_osc_velocity_limiting_batch(int dtype, const abrk_osc_params* P, int64_t B, const void* u_task,
                                                void* out, int device, void* stream) {
  if (int rc = check_helper(dtype, B)) return rc;
  if (!P) return fail(ABRK_EINVAL, "params is NULL");
  if (!P->use_vmax) return fail(ABRK_EINVAL, "velocity limiting needs vmax (OSC(vmax=[xyz, abg]))");
  if (!u_task || !out) return fail(ABRK_EINVAL, "u_task and out are required");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* i_ = st.add(u_task, B * 6 * s, true, false);
  void* o_ = st.add(out, B * 6 * s, false, true);
  if (int rc = st.reserve()) return rc;
  LaunchArgs la{nullptr, (long)B, (hipStream_t)stream};
  const double g[5] = {P->kp, P->ko, P->kv, P->vmax[0], P->vmax[1]};
  HIPCHK(launch_velocity_limiting(dtype, la, g, st.fix(i_, u_task), st.fix(o_, out)));
  return st.finish();
}

extern "C" int abrk_osc_orientation_forces_batch(int algorithm, int dtype, int64_t B, const void* R,
                                                 const void* target_abg, void* u_task_orientation, int device,
                                                 void* stream) {
  if (int rc = check_helper(dtype, B)) return rc;
  if (algorithm != 0 && algorithm != 1)
    return fail(ABRK_EINVAL, "Invalid algorithm number %d for calculating orientation error", algorithm);
  if (!R || !target_abg || !u_task_orientation) return fail(ABRK_EINVAL, "R, target_abg and the output are required");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* R_ = st.add(R, B * 9 * s, true, false);
  const void* a_ = st.add(target_abg, B * 3 * s, true, false);
  void* o_ = st.add(u_task_orientation, B * 3 * s, false, true);
  if (int rc = st.reserve()) return rc;
  LaunchArgs la{nullptr, (long)B, (hipStream_t)stream};
  HIPCHK(launch_orientation_forces(dtype, la, algorithm, st.fix(R_, R), st.fix(a_, target_abg),
                                   st.fix(o_, u_task_orientation)));
  return st.finish();
}

extern "C" int abrk_transformations_batch(int op, int dtype, int64_t B, const void* a, const void* b, void* out,
                                          int device, void* stream) {
  // elements per row of (a, b, out) for each ABRK_TF_* operation
  static const int kIn[8] = {3, 3, 9, 4, 4, 4, 3, 3}, kIn2[8] = {0, 0, 0, 4, 0, 0, 0, 0}, kOut[8] = {4, 4, 4, 4, 4, 4, 3, 9};
  if (op < 0 || op > 7) return fail(ABRK_EINVAL, "unknown transformations op %d", op);
  if (int rc = check_helper(dtype, B)) return rc;
  if (!a || !out || (kIn2[op] && !b)) return fail(ABRK_EINVAL, "input and output arrays are required");
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  const void* b_in = kIn2[op] ? b : nullptr;
  const void* a_ = st.add(a, B * kIn[op] * s, true, false);
  const void* b_ = st.add(b_in, B * kIn2[op] * s, true, false);
  void* o_ = st.add(out, B * kOut[op] * s, false, true);
  if (int rc = st.reserve()) return rc;
  LaunchArgs la{nullptr, (long)B, (hipStream_t)stream};
  HIPCHK(launch_transformations(dtype, la, op, st.fix(a_, a), st.fix(b_, b_in), st.fix(o_, out)));
  return st.finish();
}

// ------------------------------------------------------------------------------- two-link plant / closed loop
namespace {
template <class T>
TwoLinkP<T> make_plant(const abrk_twolink_plant& s) {
  TwoLinkP<T> k;
  k.K1 = T(s.K1);
  k.K2 = T(s.K2);
  k.K3 = T(s.K3);
  k.K4 = T(s.K4);
  k.dt = T(s.dt);
  return k;
}
}  // namespace

extern "C" int abrk_twolink_step_batch(int dtype, const abrk_twolink_plant* plant, int64_t B, void* q, void* dq,
                                       const void* u, int device, void* stream) {
  if (dtype != ABRK_F64 && dtype != ABRK_F32) return fail(ABRK_EINVAL, "dtype %d is not ABRK_F64/ABRK_F32", dtype);
  if (!plant || !q || !dq || !u) return fail(ABRK_EINVAL, "plant, q, dq and u are required");
  if (B < 0) return fail(ABRK_EINVAL, "negative batch %lld", (long long)B);
  if (B == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  Stager st{device, (hipStream_t)stream};
  void* q_ = st.add(q, B * 2 * s, true, true);
  void* dq_ = st.add(dq, B * 2 * s, true, true);
  const void* u_ = st.add(u, B * 2 * s, true, false);
  if (int rc = st.reserve()) return rc;
  const TwoLinkP<double> k64 = make_plant<double>(*plant);
  const TwoLinkP<float> k32 = make_plant<float>(*plant);
  void *qd = st.fix(q_, q), *dqd = st.fix(dq_, dq);
  const void* ud = st.fix(u_, u);
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, nullptr, dtype, [=](const void*) {
    return launch_twolink_step(dtype, LaunchArgs{nullptr, (long)B, hs},
                               dtype == ABRK_F64 ? (const void*)&k64 : (const void*)&k32, qd, dqd, ud);
  });
}

extern "C" int abrk_osc_rollout_twolink_batch(int arm_id, int dtype, const abrk_osc_params* P,
                                              const abrk_twolink_plant* plant, int64_t B, int32_t n_steps,
                                              int32_t every, void* q, void* dq, const void* target,
                                              void* integrated_error, void* q_traj, void* dq_traj, void* u_traj,
                                              int device, void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (n != 2 || !a->ops->rollout)
    return fail(ABRK_EINVAL, "the plant of this entry point is the two-link arm (arms/twojoint/arm_sim.py); arm has %d joints", n);
  if (!P || !plant) return fail(ABRK_EINVAL, "params / plant is NULL");
  if (P->ref_frame < 0 || P->ref_frame > 2 * n + 1)
    return fail(ABRK_EFRAME, "Invalid transformation name: frame id %d", P->ref_frame);
  if (P->n_null < 0 || P->n_null > ABRK_MAX_NULL) return fail(ABRK_EINVAL, "n_null=%d outside 0..%d", P->n_null, ABRK_MAX_NULL);
  int k = 0;
  for (int r = 0; r < 6; r++) k += P->ctrlr_dof[r] ? 1 : 0;
  if (k == 0) return fail(ABRK_EINVAL, "ctrlr_dof selects no task-space dimension");
  if (n_steps < 0 || every < 0) return fail(ABRK_EINVAL, "negative n_steps / every");
  if (!q || !dq || !target) return fail(ABRK_EINVAL, "q, dq and target are required");
  if (P->ki != 0 && !integrated_error) return fail(ABRK_EINVAL, "ki != 0 needs the integrated_error state array");
  if ((q_traj || dq_traj || u_traj) && every <= 0) return fail(ABRK_EINVAL, "trajectory outputs need every > 0");
  if (B == 0 || n_steps == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  const size_t n_chk = every > 0 ? (size_t)(n_steps / every) : 0;
  Stager st{device, (hipStream_t)stream};
  void* ie = (P->ki != 0) ? integrated_error : nullptr;
  void* q_ = st.add(q, B * 2 * s, true, true);
  void* dq_ = st.add(dq, B * 2 * s, true, true);
  const void* t_ = st.add(target, B * 6 * s, true, false);
  void* ie_ = st.add(ie, B * 6 * s, true, true);
  void* qt_ = st.add(q_traj, B * n_chk * 2 * s, false, true);
  void* dqt_ = st.add(dq_traj, B * n_chk * 2 * s, false, true);
  void* ut_ = st.add(u_traj, B * n_chk * 2 * s, false, true);
  if (int rc = st.reserve()) return rc;
  RolloutArgs ra;
  ra.use_C = P->use_C ? 1 : 0;
  ra.fast = osc_fast_rows(*P, n, false);
  ra.n_steps = n_steps;
  ra.every = every;
  ra.q = st.fix(q_, q);
  ra.dq = st.fix(dq_, dq);
  ra.target = st.fix(t_, target);
  ra.ierr = st.fix(ie_, ie);
  ra.qt = st.fix(qt_, q_traj);
  ra.dqt = st.fix(dqt_, dq_traj);
  ra.ut = st.fix(ut_, u_traj);
  OscP<double> p64 = make_oscp<double>(*P, n);
  OscP<float> p32 = make_oscp<float>(*P, n);
  StatusWord* sw = status_word(st.staged, device, stream);
  p64.status = p32.status = sw ? sw->dev : nullptr;
  if (sw && st.staged) *sw->host = 0;  // (this thread's word: nothing of an earlier, failed call is left in it)
  const TwoLinkP<double> k64 = make_plant<double>(*plant);
  const TwoLinkP<float> k32 = make_plant<float>(*plant);
  ra.P = ra.K = nullptr;
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  const int rc = dispatch(st, a, dtype, [=](const void* rt) {
    RolloutArgs o = ra;
    o.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    o.K = dtype == ABRK_F64 ? (const void*)&k64 : (const void*)&k32;
    return ops->rollout(dtype, LaunchArgs{rt, (long)B, hs}, o);
  });
  if (rc == 0 && st.staged && sw && sw->take()) return singular_error();
  return rc;
}

// ------------------------------------------------------------------------------- launch plans
// A plan is the list of kernel launches recorded between abrk_plan_begin and abrk_plan_end (one control tick: one
// law, or several secondary controllers accumulating into the buffer the OSC law then filters).  Launching it only
// enqueues those kernels on the plan's stream - no validation, no pointer classification, no locks, no staging.
namespace {
struct Plan {
  int id = -1, device = 0;
  hipStream_t stream = nullptr;
  std::vector<std::function<hipError_t()>> steps;
  std::vector<std::unique_ptr<std::vector<unsigned char>>> tables;
  std::vector<void*> dev_bufs;
  hipError_t enqueue() const {
    for (const auto& f : steps) {
      hipError_t e = f();
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  int graph_repeat = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  void drop_graph() {
    if (graph_exec || graph) {
      if (hipSetDevice(device) == hipSuccess) {
        t_current_device = device;
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (graph) (void)hipGraphDestroy(graph);
      }
      (void)hipGetLastError();
    }
    graph_exec = nullptr;
    graph = nullptr;
    graph_repeat = 0;
  }
};
// Slot table: abrk_plan_launch reads a slot without taking the lock (hot path of a control loop).  Slots are
// recycled: a plan id is slot | generation << 12, so a stale id of a destroyed plan never matches the slot's new
// tenant; the payload of a destroyed plan is freed.  (Destroying a plan while another thread launches it is a
// caller error, as with any handle.)
constexpr int kSlotBits = 12, kMaxPlans = 1 << kSlotBits, kGenMask = (1 << (31 - kSlotBits)) - 1;
std::mutex g_plan_mu;
Plan* g_plans[kMaxPlans] = {};
int g_plan_gen[kMaxPlans] = {};
std::vector<int> g_plan_free;
int g_plan_hi = 0;  // slots [0, g_plan_hi) have been handed out at least once

Plan* find_plan(int id) {
  if (id < 0) return nullptr;
  Plan* pl = __atomic_load_n(&g_plans[id & (kMaxPlans - 1)], __ATOMIC_ACQUIRE);
  return (pl && pl->id == id) ? pl : nullptr;
}
int register_plan(Plan* pl) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  int slot;
  if (!g_plan_free.empty()) {
    slot = g_plan_free.back();
    g_plan_free.pop_back();
  } else if (g_plan_hi < kMaxPlans) {
    slot = g_plan_hi++;
  } else {
    delete pl;
    return fail(ABRK_ENOMEM, "too many live plans (max %d)", kMaxPlans);
  }
  pl->id = slot | (g_plan_gen[slot] << kSlotBits);
  __atomic_store_n(&g_plans[slot], pl, __ATOMIC_RELEASE);
  return pl->id;
}
void free_dev_bufs(int device, std::vector<void*>& bufs) {
  if (bufs.empty()) return;
  if (hipSetDevice(device) == hipSuccess) {
    t_current_device = device;
    for (void* p : bufs) (void)hipFree(p);
  }
  (void)hipGetLastError();
  bufs.clear();
}
void abort_recording() {
  if (t_rec) free_dev_bufs(t_rec->device, t_rec->dev_bufs);
  delete t_rec;
  t_rec = nullptr;
}
}  // namespace

extern "C" int abrk_plan_begin(int device, void* stream) {
  if (t_rec) return fail(ABRK_EINVAL, "this thread is already recording a plan");
  if (int rc = use_device(device)) return rc;
  t_rec = new Recorder;
  t_rec->device = device;
  t_rec->stream = (hipStream_t)stream;
  return 0;
}

extern "C" int abrk_plan_abort(void) {
  abort_recording();
  return 0;
}

extern "C" int abrk_plan_end(void) {
  if (!t_rec) return fail(ABRK_EINVAL, "abrk_plan_end without abrk_plan_begin");
  if (t_rec->steps.empty()) {
    abort_recording();
    return fail(ABRK_EINVAL, "nothing was recorded (every call had B == 0 or failed)");
  }
  Plan* pl = new Plan;
  pl->device = t_rec->device;
  pl->stream = t_rec->stream;
  pl->steps = std::move(t_rec->steps);
  pl->tables = std::move(t_rec->tables);
  pl->dev_bufs = std::move(t_rec->dev_bufs);
  t_rec->dev_bufs.clear();
  abort_recording();
  return register_plan(pl);
}

// the two original single-law plans: record exactly one call
extern "C" int abrk_osc_plan_create(int arm_id, int dtype, const abrk_osc_params* P, int64_t B, const void* q,
                                    const void* dq, const void* target, const void* target_velocity,
                                    void* integrated_error, const void* u_null_ext, void* u,
                                    void* training_signal, int device, void* stream) {
  if (B <= 0) return fail(ABRK_EINVAL, "a plan needs a positive batch");
  if (int rc = abrk_plan_begin(device, stream)) return rc;
  if (int rc = abrk_osc_generate_batch(arm_id, dtype, P, B, q, dq, target, target_velocity, integrated_error,
                                       u_null_ext, u, training_signal, device, stream)) {
    abort_recording();
    return rc;
  }
  return abrk_plan_end();
}

extern "C" int abrk_sliding_plan_create(int arm_id, int dtype, const abrk_sliding_params* P, int64_t B, const void* q,
                                        const void* dq, const void* target, const void* target_velocity,
                                        const void* target_acc, void* u, void* s_out, int device, void* stream) {
  if (B <= 0) return fail(ABRK_EINVAL, "a plan needs a positive batch");
  if (int rc = abrk_plan_begin(device, stream)) return rc;
  if (int rc = abrk_sliding_generate_batch(arm_id, dtype, P, B, q, dq, target, target_velocity, target_acc, u, s_out,
                                           device, stream)) {
    abort_recording();
    return rc;
  }
  return abrk_plan_end();
}

extern "C" int abrk_plan_launch(int plan) {
  Plan* pl = find_plan(plan);
  if (!pl) return fail(ABRK_EINVAL, "unknown plan %d", plan);
  if (t_current_device != pl->device) {
    HIPCHK(hipSetDevice(pl->device));
    t_current_device = pl->device;
  }
  HIPCHK(pl->enqueue());
  return 0;
}

extern "C" int abrk_plan_launch_repeat(int plan, int repeat) {
  Plan* pl = find_plan(plan);
  if (!pl) return fail(ABRK_EINVAL, "unknown plan %d", plan);
  if (repeat < 1) return fail(ABRK_EINVAL, "repeat must be >= 1");
  if (t_current_device != pl->device) {
    HIPCHK(hipSetDevice(pl->device));
    t_current_device = pl->device;
  }
  for (int i = 0; i < repeat; i++) HIPCHK(pl->enqueue());
  return 0;
}

extern "C" int abrk_plan_launch_graph(int plan, int repeat) {
  Plan* pl = find_plan(plan);
  if (!pl) return fail(ABRK_EINVAL, "unknown plan %d", plan);
  if (repeat < 1) return fail(ABRK_EINVAL, "repeat must be >= 1");
  if (t_current_device != pl->device) {
    HIPCHK(hipSetDevice(pl->device));
    t_current_device = pl->device;
  }
  if (!pl->stream) return fail(ABRK_EINVAL, "graph launches need a plan created on an explicit stream");
  if (pl->graph_repeat != repeat) {
    pl->drop_graph();
    HIPCHK(hipStreamBeginCapture(pl->stream, hipStreamCaptureModeThreadLocal));
    hipError_t le = hipSuccess;
    for (int i = 0; i < repeat && le == hipSuccess; i++) le = pl->enqueue();
    hipError_t ce = hipStreamEndCapture(pl->stream, &pl->graph);
    if (le != hipSuccess || ce != hipSuccess)
      return fail(ABRK_ENODEV, "graph capture failed: %s", hipGetErrorString(le != hipSuccess ? le : ce));
    HIPCHK(hipGraphInstantiate(&pl->graph_exec, pl->graph, nullptr, nullptr, 0));
    // (hipGraphUpload here was measured in round 3: no change to a 20-step replay - 4.83 us per step of wall clock with
    //  and without, the graph's first launch is an untimed warm-up anyway - and one more call for rocprofv3 to trip over)
    pl->graph_repeat = repeat;
  }
  HIPCHK(hipGraphLaunch(pl->graph_exec, pl->stream));
  return 0;
}

// One tick (or `repeat` of them) of SEVERAL plans from one call - the plans of a sharded control loop, one per shard /
// device: mode 0 = `repeat` plain launches of each plan, 1 = one hipGraph of `repeat` ticks per plan (captured on first use,
// abrk_plan_launch_graph).  Every device has its work before the call returns; nothing is waited for.
extern "C" int abrk_plans_launch(const int* plans, int n_plans, int repeat, int mode) {
  if (!plans || n_plans < 1) return fail(ABRK_EINVAL, "plans / n_plans");
  if (mode != 0 && mode != 1) return fail(ABRK_EINVAL, "mode must be 0 (plain launches) or 1 (graph replay)");
  for (int i = 0; i < n_plans; i++)
    if (int rc = mode ? abrk_plan_launch_graph(plans[i], repeat) : abrk_plan_launch_repeat(plans[i], repeat)) return rc;
  return 0;
}

extern "C" int abrk_plan_destroy(int plan) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  Plan* pl = find_plan(plan);
  if (!pl) return fail(ABRK_EINVAL, "unknown plan %d", plan);
  const int slot = plan & (kMaxPlans - 1);
  __atomic_store_n(&g_plans[slot], (Plan*)nullptr, __ATOMIC_RELEASE);
  g_plan_gen[slot] = (g_plan_gen[slot] + 1) & kGenMask;
  g_plan_free.push_back(slot);
  pl->drop_graph();
  free_dev_bufs(pl->device, pl->dev_bufs);
  delete pl;
  return 0;
}

extern "C" int abrk_plan_count(void) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  return g_plan_hi - (int)g_plan_free.size();
}

// ------------------------------------------------------------------------------- inverse kinematics
extern "C" int abrk_ik_generate_path_batch(int arm_id, int dtype, const abrk_ik_params* P, int64_t B,
                                           const void* position, const void* target, void* position_path,
                                           void* velocity_path, int device, void* stream) {
  ArmEntry* a;
  if (int rc = check_common(arm_id, dtype, B, &a)) return rc;
  const int n = a->desc.n_joints;
  if (!P) return fail(ABRK_EINVAL, "params is NULL");
  if (P->method < 1 || P->method > 3) return fail(ABRK_EINVAL, "method %d outside 1..3", P->method);
  if (P->n_timesteps < 0) return fail(ABRK_EINVAL, "negative n_timesteps");
  if (!position || !target || !position_path || !velocity_path)
    return fail(ABRK_EINVAL, "position, target, position_path and velocity_path are required");
  if (B == 0 || P->n_timesteps == 0) return 0;
  if (int rc = use_device(device)) return rc;
  const size_t s = esz(dtype);
  const size_t T = (size_t)P->n_timesteps;
  Stager st{device, (hipStream_t)stream};
  const void* q_ = st.add(position, B * n * s, true, false);
  const void* t_ = st.add(target, B * 6 * s, true, false);
  void* pp_ = st.add(position_path, B * T * n * s, false, true);
  void* vp_ = st.add(velocity_path, B * T * n * s, false, true);
  if (int rc = st.reserve()) return rc;
  IkArgs ia;
  ia.q = st.fix(q_, position);
  ia.target = st.fix(t_, target);
  ia.pp = st.fix(pp_, position_path);
  ia.vp = st.fix(vp_, velocity_path);
  IkP<double> p64{P->max_dx * P->dt, P->max_dr * P->dt, P->max_dq * P->dt, P->n_timesteps, P->method};
  IkP<float> p32{(float)(P->max_dx * P->dt), (float)(P->max_dr * P->dt), (float)(P->max_dq * P->dt), P->n_timesteps,
                 P->method};
  ia.P = nullptr;
  const ArmOps* ops = a->ops;
  const hipStream_t hs = (hipStream_t)stream;
  return dispatch(st, a, dtype, [=](const void* rt) {
    IkArgs o = ia;
    o.P = dtype == ABRK_F64 ? (const void*)&p64 : (const void*)&p32;
    return ops->ik(dtype, LaunchArgs{rt, (long)B, hs}, o);
  });
}
