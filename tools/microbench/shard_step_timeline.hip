// Where does a shard-sized step of BASELINE config 4 (UR5 OSC + g + C, 131 072 rows: the 8-way shard of 2^20) spend its
// ~10 us when 2048 wavefronts x ~1650 issue slots x 4 cycles / 2.4 GHz is 5.5 us?  (VERDICT r4 "Next" #3.)
//
// The product's own x,y,z kernel - the same headers, the same flags - compiled with -DABRK_TIMELINE: every wavefront
// stamps the constant-rate counter (s_memrealtime, 100 MHz) at seven points (abrk_device.h ABRK_STAMP) and writes its
// eight values through the kernel's unused worklist pointer.  The kernel is launched exactly as bench.py's step is: K
// kernel nodes in one hipGraph, replayed; node k writes its stamps to its own slice, so the gaps BETWEEN the kernels of
// a replay are on the record too.  A second instantiation (-DABRK_TIMELINE_LIGHT: only entry and exit are meaningful, no
// forced waits inside) checks how much the stamps themselves cost.
//
//   build (cross-compiles in the build container):  tools/gpu_r5_timeline.sh build
//   run on the GPU box:                             tools/microbench/shard_step_timeline.bin <rows> <use_C> <K> <out.json>
//
// Output: JSON with, per launch size, the per-stamp statistics over all wavefronts of the LAST replay's middle nodes
// (relative to the node's first wavefront entering), the node-to-node gaps, the HIP-event time per node of the same
// replay, and the shader clock each wavefront saw (cycles / realtime).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "abrk_kernels.h"
#include "abrk_params.h"

using namespace abrk;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct Stat {
  double mn, p10, med, p90, mx, mean;
};
static Stat stat(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  double s = 0;
  for (double x : v) s += x;
  auto q = [&](double f) { return v[(size_t)(f * (v.size() - 1))]; };
  return {v.front(), q(0.1), q(0.5), q(0.9), v.back(), s / v.size()};
}
static void put(FILE* f, const char* name, const Stat& s, const char* tail) {
  fprintf(f, "\"%s\": {\"min\": %.3f, \"p10\": %.3f, \"median\": %.3f, \"p90\": %.3f, \"max\": %.3f, \"mean\": %.3f}%s", name,
          s.mn, s.p10, s.med, s.p90, s.mx, s.mean, tail);
}

template <bool USE_C>
static void run(long B, int K, FILE* out, bool last) {
  using A = StaticArm<Tab_ur5>;
  using T = double;
  constexpr int N = 6;
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> uq(0, 6.283185307179586), ud(0, 5), ut(-1, 1);
  std::vector<double> q(B * N), dq(B * N), t(B * 6);
  for (auto& x : q) x = uq(rng);
  for (auto& x : dq) x = ud(rng);
  for (auto& x : t) x = ut(rng);
  double *dq_, *q_, *t_, *u_;
  CK(hipMalloc(&q_, B * N * 8));
  CK(hipMalloc(&dq_, B * N * 8));
  CK(hipMalloc(&t_, B * 6 * 8));
  CK(hipMalloc(&u_, B * N * 8));
  CK(hipMemcpy(q_, q.data(), B * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dq_, dq.data(), B * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(t_, t.data(), B * 6 * 8, hipMemcpyHostToDevice));
  const long blocks = (B + kBlock - 1) / kBlock;
  unsigned long long* tl;
  CK(hipMalloc(&tl, (size_t)K * blocks * 8 * sizeof(unsigned long long)));
  CK(hipMemset(tl, 0, (size_t)K * blocks * 8 * sizeof(unsigned long long)));

  abrk_osc_params hp;
  memset(&hp, 0, sizeof hp);
  hp.kp = 200;
  hp.ko = 200;
  hp.kv = std::sqrt(400.0);
  hp.use_g = 1;
  hp.use_C = USE_C ? 1 : 0;
  hp.ctrlr_dof[0] = hp.ctrlr_dof[1] = hp.ctrlr_dof[2] = 1;
  hp.ref_frame = 2 * N + 1;
  const OscP<T> P = make_oscp<T>(hp, N);

  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto launch = [&](int k) {
    hipLaunchKernelGGL((osc_kernel<A, T, 3, USE_C, 0>), dim3((unsigned)blocks), dim3(kBlock), 0, st, A{}, P, B,
                       (const T*)q_, (const T*)dq_, (const T*)t_, (const T*)nullptr, (T*)nullptr, (const T*)nullptr, u_,
                       (T*)nullptr, 0, reinterpret_cast<int*>(tl + (size_t)k * blocks * 8), (T*)nullptr);
  };
  launch(0);  // loads the code object
  CK(hipStreamSynchronize(st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int k = 0; k < K; k++) launch(k);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 6; rep++) {  // the last replay is the one that is read
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<unsigned long long> h((size_t)K * blocks * 8);
  CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
  const double tick_us = 0.01;  // s_memrealtime: 100 MHz
  // middle nodes of the graph: away from the launch's lead-in and tail
  const int k0 = K / 4, k1 = 3 * K / 4;
  static const char* names[7] = {"entry", "table_barrier", "inputs_landed", "dynamics_done", "law_done", "stores_issued",
                                 "stores_complete"};
  std::vector<double> rel[7], clk, span, gap, busy;
  std::vector<double> per_wave_life;
  for (int k = k0; k < k1; k++) {
    const unsigned long long* n = h.data() + (size_t)k * blocks * 8;
    unsigned long long first = ~0ull, lastend = 0;
    for (long w = 0; w < blocks; w++) {
      first = std::min(first, n[w * 8 + 0]);
      lastend = std::max(lastend, n[w * 8 + 6]);
    }
    for (long w = 0; w < blocks; w++) {
      for (int s = 0; s < 7; s++) rel[s].push_back((double)(n[w * 8 + s] - first) * tick_us);
      const double life = (double)(n[w * 8 + 6] - n[w * 8 + 0]) * tick_us;
      per_wave_life.push_back(life);
      if (life > 0) clk.push_back((double)n[w * 8 + 7] / (life * 1e3));  // cycles per ns = GHz
    }
    span.push_back((double)(lastend - first) * tick_us);
    if (k + 1 < k1) {
      const unsigned long long* nn = h.data() + (size_t)(k + 1) * blocks * 8;
      unsigned long long nfirst = ~0ull;
      for (long w = 0; w < blocks; w++) nfirst = std::min(nfirst, nn[w * 8 + 0]);
      gap.push_back(((double)nfirst - (double)lastend) * tick_us);
      busy.push_back((double)(nfirst - first) * tick_us);
    }
  }
  fprintf(out, "  {\"rows\": %ld, \"use_C\": %d, \"wavefronts\": %ld, \"graph_nodes\": %d, \"hip_event_us_per_node\": %.3f,\n", B,
          USE_C ? 1 : 0, blocks, K, ms * 1e3 / K);
  fprintf(out, "   \"what\": \"us since the node's first wavefront entered, over all wavefronts of graph nodes %d..%d of the last of 6 "
               "replays (s_memrealtime, 10 ns ticks)\",\n", k0, k1 - 1);
  fprintf(out, "   \"stamps\": {");
  for (int s = 0; s < 7; s++) put(out, names[s], stat(rel[s]), s < 6 ? ",\n              " : "},\n");
  fprintf(out, "   ");
  put(out, "wavefront_lifetime_us", stat(per_wave_life), ",\n   ");
  put(out, "node_span_first_entry_to_last_exit_us", stat(span), ",\n   ");
  put(out, "gap_last_exit_to_next_first_entry_us", stat(gap), ",\n   ");
  put(out, "first_entry_to_next_first_entry_us", stat(busy), ",\n   ");
  put(out, "shader_clock_ghz_seen_by_wavefronts", stat(clk), "}");
  fprintf(out, "%s\n", last ? "" : ",");
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipFree(q_));
  CK(hipFree(dq_));
  CK(hipFree(t_));
  CK(hipFree(u_));
  CK(hipFree(tl));
  CK(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/dev/stdout";
  const int K = argc > 2 ? atoi(argv[2]) : 100;
  FILE* out = fopen(path, "w");
  if (!out) return 1;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
#if defined(ABRK_TIMELINE_LIGHT)
  const char* variant = "light (no forced waits inside the row program: only entry / exit stamps are meaningful)";
#else
  const char* variant = "full (s_waitcnt 0 before the inputs_landed and stores_complete stamps)";
#endif
  fprintf(out, "{\"device\": \"%s\", \"variant\": \"%s\", \"legs\": [\n", pr.name, variant);
  run<true>(131072, K, out, false);   // the 8-way shard of BASELINE config 4
  run<true>(4096, K, out, false);     // config-sized
  run<true>(1 << 20, std::min(K, 40), out, false);
  run<false>(131072, K, out, false);  // config 2's kernel at the same sizes
  run<false>(4096, K, out, true);
  fprintf(out, "]}\n");
  fclose(out);
  return 0;
}
