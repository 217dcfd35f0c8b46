// Per-wavefront timeline of the six-row law's config-sized step (UR5, all six task rows, 4096 rows): the product's first
// pass (osc_kernel<.., 6, false, 0, 1, true>, -DABRK_TIMELINE: stamps leave through the unused `uneg` pointer) followed by
// the product's finish kernel, as the host layer launches them, 100 step pairs per hipGraph.  Where do the first pass's
// ~7.8 us go?  (The finish kernel carries no stamps: its span is what is left of the period.)
//   build: tools/gpu_r5_timeline.sh build6    run: tools/microbench/osc6_step_timeline.bin out.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "abrk_kernels.h"
#include "abrk_params.h"

using namespace abrk;
#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

static void stat(FILE* f, const char* name, std::vector<double> v, const char* tail) {
  std::sort(v.begin(), v.end());
  double s = 0;
  for (double x : v) s += x;
  auto q = [&](double p) { return v.empty() ? 0.0 : v[(size_t)(p * (v.size() - 1))]; };
  fprintf(f, "\"%s\": {\"n\": %zu, \"min\": %.3f, \"p10\": %.3f, \"median\": %.3f, \"p90\": %.3f, \"max\": %.3f}%s", name, v.size(),
          q(0), q(0.1), q(0.5), q(0.9), q(1.0), tail);
}

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
  const long nchunk = (B + kBlock - 1) / kBlock;
  double *q_, *dq_, *t_, *u_, *rec;
  unsigned long long *masks, *tl;
  CK(hipMalloc(&q_, B * N * 8));
  CK(hipMalloc(&dq_, B * N * 8));
  CK(hipMalloc(&t_, B * 6 * 8));
  CK(hipMalloc(&u_, B * N * 8));
  CK(hipMalloc(&rec, (size_t)nchunk * kBlock * rec_len(N) * 8));
  CK(hipMalloc(&masks, nchunk * 8));
  CK(hipMalloc(&tl, (size_t)K * nchunk * 16 * 8));
  CK(hipMemset(tl, 0, (size_t)K * nchunk * 16 * 8));
  CK(hipMemcpy(q_, q.data(), B * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dq_, dq.data(), B * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(t_, t.data(), B * 6 * 8, hipMemcpyHostToDevice));
  abrk_osc_params hp;
  memset(&hp, 0, sizeof hp);
  hp.kp = 200;
  hp.ko = 150;
  hp.kv = 25;
  hp.use_g = 1;
  for (int r = 0; r < 6; r++) hp.ctrlr_dof[r] = 1;
  hp.ref_frame = 2 * N + 1;
  const OscP<T> P = make_oscp<T>(hp, N);
  const int slots = finish_slots(B), rounds = finish_rounds(B);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto step = [&](int k) {
    hipLaunchKernelGGL((osc_kernel<A, T, 6, false, 0, 1, true>), dim3((unsigned)nchunk), dim3(kBlock), 0, st, A{}, P, B,
                       (const T*)q_, (const T*)dq_, (const T*)t_, (const T*)nullptr, (T*)nullptr,
                       reinterpret_cast<const T*>(tl + (size_t)k * nchunk * 16), u_, (T*)nullptr, 1,
                       reinterpret_cast<int*>(masks), rec);
    hipLaunchKernelGGL((osc6_finish_kernel<N, T>), dim3((unsigned)nchunk, (unsigned)slots), dim3(kBlock), 0, st,
                       (const unsigned long long*)masks, (const T*)rec, 0, rounds, B, u_, (T*)nullptr);
  };
  step(0);
  CK(hipStreamSynchronize(st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int k = 0; k < K; k++) step(k);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 6; rep++) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<unsigned long long> h((size_t)K * nchunk * 16);
  CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
  static const char* names[16] = {"entry", "table_barrier", "inputs_landed", "kinematics_dynamics_jacobian_done", "law_done",
                                  "stores_issued", "stores_complete", "", "law:before_chol_M", "law:before_Y", "law:Am_done",
                                  "law:certificates_done", "law:record_written", "law:before_f", "law:JTf_done", ""};
  const int order[13] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 4, 6};
  const double tick = 0.01;
  std::vector<double> rel[16], span, period, clk;
  for (int k = K / 4; k < 3 * K / 4; k++) {
    const unsigned long long* n = h.data() + (size_t)k * nchunk * 16;
    unsigned long long first = ~0ull, lastend = 0;
    for (long w = 0; w < nchunk; w++) {
      first = std::min(first, n[w * 16]);
      lastend = std::max(lastend, n[w * 16 + 6]);
    }
    for (long w = 0; w < nchunk; w++) {
      for (int s = 0; s < 16; s++)
        if (n[w * 16 + s] >= first && s != 7) rel[s].push_back((double)(n[w * 16 + s] - first) * tick);
      const double life = (double)(n[w * 16 + 6] - n[w * 16]) * tick;
      if (life > 0) clk.push_back((double)n[w * 16 + 7] / (life * 1e3));
    }
    span.push_back((double)(lastend - first) * tick);
    if (k + 1 < 3 * K / 4) {
      const unsigned long long* nn = h.data() + (size_t)(k + 1) * nchunk * 16;
      unsigned long long nf = ~0ull;
      for (long w = 0; w < nchunk; w++) nf = std::min(nf, nn[w * 16]);
      period.push_back((double)(nf - first) * tick);
    }
  }
  fprintf(out, "  {\"rows\": %ld, \"wavefronts\": %ld, \"finish_slots\": %d, \"hip_event_us_per_step\": %.3f,\n   \"first_pass_stamps_us_since_first_entry\": {",
          B, nchunk, slots, ms * 1e3 / K);
  for (int i = 0; i < 13; i++) stat(out, names[order[i]], rel[order[i]], i < 12 ? ",\n      " : "},\n   ");
  stat(out, "first_pass_span_us", span, ",\n   ");
  stat(out, "step_period_us_first_pass_entry_to_next_first_pass_entry", period, ",\n   ");
  stat(out, "shader_clock_ghz", clk, "}");
  fprintf(out, "%s\n", last ? "" : ",");
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
  FILE* out = fopen(argc > 1 ? argv[1] : "/dev/stdout", "w");
  if (!out) return 1;
  fprintf(out, "{\"what\": \"six-row law, UR5, fp64, NOTS first pass + finish kernel per step, 100 steps per hipGraph, middle 50 steps of the last of 6 replays; stamps: s_memrealtime (10 ns)\", \"legs\": [\n");
  run(4096, 100, out, false);
  run(16384, 100, out, true);
  fprintf(out, "]}\n");
  fclose(out);
  return 0;
}
