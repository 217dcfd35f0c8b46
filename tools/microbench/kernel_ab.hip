// Same-box A/B of ONE law of the product's kernels against a header directory (seconds to build instead of the library's
// minutes): the OSC launch logic itself - abrk_kernels.h Launch::osc_launch<KM, USE_C, FEAT>, i.e. the same passes, grids
// and worklist handling libabrk enqueues - compiled for one arm and one law, timed at several batch sizes.
//
//   build:  hipcc <library flags> -I<hdr-dir> -Iinclude -DAB_KM=6 -DAB_USE_C=0 -DAB_FEAT=0 [-DAB_ARM=Tab_jaco2] [-DAB_DOF5]
//           [-DAB_TS] tools/microbench/kernel_ab.hip <hdr-dir's or the library's build/abrk_law.o> -o kernel_ab_X.bin
//           (tools/gpu_r6_ab.sh does it for two header directories)
//   run:    kernel_ab_X.bin <out.json> rows[,rows...]
// Per size: K-node hipGraph replays below 1 M rows (the config-sized, launch-bound regime: us per step), back-to-back
// launches for 2 s from 1 M rows on (mean of the last second: the power-limited sustained rate), and an FNV hash of u so
// that two variants can be checked for bit-equality.  Six-row laws: hand-over form (first pass + finish kernel) from 64
// to 65 536 rows, recompute form beyond - as the host layer picks them (abrk_host.cpp worklist_for).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "abrk_kernels.h"
#include "abrk_params.h"

using namespace abrk;
#ifndef AB_ARM
#define AB_ARM Tab_ur5
#endif
#ifndef AB_KM
#define AB_KM 3
#endif
#ifndef AB_USE_C
#define AB_USE_C 0
#endif
#ifndef AB_FEAT
#define AB_FEAT 0
#endif

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  using A = StaticArm<AB_ARM>;
  using T = double;
  constexpr int N = A::N;
  if (argc < 3) return 1;
  FILE* out = fopen(argv[1], "w");
  std::vector<long> sizes;
  for (char* p = strtok(argv[2], ","); p; p = strtok(nullptr, ",")) sizes.push_back(atol(p));
  abrk_osc_params hp;
  memset(&hp, 0, sizeof hp);
  hp.kp = 200;
  hp.ko = AB_KM == 6 ? 150 : 200;
  hp.kv = AB_KM == 6 ? 25 : std::sqrt(400.0);
  hp.use_g = 1;
  hp.use_C = AB_USE_C;
  for (int r = 0; r < (AB_KM == 6 ? 6 : 3); r++) hp.ctrlr_dof[r] = 1;
#ifdef AB_DOF5
  hp.ctrlr_dof[5] = 0;
#endif
#if AB_FEAT == 1
  hp.n_null = 1;
  hp.null_ctrl[0].kind = ABRK_NULL_DAMPING;
  hp.null_ctrl[0].kv = 10;
#endif
  hp.ref_frame = 2 * N + 1;
  // run-time variations of the law (instruction-count differences under rocprofv3 --pmc: tools/gpu_r6_pmc.sh)
  if (const char* e = getenv("AB_DOF"))
    for (int r = 0; r < 6 && e[r]; r++) hp.ctrlr_dof[r] = e[r] == '1';
  if (const char* e = getenv("AB_ALG")) hp.orientation_algorithm = atoi(e);
  if (getenv("AB_VMAX")) {
    hp.use_vmax = 1;
    hp.vmax[0] = 0.5;
    hp.vmax[1] = 1.0;
  }
  if (getenv("AB_NOG")) hp.use_g = 0;
  const bool quick = getenv("AB_QUICK") != nullptr;  // one warm launch + 3 timed ones per size (counter passes)
  const OscP<T> P = make_oscp<T>(hp, N);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  fprintf(out, "{\"arm\": \"%s\", \"km\": %d, \"use_C\": %d, \"feat\": %d, \"legs\": [", AB_ARM::kName, AB_KM, AB_USE_C, AB_FEAT);
  for (size_t si = 0; si < sizes.size(); si++) {
    const long B = sizes[si];
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> uq(0, 6.283185307179586), ud(0, 5), ut(-1, 1);
    std::vector<double> q(B * N), dq(B * N), t(B * 6);
    for (auto& x : q) x = uq(rng);
    for (auto& x : dq) x = ud(rng);
    for (auto& x : t) x = ut(rng);
    double *q_, *dq_, *t_, *u_, *ts_ = nullptr;
    CK(hipMalloc(&q_, B * N * 8));
    CK(hipMalloc(&dq_, B * N * 8));
    CK(hipMalloc(&t_, B * 6 * 8));
    CK(hipMalloc(&u_, B * N * 8));
#ifdef AB_TS
    CK(hipMalloc(&ts_, B * N * 8));
#endif
    CK(hipMemcpy(q_, q.data(), B * N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dq_, dq.data(), B * N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(t_, t.data(), B * 6 * 8, hipMemcpyHostToDevice));
    CK(hipMemset(u_, 0, B * N * 8));
    OscArgs oa;
    oa.P = &P;
    oa.fast = AB_KM == 6 ? 0 : AB_KM;
    oa.use_C = AB_USE_C;
    oa.q = q_;
    oa.dq = dq_;
    oa.target = t_;
    oa.tv = nullptr;
    oa.une = nullptr;
    oa.ierr = nullptr;
    oa.u = u_;
    oa.ts = ts_;
#ifdef AB_DENSE
    const bool handover = AB_KM == 6 && B >= 64;  // beyond 65 536 rows: hand-over records + the dense finish kernel
#else
    const bool handover = AB_KM == 6 && B >= 64 && B <= 65536;
#endif
    FinishArgs fa{};
    if (AB_KM == 6 && B >= 64) {
      CK(hipMalloc(&oa.wl, wl_ints(B) * sizeof(int) + 4096));
      CK(hipMemset(oa.wl, 0, wl_ints(B) * sizeof(int)));
      if (handover) {
        CK(hipMalloc(&oa.rec, (size_t)B * rec_len(N) * sizeof(T)));
        const long nchunk = (B + kBlock - 1) / kBlock;
        fa = FinishArgs{oa.wl, oa.rec, AB_FEAT >= 1 ? 1 : 0, finish_slots(B), finish_rounds(B), oa.u, oa.ts,
                        nchunk > 128 && nchunk <= 256 ? 16 : 0};
#ifdef AB_DENSE
        fa.dense = B > 65536;
#endif
      }
    }
    const LaunchArgs la{nullptr, B, st};
    auto step = [&]() {
      CK((Launch<A, T>::template osc_launch<AB_KM, AB_USE_C != 0, AB_FEAT>(la, oa)));
      if (handover) CK(launch_osc6_finish(N, 0, la, fa));
    };
    step();
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    std::vector<double> uh(B * N);
    CK(hipMemcpy(uh.data(), u_, B * N * 8, hipMemcpyDeviceToHost));
    unsigned long long h = 1469598103934665603ull;
    for (double v : uh) {
      unsigned long long b;
      memcpy(&b, &v, 8);
      h = (h ^ b) * 1099511628211ull;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double us = 0, us_first = 0;
    const char* how;
    if (quick) {
      how = "quick";
      float ms;
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < 3; k++) step();
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      us = us_first = ms * 1e3 / 3;
    } else if (B < (1 << 20)) {
      how = "graph";
      const int K = 100;
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < K; k++) step();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      std::vector<double> reps;
      for (int rep = 0; rep < 12; rep++) {
        float ms;
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) reps.push_back(ms * 1e3 / K);
      }
      std::sort(reps.begin(), reps.end());
      us = reps[reps.size() / 2];
      us_first = reps.front();
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    } else {
      how = "sustained";
      for (int w = 0; w < 12; w++) step();
      CK(hipStreamSynchronize(st));
      std::vector<std::pair<double, double>> chunks;  // (end time s, us per launch)
      const auto t0 = std::chrono::steady_clock::now();
      double elapsed = 0;
      const int chunk = 16;
      while (elapsed < 2.0) {
        float ms;
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < chunk; k++) step();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        chunks.push_back({elapsed, ms * 1e3 / chunk});
      }
      double s = 0;
      int n = 0;
      for (auto& c : chunks)
        if (c.first > elapsed - 1.0) {
          s += c.second;
          n++;
        }
      us = s / n;
      us_first = chunks.front().second;
    }
    fprintf(out, "%s{\"rows\": %ld, \"how\": \"%s\", \"us_per_step\": %.3f, \"us_first_or_min\": %.3f, \"u_hash\": \"%016llx\"}",
            si ? ", " : "", B, how, us, us_first, h);
    fprintf(stderr, "%s km%d C%d F%d rows %ld: %.3f us (%s), hash %016llx\n", AB_ARM::kName, AB_KM, AB_USE_C, AB_FEAT, B, us, how, h);
    CK(hipFree(q_));
    CK(hipFree(dq_));
    CK(hipFree(t_));
    CK(hipFree(u_));
    if (ts_) CK(hipFree(ts_));
    if (oa.wl) CK(hipFree(oa.wl));
    if (oa.rec) CK(hipFree(oa.rec));
  }
  fprintf(out, "]}\n");
  fclose(out);
  return 0;
}
