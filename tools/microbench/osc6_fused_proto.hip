// Prototype: the six-row law's small-batch step as ONE launch (profiles/round4/NOTES.md "what is left" item 2).
//
// Today a 4096-row step of the six-row law is two kernels: the first pass (osc_kernel<.., PASS = 1>: a deferring row
// leaves a hand-over record, one ballot mask per 64-row chunk) and the finish kernel (osc6_finish_kernel: wavefront
// (chunk, slot) decomposes one record's Mx_inv).  15.6 us = 7.8 + 8.1, of which the finish kernel's own arithmetic is
// 3.6 us: the rest of it is a launch gap (1.0 us between dependent kernel nodes of a graph replay), its ramp, and a
// memory round trip that only starts when the kernel does.
// Here both roles live in one grid: blocks [0, nchunk) are the first pass, blocks nchunk + s * nchunk + j are the
// finish wavefronts (chunk j, slot s).  A finish wavefront is resident from the start of the launch and WAITS for its
// chunk: the first-pass wavefront publishes `ready[j] = slots` (release, agent scope) after its mask and records; every
// finish wavefront of the chunk takes one count back when it is done, so the word is zero again when the launch ends
// (no clearing between launches, nothing epoch-like in the kernel arguments: a recorded plan / hipGraph replays the
// same arguments).  Forward progress: workgroups are dispatched in index order (per XCD, round-robin over XCDs), so
// every first-pass workgroup is resident before any finish workgroup that could wait for it; the wait is bounded
// (kSpinLimit polls, then an error flag) - a broken assumption cannot hang the device.
// This file measures the fused launch against the two-kernel step on the same inputs (bit-equal outputs required).
//   build:  tools/gpu_r5_fused.sh build        run (GPU box):  tools/microbench/osc6_fused_proto.bin out.json
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

constexpr int kSpinLimit = 1 << 20;  // polls of ~1 us each at the longest: a second, then the error flag

// the finish role for (chunk j, slot s0): the body of osc6_finish_kernel (abrk_kernels.h), mask and record asked for
// together once the chunk is published
template <int N, class T>
__device__ __forceinline__ void finish_role(long j, int s0, int slots, const unsigned long long* __restrict__ masks,
                                            const T* __restrict__ recs, int nulls, int coop_rounds, long B,
                                            T* __restrict__ ug, T* __restrict__ tsg) {
  const int lane = (int)threadIdx.x;
  const int c = lane < N + 2 ? lane : N + 1;
  const int jc = lane < N ? lane : 0;
  const T* rec = recs + (j * kBlock + s0) * rec_len(N);
  long jv = j;
  pin_loaded(jv);
  unsigned long long mask = masks[jv];
  T S[21], G[1][6], ridx, b1, b2;
  auto load = [&]() ABRK_LAMBDA {
    osc6_rec_load<N, T, 1>(rec, c, S, G);
    ridx = rec[21];
    b1 = rec[rec_off_b1(N) + jc];
    b2 = rec[rec_off_b1(N) + N + jc];
  };
  load();
  sfor<21>([&](auto e) ABRK_LAMBDA { pin_loaded(S[e()]); });
  sfor<6>([&](auto r) ABRK_LAMBDA { pin_loaded(G[0][r()]); });
  pin_loaded(ridx);
  pin_loaded(b1);
  pin_loaded(b2);
  pin_loaded(mask);
  const int cnt = __builtin_amdgcn_readfirstlane(__popcll(mask));
  if (s0 >= cnt) return;
  if (cnt <= coop_rounds * slots) {
    for (int s = s0;;) {
      const bool row_ok = ridx >= T(0) && ridx < T(B);
      const long b = row_ok ? (long)ridx : 0;
      {
#pragma clang fp contract(off)
        T li[6], iq[6], y[6];
        osc6_rec_solve<N, T, 1, true>(rec, c, S, G, li, iq);
        ql_pinv_solve<6>(li, iq, G[0], y);  // this lane's column through the pseudo-inverse
        T a1 = T(-0.0), a2 = T(-0.0);
        sfor<6>([&](auto i) ABRK_LAMBDA {
          const T yu = lane_bcast(y[i()], N), yw = lane_bcast(y[i()], N + 1);
          a1 = Rm<T>::fma(G[0][i()], yu, a1);
          a2 = Rm<T>::fma(G[0][i()], yw, a2);
        });
        if (lane < N && row_ok) {
          const T ts = b1 - a1;
          ug[b * N + lane] = ts + b2 - (nulls ? a2 : T(0));
          if (tsg) tsg[b * N + lane] = ts;
        }
      }
      s += slots;
      if (s >= cnt) break;
      rec = recs + (j * kBlock + s) * rec_len(N);
      load();
    }
  } else if (s0 == 0 && lane < cnt) {
    rec = recs + (j * kBlock + lane) * rec_len(N);
    const T rix = rec[21];
    if (rix >= T(0) && rix < T(B)) {
      T u[N], ts[N];
      osc6_finish_row<N, T>(rec, nulls != 0, u, ts);
      store_row<N>(ug, (long)rix, u);
      if (tsg) store_row<N>(tsg, (long)rix, ts);
    }
  }
}

template <class A, class T, bool USE_C, int FEAT, bool NOTS>
__global__ void __launch_bounds__(kBlock, 2)
osc6_fused_kernel(A arm, OscP<T> P, long B, const T* __restrict__ qg, const T* __restrict__ dqg, const T* __restrict__ tg,
                  T* __restrict__ ug, T* __restrict__ tsg, unsigned long long* __restrict__ masks, int* __restrict__ ready,
                  T* __restrict__ rec, int slots, int coop_rounds, int nulls, int* __restrict__ err) {
  const long nchunk = (B + kBlock - 1) / kBlock;
  using V2 = typename LdsScratch<T, A::N>::V2;
  __shared__ T sctab[2 * kSinCosN];
  __shared__ V2 slab[slab_pairs<A::N>() * kBlock];
  const int lane = (int)threadIdx.x;
  if ((long)blockIdx.x < nchunk) {
    load_sincos_table(sctab, lane);
    const long b = (long)blockIdx.x * kBlock + lane;
    bool deferred = false;
    if (b < B) {
      std::conditional_t<NOTS, NoTs<DeferOnly<LdsScratch<T, A::N>>>, DeferOnly<LdsScratch<T, A::N>>> scr;
      scr.slab = slab;
      scr.lane = lane;
      scr.sctab = sctab;
      scr.allow_defer = true;
      scr.wl = reinterpret_cast<int*>(masks);
      scr.rec_base = rec;
      scr.handover = true;
      scr.wl_sub = 0;
      scr.wl_cap = 0;
      scr.row = b;
      osc_body<A, T, 6, USE_C, FEAT>(b, arm, P, B, qg, dqg, tg, (const T*)nullptr, (T*)nullptr, (const T*)nullptr, ug,
                                      NOTS ? (T*)nullptr : tsg, scr);
      deferred = scr.deferred;
    }
    const unsigned long long m = __ballot(deferred);
    if (lane == 0) masks[blockIdx.x] = m;
    // publish: every store of this wavefront (records, mask) before the count
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(ready + blockIdx.x, slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const long idx = (long)blockIdx.x - nchunk;
    const long j = idx % nchunk;
    const int s0 = (int)(idx / nchunk);
    int polls = 0;
    while (__hip_atomic_load(ready + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++polls > kSpinLimit) {
        if (lane == 0) *err = 1;
        return;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    finish_role<A::N, T>(j, s0, slots, masks, rec, nulls, coop_rounds, B, ug, tsg);
    if (lane == 0) __hip_atomic_fetch_add(ready + j, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct Timing {
  double us_per_step;
};

template <bool NOTS>
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
  double *q_, *dq_, *t_, *u2, *ts2, *u1, *ts1, *rec;
  const long nchunk = (B + kBlock - 1) / kBlock;
  CK(hipMalloc(&q_, B * N * 8));
  CK(hipMalloc(&dq_, B * N * 8));
  CK(hipMalloc(&t_, B * 6 * 8));
  CK(hipMalloc(&u1, B * N * 8));
  CK(hipMalloc(&ts1, B * N * 8));
  CK(hipMalloc(&u2, B * N * 8));
  CK(hipMalloc(&ts2, B * N * 8));
  CK(hipMalloc(&rec, (size_t)nchunk * kBlock * rec_len(N) * 8));
  CK(hipMemcpy(q_, q.data(), B * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dq_, dq.data(), B * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(t_, t.data(), B * 6 * 8, hipMemcpyHostToDevice));
  unsigned long long* masks;
  int *ready, *err;
  CK(hipMalloc(&masks, nchunk * 8));
  CK(hipMalloc(&ready, nchunk * 4));
  CK(hipMalloc(&err, 4));
  CK(hipMemset(ready, 0, nchunk * 4));
  CK(hipMemset(err, 0, 4));
  CK(hipMemset(u1, 0xff, B * N * 8));
  CK(hipMemset(u2, 0xff, B * N * 8));
  CK(hipMemset(ts1, 0xff, B * N * 8));
  CK(hipMemset(ts2, 0xff, B * N * 8));

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
  auto two = [&]() {
    hipLaunchKernelGGL((osc_kernel<A, T, 6, false, 0, 1, NOTS>), dim3((unsigned)nchunk), dim3(kBlock), 0, st, A{}, P, B,
                       (const T*)q_, (const T*)dq_, (const T*)t_, (const T*)nullptr, (T*)nullptr, (const T*)nullptr, u1,
                       NOTS ? (T*)nullptr : ts1, 1, reinterpret_cast<int*>(masks), rec);
    hipLaunchKernelGGL((osc6_finish_kernel<N, T>), dim3((unsigned)nchunk, (unsigned)slots), dim3(kBlock), 0, st,
                       (const unsigned long long*)masks, (const T*)rec, 0, rounds, B, u1, NOTS ? (T*)nullptr : ts1);
  };
  auto fused = [&]() {
    hipLaunchKernelGGL((osc6_fused_kernel<A, T, false, 0, NOTS>), dim3((unsigned)(nchunk * (1 + slots))), dim3(kBlock), 0, st,
                       A{}, P, B, (const T*)q_, (const T*)dq_, (const T*)t_, u2, NOTS ? (T*)nullptr : ts2, masks, ready, rec,
                       slots, rounds, 0, err);
  };
  auto time_graph = [&](auto&& fn) {
    fn();
    CK(hipStreamSynchronize(st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; k++) fn();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, ms = 0;
    std::vector<double> all;
    for (int rep = 0; rep < 8; rep++) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) all.push_back(ms * 1e3 / K);
      best = std::min(best, ms);
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    std::sort(all.begin(), all.end());
    return all[all.size() / 2];
  };
  const double t2 = time_graph(two);
  const double t1 = time_graph(fused);
  const double t2b = time_graph(two);
  const double t1b = time_graph(fused);
  std::vector<double> h1(B * N), h2(B * N), g1(B * N), g2(B * N);
  std::vector<int> rd(nchunk);
  int herr = 0;
  CK(hipMemcpy(h1.data(), u1, B * N * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), u2, B * N * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g1.data(), ts1, B * N * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g2.data(), ts2, B * N * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rd.data(), ready, nchunk * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  long diff_u = 0, diff_ts = 0, left = 0, nan = 0;
  for (long i = 0; i < B * N; i++) {
    diff_u += memcmp(&h1[i], &h2[i], 8) != 0;
    if (!NOTS) diff_ts += memcmp(&g1[i], &g2[i], 8) != 0;
    nan += !(h2[i] == h2[i]);
  }
  for (long j = 0; j < nchunk; j++) left += rd[j] != 0;
  fprintf(out,
          "  {\"rows\": %ld, \"nots\": %d, \"slots\": %d, \"rounds\": %d, \"graph_nodes\": %d, \"two_kernels_us_per_step\": [%.3f, %.3f], "
          "\"fused_us_per_step\": [%.3f, %.3f], \"u_values_that_differ\": %ld, \"ts_values_that_differ\": %ld, "
          "\"non_finite_u\": %ld, \"ready_words_left_nonzero\": %ld, \"spin_timeout_flag\": %d}%s\n",
          B, NOTS ? 1 : 0, slots, rounds, K, t2, t2b, t1, t1b, diff_u, diff_ts, nan, left, herr, last ? "" : ",");
  fflush(out);
  CK(hipFree(q_));
  CK(hipFree(dq_));
  CK(hipFree(t_));
  CK(hipFree(u1));
  CK(hipFree(u2));
  CK(hipFree(ts1));
  CK(hipFree(ts2));
  CK(hipFree(rec));
  CK(hipFree(masks));
  CK(hipFree(ready));
  CK(hipFree(err));
  CK(hipStreamDestroy(st));
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/dev/stdout";
  FILE* out = fopen(path, "w");
  if (!out) return 1;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  fprintf(out, "{\"device\": \"%s\", \"what\": \"UR5, all six task rows, fp64: hipGraph replay, median us per step of 6 replays; "
               "two kernels (first pass + finish) vs one fused launch, same inputs\", \"legs\": [\n", pr.name);
  run<true>(4096, 100, out, false);
  run<false>(4096, 100, out, false);
  run<true>(1024, 100, out, false);
  run<true>(8192, 100, out, false);
  run<true>(4000, 100, out, true);  // a partial last chunk
  fprintf(out, "]}\n");
  fclose(out);
  return 0;
}
