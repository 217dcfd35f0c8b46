// Law-only OSC kernels (caller-supplied J, M, g, ...) for 1..7 joints, both arithmetic types.
#include "abrk_kernels.h"
namespace abrk {
template <int N, class T>
static hipError_t law_launch(const LaunchArgs& la, const LawArgs& a) {
  hipLaunchKernelGGL((osc_law_kernel<N, T>), grid_for(la.B), dim3(kBlock), 0, la.stream,
                     *static_cast<const OscP<T>*>(a.P), la.B, (const T*)a.J, (const T*)a.M, (const T*)a.g,
                     (const T*)a.c, (const T*)a.xyz, (const T*)a.R, (const T*)a.q, (const T*)a.dq, (const T*)a.target,
                     (const T*)a.tv, (T*)a.ierr, (const T*)a.une, (T*)a.u, (T*)a.ts);
  return hipGetLastError();
}
hipError_t launch_osc_law(int n, int dtype, const LaunchArgs& la, const LawArgs& a) {
#define ABRK_CASE(NN) \
  case NN:            \
    return dtype == 0 ? law_launch<NN, double>(la, a) : law_launch<NN, float>(la, a);
  switch (n) {
    ABRK_CASE(1) ABRK_CASE(2) ABRK_CASE(3) ABRK_CASE(4) ABRK_CASE(5) ABRK_CASE(6) ABRK_CASE(7)
  }
#undef ABRK_CASE
  return hipErrorInvalidValue;
}
// ---- finish kernel, grouped form (round 5): the batch sizes where the per-chunk grid (chunks x slots) puts two WORKING
// wavefronts on one SIMD.  At 16384 rows 256 chunks x 4 slots fill the 1024 SIMDs, and the wavefront of slot s >= 4 lands
// on the SIMD of slot s - 4 of the same chunk - busy whenever the chunk holds five records or more (17 % of the chunks of
// random UR5 states): the kernel then lasts 12.0 us instead of 8.2 (profiles/round4/finish_per_chunk).  Here a GROUP of
// `gc` consecutive chunks shares 4 gc wavefronts: every wavefront reads the group's masks (one 8-byte value per lane,
// one coalesced request), numbers the group's records through - a pure function of the masks: no atomics, no counters -
// and wavefront i takes records i, i + 4 gc, ...: 46 +- 7 records on 64 wavefronts for gc = 16, so a second round is
// rare (0.6 % of the groups) where the per-chunk rule needed a fifth slot for every sixth chunk.  The price is the
// dependent chain mask -> record (one more memory round trip, ~0.8 us), which is why the smaller batches - every
// (chunk, slot) has a SIMD of its own there - keep the per-chunk form.  Same arithmetic, same bits.
// A group with more records than `coop_max`: one record per lane, wavefront i < gc takes chunk i of the group.
template <int N, class T>
__global__ void __launch_bounds__(kBlock)
osc6_finish_group_kernel(const unsigned long long* __restrict__ masks, const T* __restrict__ recs, int nulls, int gc,
                         long nchunk, int coop_max, long B, T* __restrict__ ug, T* __restrict__ tsg) {
  const int lane = (int)threadIdx.x;
  const long c0 = (long)blockIdx.x * gc;
  const int wi = (int)blockIdx.y, wg = (int)gridDim.y;
  unsigned long long m = 0ull;
  if (lane < gc && c0 + lane < nchunk) m = masks[c0 + lane];
  const int cnt = __popcll(m);
  int incl = cnt;
  for (int d = 1; d < kBlock; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (lane >= d) incl += v;
  }
  const int total = __builtin_amdgcn_readlane(incl, kBlock - 1);
  if (total == 0) return;
  const int c = lane < N + 2 ? lane : N + 1;  // (idle lanes shadow the last column)
  const int jc = lane < N ? lane : 0;
  if (total <= coop_max) {
    // (a wavefront beyond the group's record count has nothing to do HERE; in the one-record-per-lane branch below
    //  wavefront wi owns CHUNK wi, whatever `total` is - coop_max == 0, i.e. ABRK_FINISH_ROUNDS=0, sends every group there)
    for (int r = wi; r < total; r += wg) {
      const unsigned long long above = __ballot(incl > r);  // the chunk that holds record r: the first lane whose count passes it
      const int ch = __builtin_ctzll(above);
      const int k = r - (__builtin_amdgcn_readlane(incl, ch) - __builtin_amdgcn_readlane(cnt, ch));
      const T* rec = recs + ((c0 + ch) * kBlock + k) * rec_len(N);
      T S[21], G[1][6];
      osc6_rec_load<N, T, 1>(rec, c, S, G);
      const T rix = rec[21];
      const bool row_ok = rix >= T(0) && rix < T(B);  // (a record's row index is data: nothing is stored outside [0, B))
      const long b = row_ok ? (long)rix : 0;
      const T b1 = rec[rec_off_b1(N) + jc], b2 = rec[rec_off_b1(N) + N + jc];
      {
#pragma clang fp contract(off)  // the same bits as osc6_tail / osc6_finish_kernel
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
    }
  } else if (wi < gc && c0 + wi < nchunk) {
    const int mine = __builtin_amdgcn_readlane(cnt, wi);
    if (lane < mine) {
      const T* rec = recs + ((c0 + wi) * kBlock + lane) * rec_len(N);
      const T rix = rec[21];
      if (rix >= T(0) && rix < T(B)) {
        T u[N], ts[N];
        osc6_finish_row<N, T>(rec, nulls != 0, u, ts);
        store_row<N>(ug, (long)rix, u);
        if (tsg) store_row<N>(tsg, (long)rix, ts);
      }
    }
  }
}
// ---- finish kernel, dense form (round 6): batches beyond 65 536 rows, where the deferred rows are many enough to fill
// wavefronts ONE RECORD PER LANE.  Until round 5 such batches took the recompute form - a second pass of the complete
// row program over a worklist (kinematics, dynamics, Jacobian, the whole law again for 4.6 % of the rows: ~7000
// instructions per row, 124 us of an 8 M-row step's 752).  Here a group of 64 consecutive chunks (4096 rows; 188 +- 13
// records for random UR5 states) is numbered through from its 64 masks - lane l holds chunk l's mask, one wave scan, no
// atomics - and wavefront w takes records 64 w .. 64 w + 63 of that numbering, each lane finding its record's chunk by a
// binary search over the scan (six __shfl) and finishing it from the record alone (osc6_finish_row: the lane form,
// ~5200 instructions).  Lanes are 98 % occupied (the recompute pass packs 100 %, but runs the 1800-instruction
// kinematics on top).  Grid: groups x 4 wavefronts, a wavefront loops while the group has more records (> 256 of 4096
// rows deferring: dense fuzz arms, near-singular sets).  Same arithmetic, same bits as every other form.
template <int N, class T>
__global__ void __launch_bounds__(kBlock)
osc6_finish_dense_kernel(const unsigned long long* __restrict__ masks, const T* __restrict__ recs, int nulls,
                         long nchunk, long B, T* __restrict__ ug, T* __restrict__ tsg) {
  const int lane = (int)threadIdx.x;
  const long c0 = (long)blockIdx.x * kBlock;
  unsigned long long m = 0ull;
  if (c0 + lane < nchunk) m = masks[c0 + lane];
  const int cnt = __popcll(m);
  int incl = cnt;
  for (int d = 1; d < kBlock; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (lane >= d) incl += v;
  }
  const int total = __builtin_amdgcn_readlane(incl, kBlock - 1);
  for (int w = (int)blockIdx.y; w * kBlock < total; w += (int)gridDim.y) {
    const int r = w * kBlock + lane;
    const int rr = r < total ? r : total - 1;  // (idle lanes of the last block shadow its last record: the shuffles stay uniform)
    int ch = 0;  // the chunk that holds record rr: the first lane whose inclusive count passes it
    for (int step = kBlock / 2; step >= 1; step >>= 1) {
      const int v = __shfl(incl, ch + step - 1);
      if (v <= rr) ch += step;
    }
    const int k = rr - (__shfl(incl, ch) - __shfl(cnt, ch));
    if (r < total) {
      const T* rec = recs + ((c0 + ch) * kBlock + k) * rec_len(N);
      const T rix = rec[21];
      if (rix >= T(0) && rix < T(B)) {  // (a record's row index is data: nothing is stored outside [0, B))
        T u[N], ts[N];
        osc6_finish_row<N, T>(rec, nulls != 0, u, ts);
        store_row<N>(ug, (long)rix, u);
        if (tsg) store_row<N>(tsg, (long)rix, ts);
      }
    }
  }
}
template <int N, class T>
static hipError_t finish_launch(const LaunchArgs& la, const FinishArgs& a) {
  const unsigned nchunk = (unsigned)((la.B + kBlock - 1) / kBlock);
  if (a.group < 0) {  // (FinishArgs::group == -1: the dense form)
    hipLaunchKernelGGL((osc6_finish_dense_kernel<N, T>), dim3((nchunk + kBlock - 1) / kBlock, 4u), dim3(kBlock), 0,
                       la.stream, (const unsigned long long*)a.masks, (const T*)a.rec, a.nulls, (long)nchunk, la.B,
                       (T*)a.u, (T*)a.ts);
    return hipGetLastError();
  }
  if (a.group > 0) {
    // grouped form: four wavefronts per chunk; a group with more than coop_rounds x its wavefronts goes one record per lane
    const int gc = a.group, wg = 4 * gc;
    hipLaunchKernelGGL((osc6_finish_group_kernel<N, T>), dim3((nchunk + gc - 1) / gc, (unsigned)wg), dim3(kBlock), 0,
                       la.stream, (const unsigned long long*)a.masks, (const T*)a.rec, a.nulls, gc, (long)nchunk,
                       a.coop_rounds * wg, la.B, (T*)a.u, (T*)a.ts);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((osc6_finish_kernel<N, T>), dim3(nchunk, (unsigned)a.slots), dim3(kBlock), 0, la.stream,
                     (const unsigned long long*)a.masks, (const T*)a.rec, a.nulls, a.coop_rounds, la.B, (T*)a.u, (T*)a.ts);
  return hipGetLastError();
}
hipError_t launch_osc6_finish(int n, int dtype, const LaunchArgs& la, const FinishArgs& a) {
  // (group: at most 16 chunks - four wavefronts per chunk - so that one lane per chunk holds the group's masks)
  if (a.slots < 1 || a.slots > kBlock || a.group < -1 || a.group > 16 || la.B < 1 || (a.group >= 0 && la.B > kHandoverMaxRows))
    return hipErrorInvalidValue;
#define ABRK_CASE(NN) \
  case NN:            \
    return dtype == 0 ? finish_launch<NN, double>(la, a) : finish_launch<NN, float>(la, a);
  switch (n) {
    ABRK_CASE(1) ABRK_CASE(2) ABRK_CASE(3) ABRK_CASE(4) ABRK_CASE(5) ABRK_CASE(6) ABRK_CASE(7)
  }
#undef ABRK_CASE
  return hipErrorInvalidValue;
}
template <int N, class T>
static hipError_t limits_launch(const LaunchArgs& la, const void* P, const void* q, void* u, int acc) {
  hipLaunchKernelGGL((limits_kernel<N, T>), grid_for(la.B), dim3(kBlock), 0, la.stream,
                     *static_cast<const LimitsP<T>*>(P), la.B, (const T*)q, (T*)u, acc);
  return hipGetLastError();
}
hipError_t launch_limits(int n, int dtype, const LaunchArgs& la, const void* P, const void* q, void* u, int acc) {
#define ABRK_CASE(NN) \
  case NN:            \
    return dtype == 0 ? limits_launch<NN, double>(la, P, q, u, acc) : limits_launch<NN, float>(la, P, q, u, acc);
  switch (n) {
    ABRK_CASE(1) ABRK_CASE(2) ABRK_CASE(3) ABRK_CASE(4) ABRK_CASE(5) ABRK_CASE(6) ABRK_CASE(7)
  }
#undef ABRK_CASE
  return hipErrorInvalidValue;
}
template <int N, class T>
static hipError_t mx_launch(const LaunchArgs& la, int k, double thr, const void* M, const void* J, void* Mx, void* Minv) {
  hipLaunchKernelGGL((mx_kernel<N, T>), grid_for(la.B), dim3(kBlock), 0, la.stream, la.B, k, T(thr), (const T*)M,
                     (const T*)J, (T*)Mx, (T*)Minv);
  return hipGetLastError();
}
hipError_t launch_osc_mx(int n, int dtype, const LaunchArgs& la, int k, double thr, const void* M, const void* J,
                         void* Mx, void* Minv) {
#define ABRK_CASE(NN) \
  case NN:            \
    return dtype == 0 ? mx_launch<NN, double>(la, k, thr, M, J, Mx, Minv) : mx_launch<NN, float>(la, k, thr, M, J, Mx, Minv);
  switch (n) {
    ABRK_CASE(1) ABRK_CASE(2) ABRK_CASE(3) ABRK_CASE(4) ABRK_CASE(5) ABRK_CASE(6) ABRK_CASE(7)
  }
#undef ABRK_CASE
  return hipErrorInvalidValue;
}
hipError_t launch_velocity_limiting(int dtype, const LaunchArgs& la, const double (&g)[5], const void* in, void* out) {
  if (dtype == 0)
    hipLaunchKernelGGL((velocity_limiting_kernel<double>), grid_for(la.B), dim3(kBlock), 0, la.stream, la.B, g[0], g[1],
                       g[2], g[3], g[4], (const double*)in, (double*)out);
  else
    hipLaunchKernelGGL((velocity_limiting_kernel<float>), grid_for(la.B), dim3(kBlock), 0, la.stream, la.B, float(g[0]),
                       float(g[1]), float(g[2]), float(g[3]), float(g[4]), (const float*)in, (float*)out);
  return hipGetLastError();
}
hipError_t launch_orientation_forces(int dtype, const LaunchArgs& la, int alg, const void* R, const void* abg, void* out) {
  if (dtype == 0)
    hipLaunchKernelGGL((orientation_forces_kernel<double>), grid_for(la.B), dim3(kBlock), 0, la.stream, la.B, alg,
                       (const double*)R, (const double*)abg, (double*)out);
  else
    hipLaunchKernelGGL((orientation_forces_kernel<float>), grid_for(la.B), dim3(kBlock), 0, la.stream, la.B, alg,
                       (const float*)R, (const float*)abg, (float*)out);
  return hipGetLastError();
}
hipError_t launch_transformations(int dtype, const LaunchArgs& la, int op, const void* a, const void* b, void* out) {
  if (dtype == 0)
    hipLaunchKernelGGL((transformations_kernel<double>), grid_for(la.B), dim3(kBlock), 0, la.stream, la.B, op,
                       (const double*)a, (const double*)b, (double*)out);
  else
    hipLaunchKernelGGL((transformations_kernel<float>), grid_for(la.B), dim3(kBlock), 0, la.stream, la.B, op,
                       (const float*)a, (const float*)b, (float*)out);
  return hipGetLastError();
}
hipError_t launch_twolink_step(int dtype, const LaunchArgs& la, const void* K, void* q, void* dq, const void* u) {
  if (dtype == 0)
    hipLaunchKernelGGL((twolink_step_kernel<double>), grid_for(la.B), dim3(kBlock), 0, la.stream,
                       *static_cast<const TwoLinkP<double>*>(K), la.B, (double*)q, (double*)dq, (const double*)u);
  else
    hipLaunchKernelGGL((twolink_step_kernel<float>), grid_for(la.B), dim3(kBlock), 0, la.stream,
                       *static_cast<const TwoLinkP<float>*>(K), la.B, (float*)q, (float*)dq, (const float*)u);
  return hipGetLastError();
}
}  // namespace abrk
