// abrk_coop.h - the WAVE-COOPERATIVE mapping of OSC.generate that BASELINE.json's north_star names: K lanes per arm
// instance (K = 4, 8 or 16; 64 / K arms per wavefront), the frame chain, the per-link Jacobian columns and the mass
// matrix staged in LDS, the products behind M = sum_l J_l^T D_l J_l spread over the lanes of the group.
//
// It exists beside the lane-per-arm kernels (abrk_kernels.h) to be MEASURED against them at the config-sized batch
// (B = 4096 is 64 wavefronts lane-per-arm, i.e. 64 of the chip's 1024 SIMDs): profiles/round2/coop_ab.md.  Scope: the
// plain law of BASELINE config 2 - orthogonal-chain built-in arm, fp64, task rows x,y,z of the EE, no secondary
// controllers (the optional inputs of the FEAT = 2 kernels are not duplicated here).
//
// Phases of one wavefront (LDS round trips between them; a workgroup is one wavefront, so a phase boundary is a
// wave-level wait on the LDS counter, not a multi-wave barrier):
//   1. sin/cos of the joint angles            lane r of a group takes joints r, r + K, ...
//   2. forward kinematics by ROWS             lane r < 3 carries row r of every frame (a row of R Rz S only needs that
//                                             row: no cross-lane traffic); writes component r of z_i, o_i, p_l
//   3. Jacobian columns e_{l,i} = z_i x (p_l - o_i), l = 1..N (links) and the EE   one (l, i) task per lane and slot
//   4. M_ij = sum_{l > i} m_l e_{l,i} . e_{l,j} + z_i^T Ibar_i z_j,  g_i       one (i, j) pair per lane and slot
//   5. the control law proper (osc_law - the very code of the lane-per-arm kernel: Cholesky of M, J M^-1 J^T, the
//      inv / pinv gate, gains, J^T Mx u_task, gravity) on lane 0 of the group, from the staged M, g, J, p.
// What is computed is the reference's osc.py:217-320 / base_config.py:210-285 as everywhere else in this library.
#pragma once
#include "abrk_kernels.h"

namespace abrk {

// tri-order tables packed 3 bits per entry: index t = row (row + 1) / 2 + col, col <= row, up to 21 entries
constexpr unsigned long long tri_pack(int n, bool want_row) {
  unsigned long long v = 0;
  int t = 0;
  for (int r = 0; r < n; r++)
    for (int c = 0; c <= r; c++, t++) v |= (unsigned long long)(want_row ? r : c) << (3 * t);
  return v;
}

template <class A, int K>
struct CoopLayout {
  static constexpr int N = A::N;
  static constexpr int G = 64 / K;               // arms per wavefront
  static constexpr int NT = N * (N + 1) / 2;     // link tasks (l, i), i < l <= N; also the pairs (i, j), j <= i < N
  // field offsets in doubles per arm
  static constexpr int SC = 0;                   // [N][2]
  static constexpr int Z = SC + 2 * N;           // [N][3]
  static constexpr int O = Z + 3 * N;            // [N][3]
  static constexpr int PL = O + 3 * N;           // [N + 1][3]: COM of link 1..N, then the EE
  static constexpr int E = PL + 3 * (N + 1);     // [NT + N][3]: e_{l,i} in tri order, then the EE columns
  static constexpr int M = E + 3 * (NT + N);     // [NT] lower triangle
  static constexpr int GZ = M + NT;              // [N]
  static constexpr int SIZE = GZ + N;
};

template <class A, int K>
__global__ void __launch_bounds__(64) osc_coop_kernel(A arm, OscP<double> P, long B, const double* __restrict__ qg,
                                                      const double* __restrict__ dqg, const double* __restrict__ tg,
                                                      double* __restrict__ ug, double* __restrict__ tsg) {
  using T = double;
  using Lay = CoopLayout<A, K>;
  constexpr int N = A::N, G = Lay::G, NT = Lay::NT;
  static_assert(A::kStatic && A::kOrtho, "cooperative variant: built-in orthogonal chains");
  static_assert(N <= 7 && NT <= 21, "3-bit packed task tables");
  __shared__ T S[Lay::SIZE * G];
  const int lane = threadIdx.x, r = lane % K, a = lane / K;
  const long row = (long)blockIdx.x * G + a;
  const bool active = row < B;
  const long rowc = active ? row : B - 1;
  auto at = [&](int field, int idx) ABRK_LAMBDA -> T& { return S[(field + idx) * G + a]; };
  constexpr unsigned long long TROW = tri_pack(N, true), TCOL = tri_pack(N, false);

  // ---- 1. sin / cos: joint j on lane j mod K
  for (int j = r; j < N; j += K) {
    T s, c;
    Rm<T>::sincos(qg[rowc * N + j], s, c);
    at(Lay::SC, 2 * j) = s;
    at(Lay::SC, 2 * j + 1) = c;
  }
  __syncthreads();

  // ---- 2. forward kinematics, one matrix row per lane (lanes 0..2 of the group)
  if (r < 3) {
    auto pick = [&](T v0, T v1, T v2) ABRK_LAMBDA { return r == 0 ? v0 : (r == 1 ? v1 : v2); };
    T Rr[3], orr;
    sfor<3>([&](auto c) ABRK_LAMBDA {
      Rr[c()] = pick(AccJ0<A, T, 0 * 4 + c()>::get(arm), AccJ0<A, T, 1 * 4 + c()>::get(arm), AccJ0<A, T, 2 * 4 + c()>::get(arm));
    });
    orr = pick(AccJ0<A, T, 3>::get(arm), AccJ0<A, T, 7>::get(arm), AccJ0<A, T, 11>::get(arm));
    T sv[N][2];  // all sin / cos up front: one LDS round trip instead of one per joint
    sfor<N>([&](auto i) ABRK_LAMBDA {
      sv[i()][0] = at(Lay::SC, 2 * i());
      sv[i()][1] = at(Lay::SC, 2 * i() + 1);
    });
    sfor<N>([&](auto ii) ABRK_LAMBDA {
      constexpr int i = ii();
      at(Lay::Z, 3 * i + r) = Rr[2];
      at(Lay::O, 3 * i + r) = orr;
      const T s = sv[i][0], c = sv[i][1];
      const T X[3] = {c * Rr[0] + s * Rr[1], c * Rr[1] - s * Rr[0], Rr[2]};  // row r of joint_i Rz(q_i)
      // COM of link i + 1: row r of  o + X B_t
      T p = orr;
      sfor<3>([&](auto k) ABRK_LAMBDA { p = cfma<AccB<A, T, i, k() * 4 + 3>>(arm, X[k()], p); });
      at(Lay::PL, 3 * i + r) = p;
      if constexpr (i + 1 < N) {
        T R2[3], o2 = orr;
        sfor<3>([&](auto cc) ABRK_LAMBDA {
          T acc = T(-0.0);
          sfor<3>([&](auto k) ABRK_LAMBDA { acc = cfma<AccS<A, T, i, k() * 4 + cc()>>(arm, X[k()], acc); });
          R2[cc()] = acc;
        });
        sfor<3>([&](auto k) ABRK_LAMBDA { o2 = cfma<AccS<A, T, i, k() * 4 + 3>>(arm, X[k()], o2); });
        sfor<3>([&](auto cc) ABRK_LAMBDA { Rr[cc()] = R2[cc()]; });
        orr = o2;
      } else {
        T pe = orr;  // the EE hangs off the last rotated joint frame
        sfor<3>([&](auto k) ABRK_LAMBDA { pe = cfma<AccBE<A, T, k() * 4 + 3>>(arm, X[k()], pe); });
        at(Lay::PL, 3 * N + r) = pe;
      }
    });
  }
  __syncthreads();

  // ---- 3. Jacobian columns: task t < NT is (link l = row + 1, joint i = col) in tri order; t >= NT the EE columns
  for (int t = r; t < NT + N; t += K) {
    const bool ee = t >= NT;
    const int l = ee ? N : (int)((TROW >> (3 * t)) & 7) + 1;
    const int i = ee ? t - NT : (int)((TCOL >> (3 * t)) & 7);
    const int pl = ee ? N : l - 1;
    T z[3], d[3];
    sfor<3>([&](auto c) ABRK_LAMBDA {
      z[c()] = at(Lay::Z, 3 * i + c());
      d[c()] = at(Lay::PL, 3 * pl + c()) - at(Lay::O, 3 * i + c());
    });
    T e[3];
    cross3(z, d, e);
    sfor<3>([&](auto c) ABRK_LAMBDA { at(Lay::E, 3 * t + c()) = e[c()]; });
  }
  __syncthreads();

  // ---- 4. M and g: pair t = (i = row, j = col); links l = i + 1 .. N contribute m_l e_{l,i} . e_{l,j}
  for (int t = r; t < NT; t += K) {
    const int i = (int)((TROW >> (3 * t)) & 7), j = (int)((TCOL >> (3 * t)) & 7);
    T acc = T(0), gz = T(0);
    sfor<N>([&](auto ll) ABRK_LAMBDA {
      constexpr int l = ll() + 1;
      if constexpr (l < A::NL && link_has_linear_mass<A, l>()) {
        // branch-free: every link's columns are read (the indices stay inside the table for i >= l too) and the
        // contribution of links that joint i does not move is weighted out - the six LDS round trips overlap
        const int ti = (l - 1) * l / 2 + i, tj = (l - 1) * l / 2 + j;
        const T w = (l > i) ? T(1) : T(0);
        T part = T(0);
        sfor<3>([&](auto c) ABRK_LAMBDA {
          part = Rm<T>::fma(T(A::MD(l, c())) * at(Lay::E, 3 * ti + c()), at(Lay::E, 3 * tj + c()), part);
        });
        acc = Rm<T>::fma(w, part, acc);
        gz = Rm<T>::fma(w * T(A::MD(l, 2)), at(Lay::E, 3 * ti + 2), gz);  // kept by the lane with j == 0
      }
    });
    // angular part: z_i^T Ibar_i z_j with the compile-time suffix sums of the inertia diagonals (world frame,
    // base_config.py:628), selected by the runtime i
    T Ib[3] = {T(0), T(0), T(0)};
    sfor<N>([&](auto m) ABRK_LAMBDA {
      if (m() == i) sfor<3>([&](auto c) ABRK_LAMBDA { Ib[c()] = T(A::Isuf(m(), c())); });
    });
    sfor<3>([&](auto c) ABRK_LAMBDA { acc = Rm<T>::fma(Ib[c()] * at(Lay::Z, 3 * i + c()), at(Lay::Z, 3 * j + c()), acc); });
    at(Lay::M, t) = acc;
    if (j == 0) at(Lay::GZ, i) = gz;
  }
  __syncthreads();

  // ---- 5. the law on lane 0 of the group
  if (r == 0) {
    T Ms[NT], gzv[N], Jv[N][3], Jw[N][3], p[3], RF[9], q[N], dq[N], tgt[6], tv[6], ie[6], une[N], u[N], ts[N];
    sfor<NT>([&](auto e) ABRK_LAMBDA { Ms[e()] = at(Lay::M, e()); });
    sfor<N>([&](auto i) ABRK_LAMBDA {
      gzv[i()] = at(Lay::GZ, i());
      sfor<3>([&](auto c) ABRK_LAMBDA {
        Jv[i()][c()] = at(Lay::E, 3 * (NT + i()) + c());
        Jw[i()][c()] = at(Lay::Z, 3 * i() + c());
      });
      q[i()] = T(0);
      une[i()] = T(0);
    });
    sfor<3>([&](auto c) ABRK_LAMBDA { p[c()] = at(Lay::PL, 3 * N + c()); });
    sfor<9>([&](auto e) ABRK_LAMBDA { RF[e()] = T(0); });
    sfor<6>([&](auto e) ABRK_LAMBDA { tv[e()] = ie[e()] = T(0); });
    load_row<N>(dqg, rowc, dq);
    load_row<6>(tg, rowc, tgt);
    osc_law<N, T, 3, false, 0>(P, Ms, gzv, T(9.81), gzv, Jv, Jw, p, RF, q, dq, tgt, false, tv, false, ie, false, une, u, ts);
    if (active) {
      store_row<N>(ug, row, u);
      if (tsg) store_row<N>(tsg, row, ts);
    }
  }
}

struct CoopArgs {
  const void* P;  // OscP<double>
  int lanes;      // 4, 8 or 16
  const void *q, *dq, *target;
  void *u, *ts;
};

template <class A>
hipError_t launch_osc_coop(const LaunchArgs& la, const CoopArgs& a) {
  auto go = [&](auto k) {
    constexpr int K = decltype(k)::value;
    constexpr int G = 64 / K;
    hipLaunchKernelGGL((osc_coop_kernel<A, K>), dim3((unsigned)((la.B + G - 1) / G)), dim3(64), 0, la.stream, A{},
                       *static_cast<const OscP<double>*>(a.P), la.B, (const double*)a.q, (const double*)a.dq,
                       (const double*)a.target, (double*)a.u, (double*)a.ts);
  };
  if (a.lanes == 4) go(ic<4>{});
  else if (a.lanes == 8) go(ic<8>{});
  else if (a.lanes == 16) go(ic<16>{});
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace abrk
