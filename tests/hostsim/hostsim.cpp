// hostsim.cpp - TEST AID ONLY.  Compiles the row programs of abr_control_amd/csrc
// (abrk_rows.h: exactly what one GPU lane executes) for the HOST so that the kernel
// arithmetic can be checked against the oracle in a container without a GPU, before GPU
// minutes are spent.  Nothing in abr_control_amd/ loads this library; the product path is
// libabrk.so and fails loudly without a HIP device.
#define ABRK_HD __host__ __device__
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../abr_control_amd/csrc/abrk_params.h"
#include "../../abr_control_amd/csrc/abrk_rows.h"
#include "../../abr_control_amd/csrc/abrk_rt.h"

using namespace abrk;

// The library is built in parts (parallel compilation, see tests/hostsim/__init__.py):
//   HOSTSIM_STATIC=1            the five built-in arms
//   HOSTSIM_RT_LO / HOSTSIM_RT_HI   runtime-table arms with LO..HI joints
//   HOSTSIM_LAW=1               the law-only row program
#ifndef HOSTSIM_STATIC
#define HOSTSIM_STATIC 0
#endif
#ifndef HOSTSIM_RT_LO
#define HOSTSIM_RT_LO 1
#define HOSTSIM_RT_HI 0
#endif
#ifndef HOSTSIM_LAW
#define HOSTSIM_LAW 0
#endif

namespace {
template <class A, class T>
int run_dyn(const A& arm, int n, int64_t B, const void* q, const void* dq, int frame, const double* off,
            unsigned want, void* const* outs) {
  DynOutP<T> o;
  T** po = reinterpret_cast<T**>(&o);
  for (int i = 0; i < 10; i++) po[i] = static_cast<T*>(outs[i]);
  int m = frame_m(frame, n);
  T ox = T(off ? off[0] : 0), oy = T(off ? off[1] : 0), oz = T(off ? off[2] : 0);
  DirectStore<T> st;
  for (long b = 0; b < B; b++) {
    if (want & (W_C | W_DJ))
      dyn_body<A, T, true>(b, true, st, arm, frame, m, ox, oy, oz, want, (long)B, (const T*)q, (const T*)dq, o);
    else
      dyn_body<A, T, false>(b, true, st, arm, frame, m, ox, oy, oz, want, (long)B, (const T*)q, (const T*)dq, o);
  }
  return 0;
}
template <class A, class T>
int run_osc(const A& arm, int n, const abrk_osc_params* P, int64_t B, const void* q, const void* dq,
            const void* tg, const void* tv, void* ie, const void* une, void* u, void* ts) {
  OscP<T> p = make_oscp<T>(*P, n);
  int fast = osc_fast_rows(*P, n, une != nullptr);
  if (P->ki == 0) ie = nullptr;
  const int feat = (tv || ie || une) ? 2 : (p.n_null > 0 ? 1 : 0);  // same dispatch rule as Launch::osc_launch_feat
  for (long b = 0; b < B; b++) {
#define CALL1(KM, UC, FT)                                                                                  \
  osc_body<A, T, KM, UC, FT>(b, arm, p, (long)B, (const T*)q, (const T*)dq, (const T*)tg, (const T*)tv, (T*)ie, \
                             (const T*)une, (T*)u, (T*)ts)
#define CALL(KM, UC)                      \
  do {                                    \
    if (feat == 2) CALL1(KM, UC, 2);      \
    else if (feat == 1) CALL1(KM, UC, 1); \
    else CALL1(KM, UC, 0);                \
  } while (0)
    if (fast == 3) {
      if (P->use_C) CALL(3, true); else CALL(3, false);
    } else if (fast == 2) {
      if constexpr (A::N <= 3) {
        if (P->use_C) CALL(2, true); else CALL(2, false);
      }
    } else if (feat == 0 && !ts) {
      // what the first pass of the plain six-row kernels runs when no training signal is asked for (NoTs: the gravity
      // term folded into the velocity term ahead of the factorisations)
      NoTs<RegScratch<T, A::N>> scr;
      if (P->use_C)
        osc_body<A, T, 6, true, 0>(b, arm, p, (long)B, (const T*)q, (const T*)dq, (const T*)tg, (const T*)tv, (T*)ie,
                                   (const T*)une, (T*)u, (T*)ts, scr);
      else
        osc_body<A, T, 6, false, 0>(b, arm, p, (long)B, (const T*)q, (const T*)dq, (const T*)tg, (const T*)tv, (T*)ie,
                                    (const T*)une, (T*)u, (T*)ts, scr);
    } else {
      if (P->use_C) CALL(6, true); else CALL(6, false);
    }
#undef CALL
#undef CALL1
  }
  return 0;
}
// the six-row law in its two-pass, hand-over form (what libabrk launches for batches of up to 262144 rows): the first
// pass compiled WITHOUT the eigen-decomposition (DeferOnly), a deferring row parks itself in the worklist and leaves its
// record (osc_law6), osc6_finish_row then completes it from the record alone - here with all N + 2 columns on one
// "lane"; the GPU's wave-cooperative kernel runs the same arithmetic with one column per lane
template <class A, class T>
int run_osc_handover(const A& arm, int n, const abrk_osc_params* P, int64_t B, const void* q, const void* dq,
                     const void* tg, const void* tv, void* ie, const void* une, void* u, void* ts,
                     int64_t* n_deferred) {
  OscP<T> p = make_oscp<T>(*P, n);
  if (osc_fast_rows(*P, n, une != nullptr) != 0) return -1;
  if (P->ki == 0) ie = nullptr;
  const int feat = (tv || ie || une) ? 2 : (p.n_null > 0 ? 1 : 0);
  const bool nulls = p.n_null > 0 || une != nullptr;
  // hand-over mode as on the device, rows one by one: the record of row b at rec[b] (the device packs a chunk's records
  // at the chunk's first slots - ScratchBase::record - and carries the row's index in the record); which rows deferred is
  // the kernel's business there (a ballot per 64-row chunk) and a plain flag array here
  std::vector<T> rec((size_t)B * rec_len(A::N), T(0));
  std::vector<char> flag((size_t)B, 0);
  auto first = [&](long b, auto& scr, auto uc, auto ft) {
    scr.allow_defer = true;
    scr.rec_base = rec.data();
    scr.handover = true;
    scr.row = b;
    osc_body<A, T, 6, uc(), ft()>(b, arm, p, (long)B, (const T*)q, (const T*)dq, (const T*)tg, (const T*)tv, (T*)ie,
                                  (const T*)une, (T*)u, (T*)ts, scr);
    flag[b] = scr.deferred;
  };
  using std::integral_constant;
  for (long b = 0; b < B; b++) {
    auto with_feat = [&](auto uc) {
      if (feat == 0 && !ts) {
        NoTs<DeferOnly<RegScratch<T, A::N>>> scr;
        first(b, scr, uc, integral_constant<int, 0>{});
      } else {
        DeferOnly<RegScratch<T, A::N>> scr;
        if (feat == 2) first(b, scr, uc, integral_constant<int, 2>{});
        else if (feat == 1) first(b, scr, uc, integral_constant<int, 1>{});
        else first(b, scr, uc, integral_constant<int, 0>{});
      }
    };
    if (P->use_C) with_feat(integral_constant<bool, true>{});
    else with_feat(integral_constant<bool, false>{});
  }
  int64_t total = 0;
  for (long b = 0; b < B; b++) {
    if (!flag[b]) continue;
    total++;
    T uu[A::N], tt[A::N];
    osc6_finish_row<A::N, T>(rec.data() + b * rec_len(A::N), nulls, uu, tt);
    store_row<A::N>((T*)u, b, uu);
    if (ts) store_row<A::N>((T*)ts, b, tt);
  }
  if (n_deferred) *n_deferred = total;
  return 0;
}
// the fused "u + Tx, J, M, g" row program (osc_full_body; FEAT 0 / 2 and KM 3 / 6 as Launch::osc_full dispatches)
template <class A, class T>
int run_osc_full(const A& arm, int n, const abrk_osc_params* P, int64_t B, const void* q, const void* dq,
                 const void* tg, const void* tv, void* ie, const void* une, void* u, void* ts, unsigned want,
                 void* const* outs) {
  OscP<T> p = make_oscp<T>(*P, n);
  const int fast = osc_fast_rows(*P, n, une != nullptr);
  if (P->ki == 0) ie = nullptr;
  const bool plain = !(tv || ie || une) && p.n_null == 0;
  DynOutP<T> o{};
  o.Tx = (T*)outs[0];
  o.J = (T*)outs[1];
  o.M = (T*)outs[2];
  o.g = (T*)outs[3];
  o.C = (T*)outs[4];
  o.dJ = (T*)outs[5];
  const bool vel = (want & (W_C | W_DJ)) != 0;
  DirectStore<T> st;
  RegScratch<T, A::N> scr;
  for (long b = 0; b < B; b++) {
#define FULL(KM, UC, FT, VEL)                                                                                            \
  osc_full_body<A, T, KM, UC, FT, VEL>(b, true, st, arm, p, (long)B, (const T*)q, (const T*)dq, (const T*)tg, (const T*)tv, \
                                       (T*)ie, (const T*)une, (T*)u, (T*)ts, want, o, scr)
#define FULLF(KM, UC)                   \
  do {                                  \
    if (vel) FULL(KM, UC, 2, true);     \
    else if (plain) FULL(KM, UC, 0, false); \
    else FULL(KM, UC, 2, false);        \
  } while (0)
    if (fast == 3) {
      if (P->use_C) FULLF(3, true); else FULLF(3, false);
    } else {
      if (P->use_C) FULLF(6, true); else FULLF(6, false);
    }
#undef FULLF
#undef FULL
  }
  return 0;
}
template <class A, class T>
int run_sliding(const A& arm, int n, const abrk_sliding_params* P, int64_t B, const void* q, const void* dq,
                const void* tg, const void* tv, const void* ta, void* u, void* s) {
  SlidingP<T> p = make_slidingp<T>(*P, n);
  for (long b = 0; b < B; b++)
    sliding_body<A, T>(b, arm, p, (long)B, (const T*)q, (const T*)dq, (const T*)tg, (const T*)tv, (const T*)ta,
                       (T*)u, (T*)s);
  return 0;
}
template <class A, class T>
int run_joint(const A& arm, int n, const abrk_null_ctrl* c, int grav, int64_t B, const void* q, const void* dq,
              const void* tg, const void* tv, void* u) {
  JointP<T> p = make_jointp<T>(*c, grav);
  for (long b = 0; b < B; b++)
    joint_body<A, T>(b, arm, p, (long)B, (const T*)q, (const T*)dq, (const T*)tg, (const T*)tv, (T*)u);
  return 0;
}

// dispatch over (arm kind, dtype): static built-in by name, or runtime table by joint count
template <class F>
int with_arm(const char* name, const abrk_arm_desc* d, int dtype, F&& f) {
#define STATIC_CASE(NM)                                                                     \
  if (name && !strcmp(name, #NM)) {                                                         \
    StaticArm<Tab_##NM> a;                                                                  \
    return dtype == 0 ? f(a, double(0), Tab_##NM::N) : f(a, float(0), Tab_##NM::N);         \
  }
#if HOSTSIM_STATIC
  STATIC_CASE(ur5) STATIC_CASE(jaco2) STATIC_CASE(twojoint) STATIC_CASE(threejoint) STATIC_CASE(onejoint)
#endif
#undef STATIC_CASE
  if (!d) return -4;
#define RT_CASE(NN)                                                           \
  if (d->n_joints == NN) {                                                    \
    if (dtype == 0) {                                                         \
      RtArm<NN, double> a;                                                    \
      rt_fill<NN, double>(d, &a);                                             \
      return f(a, double(0), NN);                                             \
    } else {                                                                  \
      RtArm<NN, float> a;                                                     \
      rt_fill<NN, float>(d, &a);                                              \
      return f(a, float(0), NN);                                              \
    }                                                                         \
  }
#if HOSTSIM_RT_LO <= 1 && 1 <= HOSTSIM_RT_HI
  RT_CASE(1)
#endif
#if HOSTSIM_RT_LO <= 2 && 2 <= HOSTSIM_RT_HI
  RT_CASE(2)
#endif
#if HOSTSIM_RT_LO <= 3 && 3 <= HOSTSIM_RT_HI
  RT_CASE(3)
#endif
#if HOSTSIM_RT_LO <= 4 && 4 <= HOSTSIM_RT_HI
  RT_CASE(4)
#endif
#if HOSTSIM_RT_LO <= 5 && 5 <= HOSTSIM_RT_HI
  RT_CASE(5)
#endif
#if HOSTSIM_RT_LO <= 6 && 6 <= HOSTSIM_RT_HI
  RT_CASE(6)
#endif
#if HOSTSIM_RT_LO <= 7 && 7 <= HOSTSIM_RT_HI
  RT_CASE(7)
#endif
#undef RT_CASE
  return -4;
}
}  // namespace

extern "C" int hostsim_dynamics(const char* builtin, const abrk_arm_desc* d, int dtype, int64_t B, const void* q,
                                const void* dq, int frame, const double* off, uint32_t want,
                                const abrk_dyn_out* out) {
  void* const* outs = reinterpret_cast<void* const*>(out);
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    return run_dyn<A, T>(a, n, B, q, dq, frame, off, want, outs);
  });
}
extern "C" int hostsim_osc(const char* builtin, const abrk_arm_desc* d, int dtype, const abrk_osc_params* P,
                           int64_t B, const void* q, const void* dq, const void* tg, const void* tv, void* ie,
                           const void* une, void* u, void* ts) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    return run_osc<A, T>(a, n, P, B, q, dq, tg, tv, ie, une, u, ts);
  });
}
extern "C" int hostsim_osc_handover(const char* builtin, const abrk_arm_desc* d, int dtype, const abrk_osc_params* P,
                                    int64_t B, const void* q, const void* dq, const void* tg, const void* tv, void* ie,
                                    const void* une, void* u, void* ts, int64_t* n_deferred) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    return run_osc_handover<A, T>(a, n, P, B, q, dq, tg, tv, ie, une, u, ts, n_deferred);
  });
}
extern "C" int hostsim_osc_full(const char* builtin, const abrk_arm_desc* d, int dtype, const abrk_osc_params* P,
                                int64_t B, const void* q, const void* dq, const void* tg, const void* tv, void* ie,
                                const void* une, void* u, void* ts, uint32_t want, const abrk_dyn_out* out) {
  void* const* outs = reinterpret_cast<void* const*>(out);
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    return run_osc_full<A, T>(a, n, P, B, q, dq, tg, tv, ie, une, u, ts, want, outs);
  });
}
extern "C" int hostsim_sliding(const char* builtin, const abrk_arm_desc* d, int dtype,
                               const abrk_sliding_params* P, int64_t B, const void* q, const void* dq,
                               const void* tg, const void* tv, const void* ta, void* u, void* s) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    return run_sliding<A, T>(a, n, P, B, q, dq, tg, tv, ta, u, s);
  });
}
extern "C" int hostsim_joint(const char* builtin, const abrk_arm_desc* d, int dtype, const abrk_null_ctrl* c,
                             int grav, int64_t B, const void* q, const void* dq, const void* tg, const void* tv,
                             void* u) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    return run_joint<A, T>(a, n, c, grav, B, q, dq, tg, tv, u);
  });
}

extern "C" int hostsim_floating(const char* builtin, const abrk_arm_desc* d, int dtype, int dynamic, int task_space,
                                int64_t B, const void* q, const void* dq, void* u, int acc) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    for (long b = 0; b < B; b++) floating_body<A, T>(b, a, dynamic, task_space, (const T*)q, (const T*)dq, (T*)u, acc);
    return 0;
  });
}
// plain != 0: the one-pass row program everywhere; else what libabrk dispatches - the split program (phase A, one heavy
// pair at a time, finish) on orthogonal chains of three joints and more with at most 64 heavy slots
extern "C" int hostsim_obstacles(const char* builtin, const abrk_arm_desc* d, int dtype,
                                 const abrk_obstacles_params* P, int64_t B, const void* q, void* u, int acc, int plain) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    ObsP<T> p = make_obsp<T>(*P);
    const bool split = !plain && A::kOrtho && A::N >= 3 && p.n * (A::N - 2) <= 64;
    for (long b = 0; b < B; b++) {
      if (split) obstacles_split_body<A, T>(b, a, p, (const T*)q, (T*)u, acc);
      else obstacles_body<A, T>(b, a, p, (const T*)q, (T*)u, acc);
    }
    return 0;
  });
}

#if HOSTSIM_LAW
extern "C" int hostsim_limits(int n, int dtype, const abrk_limits_params* P, int64_t B, const void* q, void* u,
                              int acc) {
#define LIM_CASE(NN)                                                                                      \
  if (n == NN) {                                                                                          \
    if (dtype == 0) {                                                                                     \
      LimitsP<double> p = make_limitsp<double>(*P);                                                       \
      for (long b = 0; b < B; b++) limits_body<NN, double>(b, p, (const double*)q, (double*)u, acc);      \
    } else {                                                                                              \
      LimitsP<float> p = make_limitsp<float>(*P);                                                         \
      for (long b = 0; b < B; b++) limits_body<NN, float>(b, p, (const float*)q, (float*)u, acc);         \
    }                                                                                                     \
    return 0;                                                                                             \
  }
  LIM_CASE(1) LIM_CASE(2) LIM_CASE(3) LIM_CASE(4) LIM_CASE(5) LIM_CASE(6) LIM_CASE(7)
#undef LIM_CASE
  return -1;
}

extern "C" int hostsim_osc_mx(int n, int k, int dtype, int64_t B, const void* M, const void* J, double thr, void* Mx,
                              void* Minv) {
#define MX_CASE(NN)                                                                                               \
  if (n == NN) {                                                                                                  \
    for (long b = 0; b < B; b++) {                                                                                \
      if (dtype == 0)                                                                                             \
        mx_body<NN, double>(b, k, thr, (const double*)M, (const double*)J, (double*)Mx, (double*)Minv);           \
      else                                                                                                        \
        mx_body<NN, float>(b, k, float(thr), (const float*)M, (const float*)J, (float*)Mx, (float*)Minv);         \
    }                                                                                                             \
    return 0;                                                                                                     \
  }
  MX_CASE(1) MX_CASE(2) MX_CASE(3) MX_CASE(4) MX_CASE(5) MX_CASE(6) MX_CASE(7)
#undef MX_CASE
  return -1;
}
// the direct symmetric 3x3 eigen-solver on its own (A: [B,3,3] symmetric; lam [B,3], V [B,3,3] columns = eigenvectors)
// the 6 x 6 eigen-solvers behind the six-row law's truncating pinv: method 0 = cyclic Jacobi, 1 = Householder + QL
extern "C" int hostsim_sym6_eig(int method, int64_t B, const double* A, double* lam, double* V) {
  for (long b = 0; b < B; b++) {
    double S[21], Vv[6][6], l[6];
    for (int r = 0; r < 6; r++)
      for (int c = 0; c <= r; c++) S[tri(r, c)] = A[b * 36 + r * 6 + c];
    if (method == 0) jacobi_eig<6, double>(S, Vv, l);
    else ql_eig<6, double>(S, Vv, l);
    for (int r = 0; r < 6; r++) {
      lam[b * 6 + r] = l[r];
      for (int c = 0; c < 6; c++) V[b * 36 + r * 6 + c] = Vv[r][c];
    }
  }
  return 0;
}
// the truncating pseudo-inverse + tail of the six-row law on its own (abrk_ctrl.h osc6_tail: early-exit QL + tridiagonal
// solve): A [B,6,6] symmetric, G [B,8,6] (columns of [J | u_task | J v], six joints), b [B,12] (b1, b2) -> out [B,12]
// (u, ts), info [B,2] (index of the last eigenvalue the iteration isolated: -1 none ... 5 all; the cut-off it used)
extern "C" int hostsim_osc6_tail(int64_t B, const double* A, const double* G, const double* b, double* out, double* info) {
  for (long k = 0; k < B; k++) {
    double S[21], Gm[8][6], b1[6], b2[6], u[6], ts[6];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j <= i; j++) S[tri(i, j)] = A[k * 36 + i * 6 + j];
    for (int c = 0; c < 8; c++)
      for (int r = 0; r < 6; r++) Gm[c][r] = G[k * 48 + c * 6 + r];
    for (int i = 0; i < 6; i++) {
      b1[i] = b[k * 12 + i];
      b2[i] = b[k * 12 + 6 + i];
    }
    {
      double G2[1][6] = {{0, 0, 0, 0, 0, 0}}, lam[6];
      QlTail<6, double> t;
      t.rcond = 1e-4;
      ql_core<6, double, 1, false, false, true>(S, G2, lam, &t);
      info[k * 2] = t.lexit;
      info[k * 2 + 1] = t.cut;
    }
    osc6_tail<6, double>(S, Gm, b1, b2, true, u, ts);
    for (int i = 0; i < 6; i++) {
      out[k * 12 + i] = u[i];
      out[k * 12 + 6 + i] = ts[i];
    }
  }
  return 0;
}
// the cofactor inverse of the x,y,z / x,y law (abrk_ctrl.h `spd_inverse_small`): A [B,K,K] symmetric -> inv [B,K,K],
// det [B], ok [B]
extern "C" int hostsim_spd_inverse_small(int K, int64_t B, const double* A, double* inv, double* det, int* ok) {
  for (long b = 0; b < B; b++) {
    if (K == 3) {
      double S[6], I[6], d;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c <= r; c++) S[tri(r, c)] = A[b * 9 + r * 3 + c];
      ok[b] = spd_inverse_small<3, double>(S, I, d) ? 1 : 0;
      det[b] = d;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) inv[b * 9 + r * 3 + c] = I[tri(r > c ? r : c, r > c ? c : r)];
    } else if (K == 2) {
      double S[3], I[3], d;
      for (int r = 0; r < 2; r++)
        for (int c = 0; c <= r; c++) S[tri(r, c)] = A[b * 4 + r * 2 + c];
      ok[b] = spd_inverse_small<2, double>(S, I, d) ? 1 : 0;
      det[b] = d;
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) inv[b * 4 + r * 2 + c] = I[tri(r > c ? r : c, r > c ? c : r)];
    } else {
      return -1;
    }
  }
  return 0;
}
extern "C" int hostsim_sym3_eig(int dtype, int64_t B, const void* A, void* lam, void* V) {
  auto run = [&](auto tag) {
    using T = decltype(tag);
    const T* a = (const T*)A;
    for (long b = 0; b < B; b++) {
      T S[6], Vv[3][3], l[3];
      for (int r = 0; r < 3; r++)
        for (int c = 0; c <= r; c++) S[tri(r, c)] = a[b * 9 + r * 3 + c];
      sym3_eig<T>(S, Vv, l);
      for (int r = 0; r < 3; r++) {
        ((T*)lam)[b * 3 + r] = l[r];
        for (int c = 0; c < 3; c++) ((T*)V)[b * 9 + r * 3 + c] = Vv[r][c];
      }
    }
  };
  if (dtype == 0) run(double{});
  else run(float{});
  return 0;
}
extern "C" int hostsim_velocity_limiting(int dtype, const abrk_osc_params* P, int64_t B, const void* in, void* out) {
  for (long b = 0; b < B; b++) {
    if (dtype == 0)
      velocity_limiting_body<double>(b, P->kp, P->ko, P->kv, P->vmax[0], P->vmax[1], (const double*)in, (double*)out);
    else
      velocity_limiting_body<float>(b, float(P->kp), float(P->ko), float(P->kv), float(P->vmax[0]), float(P->vmax[1]),
                                    (const float*)in, (float*)out);
  }
  return 0;
}
extern "C" int hostsim_orientation_forces(int alg, int dtype, int64_t B, const void* R, const void* abg, void* out) {
  for (long b = 0; b < B; b++) {
    if (dtype == 0)
      orientation_forces_body<double>(b, alg, (const double*)R, (const double*)abg, (double*)out);
    else
      orientation_forces_body<float>(b, alg, (const float*)R, (const float*)abg, (float*)out);
  }
  return 0;
}

extern "C" int hostsim_osc_law(int n, int dtype, const abrk_osc_params* P, int64_t B, const void* J, const void* M,
                               const void* g, const void* c, const void* xyz, const void* R, const void* q,
                               const void* dq, const void* tg, const void* tv, void* ie, const void* une, void* u,
                               void* ts) {
  if (P->ki == 0) ie = nullptr;
  if (!P->use_g) g = nullptr;
  if (!P->use_C) c = nullptr;
#define LAW_CASE(NN)                                                                                              \
  if (n == NN) {                                                                                                  \
    if (dtype == 0) {                                                                                             \
      OscP<double> p = make_oscp<double>(*P, n);                                                                  \
      for (long b = 0; b < B; b++)                                                                                \
        osc_law_body<NN, double>(b, p, (long)B, (const double*)J, (const double*)M, (const double*)g,             \
                                 (const double*)c, (const double*)xyz, (const double*)R, (const double*)q,        \
                                 (const double*)dq, (const double*)tg, (const double*)tv, (double*)ie,            \
                                 (const double*)une, (double*)u, (double*)ts);                                    \
    } else {                                                                                                      \
      OscP<float> p = make_oscp<float>(*P, n);                                                                    \
      for (long b = 0; b < B; b++)                                                                                \
        osc_law_body<NN, float>(b, p, (long)B, (const float*)J, (const float*)M, (const float*)g, (const float*)c, \
                                (const float*)xyz, (const float*)R, (const float*)q, (const float*)dq,            \
                                (const float*)tg, (const float*)tv, (float*)ie, (const float*)une, (float*)u,     \
                                (float*)ts);                                                                      \
    }                                                                                                             \
    return 0;                                                                                                     \
  }
  LAW_CASE(1) LAW_CASE(2) LAW_CASE(3) LAW_CASE(4) LAW_CASE(5) LAW_CASE(6) LAW_CASE(7)
#undef LAW_CASE
  return -1;
}

#endif

extern "C" int hostsim_rollout(const char* builtin, const abrk_arm_desc* d, int dtype, const abrk_osc_params* P,
                               const abrk_twolink_plant* plant, int64_t B, int n_steps, int every, void* q, void* dq,
                               const void* tg, void* qt, void* dqt, void* ut) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    if constexpr (A::N == 2) {
      OscP<T> p = make_oscp<T>(*P, n);
      TwoLinkP<T> k{T(plant->K1), T(plant->K2), T(plant->K3), T(plant->K4), T(plant->dt)};
      for (long b = 0; b < B; b++) {
        const bool xy = osc_fast_rows(*P, n, false) == 2;
#define ROLL(UC, KM)                                                                                            \
  rollout_body<A, T, UC, KM>(b, a, p, k, (long)B, n_steps, every, (T*)q, (T*)dq, (const T*)tg, (T*)nullptr, (T*)qt, \
                             (T*)dqt, (T*)ut)
        if (xy) {
          if (P->use_C) ROLL(true, 2); else ROLL(false, 2);
        } else {
          if (P->use_C) ROLL(true, 6); else ROLL(false, 6);
        }
#undef ROLL
      }
      return 0;
    } else {
      return -1;
    }
  });
}

extern "C" int hostsim_ik(const char* builtin, const abrk_arm_desc* d, int dtype, const abrk_ik_params* P, int64_t B,
                          const void* q, const void* tg, void* pp, void* vp) {
  return with_arm(builtin, d, dtype, [&](const auto& a, auto t, int n) {
    using A = std::decay_t<decltype(a)>;
    using T = decltype(t);
    IkP<T> p{T(P->max_dx * P->dt), T(P->max_dr * P->dt), T(P->max_dq * P->dt), P->n_timesteps, P->method};
    for (long b = 0; b < B; b++) ik_body<A, T>(b, a, p, (long)B, (const T*)q, (const T*)tg, (T*)pp, (T*)vp);
    return 0;
  });
}
