// abrk_rows.h - the complete per-row programs (load a row, evaluate, store): what one GPU
// lane executes.  Kept separate from the __global__ wrappers so that tests/hostsim can
// compile the very same row code for the host and check the kernel arithmetic against the
// oracle in a container without a GPU (test aid only - the product never runs it on a CPU).
#pragma once
#include "abrk_ctrl.h"

namespace abrk {

template <class T>
struct DynOutP {
  T *Tx, *J, *M, *g, *C, *dJ, *R, *Tm, *Tinv, *quat;
};

enum {
  W_TX = 1u << 0, W_J = 1u << 1, W_M = 1u << 2, W_G = 1u << 3, W_C = 1u << 4,
  W_DJ = 1u << 5, W_R = 1u << 6, W_T = 1u << 7, W_TINV = 1u << 8, W_QUAT = 1u << 9
};

template <int N, class T>
ABRK_INL void load_row(const T* __restrict__ base, long b, T (&v)[N]) {
  const T* p = base + b * N;
  sfor<N>([&](auto i) ABRK_LAMBDA { v[i()] = p[i()]; });
}
template <int N, class T>
ABRK_INL void store_row(T* __restrict__ base, long b, const T (&v)[N]) {
  T* p = base + b * N;
  sfor<N>([&](auto i) ABRK_LAMBDA { p[i()] = v[i()]; });
}

// Output policies of the dynamics program.  DirectStore: each lane writes its own row (host
// check build; rows of 24..288 bytes at a 24..288-byte lane stride are badly coalesced).
// The GPU kernel uses LdsStore (abrk_kernels.h): rows are transposed through LDS so that every
// store instruction of the wavefront writes 512 contiguous bytes.
template <class T>
struct DirectStore {
  template <int R>
  ABRK_INL void put(T* __restrict__ out, long b, bool active, const T (&v)[R]) {
    if (active) store_row<R>(out, b, v);
  }
};

// ---- robot_config.{Tx,J,M,g,C,dJ,R,T,T_inv,quaternion} for B states (base_config.py:210-415)
// `b` may be past the end for the padding lanes of the last wavefront (active = false): they
// evaluate a clamped row and only take part in the cooperative stores.
template <class A, class T, bool WITH_DQ, class St>
ABRK_INL void dyn_body(long b, bool active, St& st, const A& arm, int frame, int m, T ox, T oy, T oz, unsigned want,
                       long B, const T* __restrict__ qg, const T* __restrict__ dqg, const DynOutP<T>& out) {
  constexpr int N = A::N;
  const long bl = active ? b : B - 1;
  T q[N], dq[N];
  load_row<N>(qg, bl, q);
  if constexpr (WITH_DQ) load_row<N>(dqg, bl, dq);
  else sfor<N>([&](auto i) ABRK_LAMBDA { dq[i()] = T(0); });
  Joints<A, T> jt;
  Dyn<A, T, WITH_DQ ? CMODE_MAT : CMODE_NONE> d;
  T XR[9], xo[3];
  FrameCap<T> cap;
  cap.frame = frame;
  sfor<9>([&](auto e) ABRK_LAMBDA { cap.R[e()] = T(0); });
  sfor<3>([&](auto r) ABRK_LAMBDA { cap.o[r()] = T(0); });
  kin_dyn(arm, q, dq, jt, d, XR, xo, cap);
  T p[3];
  sfor<3>([&](auto r) ABRK_LAMBDA {
    p[r()] = cap.o[r()] + cap.R[r() * 3] * ox + cap.R[r() * 3 + 1] * oy + cap.R[r() * 3 + 2] * oz;
  });
  if (want & W_TX) st.template put<3>(out.Tx, b, active, p);
  if (want & (W_J | W_DJ)) {
    T Jv[N][3], Jw[N][3];
    jacobian(jt, p, m, Jv, Jw);
    if (want & W_J) {
      T row[6 * N];
      sfor<3>([&](auto r) ABRK_LAMBDA {
        sfor<N>([&](auto i) ABRK_LAMBDA {
          row[r() * N + i()] = Jv[i()][r()];
          row[(3 + r()) * N + i()] = Jw[i()][r()];
        });
      });
      st.template put<6 * N>(out.J, b, active, row);
    }
    if constexpr (WITH_DQ) {
      if (want & W_DJ) {
        T dJv[N][3], dJw[N][3];
        jacobian_dot(jt, dq, Jv, m, dJv, dJw);
        T row[6 * N];
        sfor<3>([&](auto r) ABRK_LAMBDA {
          sfor<N>([&](auto i) ABRK_LAMBDA {
            row[r() * N + i()] = dJv[i()][r()];
            row[(3 + r()) * N + i()] = dJw[i()][r()];
          });
        });
        st.template put<6 * N>(out.dJ, b, active, row);
      }
    }
  }
  if (want & W_M) {
    T row[N * N];
    sfor<N>([&](auto i) ABRK_LAMBDA { sfor<N>([&](auto j) ABRK_LAMBDA { row[i() * N + j()] = d.Ms[tri(i(), j())]; }); });
    st.template put<N * N>(out.M, b, active, row);
  }
  if (want & W_G) {
    T row[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { row[i()] = T(-9.81) * d.gz[i()]; });
    st.template put<N>(out.g, b, active, row);
  }
  if constexpr (WITH_DQ) {
    if (want & W_C) st.template put<N * N>(out.C, b, active, d.Cm);
  }
  if (want & W_R) st.template put<9>(out.R, b, active, cap.R);
  if (want & W_T) {
    T row[16];
    sfor<3>([&](auto r) ABRK_LAMBDA {
      sfor<3>([&](auto c) ABRK_LAMBDA { row[r() * 4 + c()] = cap.R[r() * 3 + c()]; });
      row[r() * 4 + 3] = p[r()];  // T * [x,1] column: with x = 0 this is the frame origin
      row[12 + r()] = T(0);
    });
    row[15] = T(1);
    st.template put<16>(out.Tm, b, active, row);
  }
  if (want & W_TINV) {  // base_config.py:791-837: [R^T | -R^T t]
    T row[16];
    sfor<3>([&](auto r) ABRK_LAMBDA {
      sfor<3>([&](auto c) ABRK_LAMBDA { row[r() * 4 + c()] = cap.R[c() * 3 + r()]; });
      row[r() * 4 + 3] =
          -(cap.R[0 * 3 + r()] * cap.o[0] + cap.R[1 * 3 + r()] * cap.o[1] + cap.R[2 * 3 + r()] * cap.o[2]);
      row[12 + r()] = T(0);
    });
    row[15] = T(1);
    st.template put<16>(out.Tinv, b, active, row);
  }
  if (want & W_QUAT) {
    T qq[4];
    quat_from_R(cap.R, qq);
    st.template put<4>(out.quat, b, active, qq);
  }
}

// ---- OSC.generate for B states (osc.py:217-320)
// `scr`: per-lane scratch of the Coriolis recursion (RegScratch, or the wavefront's LDS slab on the GPU)
// the inputs a kernel may request AHEAD of osc_body (before its sin/cos table fill: the x,y,z kernels - one memory round
// trip for table and inputs together instead of two in a row, which a lone wavefront per SIMD cannot hide)
template <class T, int N>
struct OscPre {
  T q[N], dq[N], tgt[6];
};
// which of them osc_body would itself request up front (the rest it asks for after the kinematics)
template <int KM, int FEAT>
constexpr bool osc_pre_all() { return FEAT < 2 && KM <= 3; }
template <class A, class T, int KM, bool USE_C, int FEAT>
ABRK_INL void osc_prefetch(long b, const T* __restrict__ qg, const T* __restrict__ dqg, const T* __restrict__ tg,
                           OscPre<T, A::N>& pre) {
  load_row<A::N>(qg, b, pre.q);
  if constexpr (USE_C || osc_pre_all<KM, FEAT>()) load_row<A::N>(dqg, b, pre.dq);
  if constexpr (osc_pre_all<KM, FEAT>()) load_row<6>(tg, b, pre.tgt);
}
template <class A, class T, int KM, bool USE_C, int FEAT, class Scr>
ABRK_INL void osc_body(long b, const A& arm, const OscP<T>& P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
           const T* __restrict__ tg, const T* __restrict__ tvg, T* __restrict__ ierrg,
           const T* __restrict__ uneg, T* __restrict__ ug, T* __restrict__ tsg, Scr& scr,
           const OscPre<T, A::N>* pre = nullptr) {
  constexpr int N = A::N;
  T q[N], dq[N], tgt[6], tv[6], ierr[6], une[N], u[N], ts[N];
  const bool tv_given = FEAT >= 2 && tvg != nullptr, have_ierr = FEAT >= 2 && ierrg != nullptr,
             have_ext = FEAT >= 2 && uneg != nullptr;
  if (pre) sfor<N>([&](auto i) ABRK_LAMBDA { q[i()] = pre->q[i()]; });
  else load_row<N>(qg, b, q);
  // FEAT=false: every input is requested up front (one HBM round trip; 36 extra registers still
  // fit the two-waves-per-SIMD budget).  FEAT=true: the optional inputs are requested after the
  // kinematics to keep that kernel's register peak down.
  // (the six-row kernels ask late as well: their law, not the kinematics, is the register peak)
  // (the six-row FIRST pass of orthogonal chains asks early too since round 4: its law was restructured to 224 - 246
  //  registers, and at two waves per SIMD one memory round trip fewer per wavefront is worth 2.7 % at 8 M rows - 740 / 746 /
  //  744 us against 760 / 765 / 767 us, same box; the one-wave six-row kernels keep asking late)
  // (round 6: ... and so does the one-wave first pass of built-in / compiled GENERAL chains - 304 + 48 registers on Jaco2,
  //  room for the 36 more: same box, Jaco2 five task rows at 8 M rows 925.6 -> 877.2 us, 1 M rows 139.7 -> 137.8 us, the
  //  4096-row step unchanged, same bits (profiles/round6/ab/jaco2_inputs_early/).  Requesting the NEXT row's inputs at the
  //  law's tail on top of that - software pipelining of the persistent loop - was built and measured too: 360 + 104
  //  registers, 928 us: the accumulator-register traffic eats what the hidden round trip returns.)
  constexpr bool EARLY = FEAT < 2 && (KM <= 3 || (std::remove_reference<Scr>::type::kDeferOnly && A::kStatic));
  // (requesting the target after the kinematics in the use_C kernels was measured unnecessary once the link wrenches
  //  of the Coriolis recursion live in LDS: 240 VGPRs either way, one memory round trip fewer)
  constexpr bool EARLY_T = EARLY;
  if (pre) {
    static_assert(KM > 3 || EARLY == osc_pre_all<KM, FEAT>(), "osc_prefetch requests what osc_body would");
    if constexpr (USE_C || EARLY) sfor<N>([&](auto i) ABRK_LAMBDA { dq[i()] = pre->dq[i()]; });
    if constexpr (EARLY_T) sfor<6>([&](auto r) ABRK_LAMBDA { tgt[r()] = pre->tgt[r()]; });
  } else {
    if constexpr (USE_C || EARLY) load_row<N>(dqg, b, dq);
    if constexpr (EARLY_T) load_row<6>(tg, b, tgt);
  }
  auto late = [&]() ABRK_LAMBDA {
    if constexpr (!USE_C && !EARLY) load_row<N>(dqg, b, dq);
    if constexpr (!EARLY_T) load_row<6>(tg, b, tgt);
    if (tv_given) load_row<6>(tvg, b, tv);
    else sfor<6>([&](auto r) ABRK_LAMBDA { tv[r()] = T(0); });
    if (have_ierr) load_row<6>(ierrg, b, ierr);
    else sfor<6>([&](auto r) ABRK_LAMBDA { ierr[r()] = T(0); });
    if (have_ext) load_row<N>(uneg, b, une);
    else sfor<N>([&](auto i) ABRK_LAMBDA { une[i()] = T(0); });
  };
  ABRK_STAMP(scr, 2, true);  // (timeline build) every input requested above has landed
  osc_row<A, T, KM, USE_C, FEAT>(arm, P, q, dq, tgt, tv_given, tv, have_ierr, ierr, have_ext, une, u, ts, late, scr);
  ABRK_STAMP(scr, 4, false);
  if (scr.deferred) {
    // parked for the second pass: u / the training signal are not written yet.  A handed-over row is finished there from
    // its record, without its inputs: the integral state it advanced is stored here
    if (scr.handed_over && have_ierr) store_row<6>(ierrg, b, ierr);
    if (scr.singular && P.status) *P.status = 1;
    return;
  }
  store_row<N>(ug, b, u);
  if (tsg) store_row<N>(tsg, b, ts);
  if (have_ierr) store_row<6>(ierrg, b, ierr);
  if (scr.singular && P.status) *P.status = 1;  // ABRK_ESINGULAR (abrk_ctrl.h flag_singular)
}
template <class A, class T, int KM, bool USE_C, int FEAT>
ABRK_INL void osc_body(long b, const A& arm, const OscP<T>& P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
           const T* __restrict__ tg, const T* __restrict__ tvg, T* __restrict__ ierrg,
           const T* __restrict__ uneg, T* __restrict__ ug, T* __restrict__ tsg) {
  RegScratch<T, A::N> scr;
  osc_body<A, T, KM, USE_C, FEAT>(b, arm, P, B, qg, dqg, tg, tvg, ierrg, uneg, ug, tsg, scr);
}

// ---- OSC.generate + the robot_config outputs it consumed, one launch (SURVEY 8d "Mode F"): u [B,n] and, per `want`,
// Tx [B,3], J [B,6,n], M [B,n,n], g [B,n] of the controller's ref_frame / xyz_offset - what a caller of
// robot_config.J/M/g/Tx next to ctrlr.generate (osc.py:242-301 consumers, training-signal users) would otherwise
// pay a second forward kinematics for.  Cooperative stores (St = LdsStore on the GPU): padding lanes of the last
// wavefront evaluate a clamped row and only take part in the stores.
// VEL: `want` may also name C [B,n,n] (base_config.py:320-336) and dJ [B,6,n] (:225-247) of the same frame / offset -
// the two velocity-dependent robot_config functions (+288 B per UR5 row each); the dynamics pass then
// assembles the Christoffel matrix (osc_row MAT).
template <class A, class T, int KM, bool USE_C, int FEAT, bool VEL = false, class Scr, class St>
ABRK_INL void osc_full_body(long b, bool active, St& st, const A& arm, const OscP<T>& P, long B, const T* __restrict__ qg,
                            const T* __restrict__ dqg, const T* __restrict__ tg, const T* __restrict__ tvg,
                            T* __restrict__ ierrg, const T* __restrict__ uneg, T* __restrict__ ug,
                            T* __restrict__ tsg, unsigned want, const DynOutP<T>& out, Scr& scr) {
  constexpr int N = A::N;
  const long bl = active ? b : B - 1;
  T q[N], dq[N], tgt[6], tv[6], ierr[6], une[N], u[N], ts[N];
  const bool tv_given = FEAT >= 2 && tvg != nullptr, have_ierr = FEAT >= 2 && ierrg != nullptr,
             have_ext = FEAT >= 2 && uneg != nullptr;
  load_row<N>(qg, bl, q);
  load_row<N>(dqg, bl, dq);
  auto late = [&]() ABRK_LAMBDA {
    load_row<6>(tg, bl, tgt);
    if (tv_given) load_row<6>(tvg, bl, tv);
    else sfor<6>([&](auto r) ABRK_LAMBDA { tv[r()] = T(0); });
    if (have_ierr) load_row<6>(ierrg, bl, ierr);
    else sfor<6>([&](auto r) ABRK_LAMBDA { ierr[r()] = T(0); });
    if (have_ext) load_row<N>(uneg, bl, une);
    else sfor<N>([&](auto i) ABRK_LAMBDA { une[i()] = T(0); });
  };
  auto emit = [&](const T(&p)[3], const T(&Jv)[N][3], const T(&Jw)[N][3], const auto& d, const auto& jt, int m) ABRK_LAMBDA {
    const auto& Ms = d.Ms;
    const auto& gz = d.gz;
    if (want & W_TX) st.template put<3>(out.Tx, b, active, p);
    if (want & W_J) {
      T row[6 * N];
      sfor<3>([&](auto r) ABRK_LAMBDA {
        sfor<N>([&](auto i) ABRK_LAMBDA {
          row[r() * N + i()] = Jv[i()][r()];
          row[(3 + r()) * N + i()] = Jw[i()][r()];
        });
      });
      st.template put<6 * N>(out.J, b, active, row);
    }
    if (want & W_M) {
      T row[N * N];
      sfor<N>([&](auto i) ABRK_LAMBDA { sfor<N>([&](auto j) ABRK_LAMBDA { row[i() * N + j()] = Ms[tri(i(), j())]; }); });
      st.template put<N * N>(out.M, b, active, row);
    }
    if (want & W_G) {
      T row[N];
      sfor<N>([&](auto i) ABRK_LAMBDA { row[i()] = T(-9.81) * gz[i()]; });
      st.template put<N>(out.g, b, active, row);
    }
    if constexpr (VEL) {
      if (want & W_DJ) {
        T dJv[N][3], dJw[N][3];
        jacobian_dot(jt, dq, Jv, m, dJv, dJw);
        T row[6 * N];
        sfor<3>([&](auto r) ABRK_LAMBDA {
          sfor<N>([&](auto i) ABRK_LAMBDA {
            row[r() * N + i()] = dJv[i()][r()];
            row[(3 + r()) * N + i()] = dJw[i()][r()];
          });
        });
        st.template put<6 * N>(out.dJ, b, active, row);
      }
    }
  };
  auto emit_pre = [&](const auto& d) ABRK_LAMBDA {
    if constexpr (VEL) {
      if (want & W_C) st.template put<N * N>(out.C, b, active, d.Cm);
    }
  };
  osc_row<A, T, KM, USE_C, FEAT, VEL>(arm, P, q, dq, tgt, tv_given, tv, have_ierr, ierr, have_ext, une, u, ts, late, scr,
                                      emit, emit_pre);
  if (active) {
    store_row<N>(ug, b, u);
    if (tsg) store_row<N>(tsg, b, ts);
    if (have_ierr) store_row<6>(ierrg, b, ierr);
    if (scr.singular && P.status) *P.status = 1;  // ABRK_ESINGULAR (abrk_ctrl.h flag_singular)
  }
}

// ---- OSC control law on caller-supplied dynamics (osc.py:244-318): for robot_configs whose
// J / M / g / Tx / R come from elsewhere (the reference's duck-typed boundary, e.g. MujocoConfig,
// abr_control/arms/mujoco_config.py:201-451).  Inputs per row, row-major: J [6,N], M [N,N];
// optional g [N], Cdq [N] (= C(q,dq) dq), xyz [3], R [3,3], q [N] (RestingConfig only).
template <int N, class T>
ABRK_INL void osc_law_body(long b, const OscP<T>& P, long B, const T* __restrict__ Jg, const T* __restrict__ Mg,
                           const T* __restrict__ gg, const T* __restrict__ cg, const T* __restrict__ xg,
                           const T* __restrict__ Rg, const T* __restrict__ qg, const T* __restrict__ dqg,
                           const T* __restrict__ tg, const T* __restrict__ tvg, T* __restrict__ ierrg,
                           const T* __restrict__ uneg, T* __restrict__ ug, T* __restrict__ tsg) {
  T Jf[6 * N], Mf[N * N], Ms[N * (N + 1) / 2], gv[N], cv[N], Jv[N][3], Jw[N][3], p[3], RF[9];
  T q[N], dq[N], tgt[6], tv[6], ierr[6], une[N], u[N], ts[N];
  load_row<6 * N>(Jg, b, Jf);
  load_row<N * N>(Mg, b, Mf);
  sfor<N>([&](auto i) ABRK_LAMBDA {
    sfor<3>([&](auto r) ABRK_LAMBDA {
      Jv[i()][r()] = Jf[r() * N + i()];
      Jw[i()][r()] = Jf[(3 + r()) * N + i()];
    });
    sfor<i() + 1>([&](auto j) ABRK_LAMBDA { Ms[tri(i(), j())] = Mf[i() * N + j()]; });
  });
  auto opt = [&](const T* ptr, auto& dst, auto n, T fill) ABRK_LAMBDA {
    if (ptr) load_row<n()>(ptr, b, dst);
    else sfor<n()>([&](auto e) ABRK_LAMBDA { dst[e()] = fill; });
  };
  opt(gg, gv, ic<N>{}, T(0));
  opt(cg, cv, ic<N>{}, T(0));
  opt(xg, p, ic<3>{}, T(0));
  if (Rg) load_row<9>(Rg, b, RF);
  else sfor<9>([&](auto e) ABRK_LAMBDA { RF[e()] = (e() % 4 == 0) ? T(1) : T(0); });
  opt(qg, q, ic<N>{}, T(0));
  load_row<N>(dqg, b, dq);
  load_row<6>(tg, b, tgt);
  opt(tvg, tv, ic<6>{}, T(0));
  opt(ierrg, ierr, ic<6>{}, T(0));
  opt(uneg, une, ic<N>{}, T(0));
  RegScratch<T, N> rows;  // caller-supplied J: its six rows, masked (osc.py:244), in the law's row store
  sfor<6>([&](auto r) ABRK_LAMBDA {
    const bool on = P.dof[r()] != 0;
    T row[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { row[i()] = on ? ((r() < 3) ? Jv[i()][r() % 3] : Jw[i()][r() % 3]) : T(0); });
    rows.put_row(r, row);
  });
  osc_law6<N, T, true, 2>(P, Ms, gv, T(-1), cv, rows, p, RF, q, dq, tgt, tvg != nullptr, tv, ierrg != nullptr, ierr,
                          uneg != nullptr, une, u, ts);
  store_row<N>(ug, b, u);
  if (tsg) store_row<N>(tsg, b, ts);
  if (ierrg) store_row<6>(ierrg, b, ierr);
}

// ---- Sliding.generate for B states (sliding.py:34-99)
template <class A, class T, bool TAB = false>
ABRK_INL void sliding_body(long b, const A& arm, const SlidingP<T>& P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
               const T* __restrict__ tg, const T* __restrict__ tvg, const T* __restrict__ tag,
               T* __restrict__ ug, T* __restrict__ sg, const void* sctab = nullptr) {
  constexpr int N = A::N;
  constexpr int NT = N > 3 ? N : 3;
  T q[N], dq[N], tgt[NT], tv[NT], ta[NT], u[N], s[N];
  load_row<N>(qg, b, q);
  load_row<N>(dqg, b, dq);
  const int nt = P.cartesian ? 3 : N;
  sfor<NT>([&](auto i) ABRK_LAMBDA {
    bool in = i() < nt;
    tgt[i()] = in ? tg[b * nt + i()] : T(0);
    tv[i()] = (in && tvg) ? tvg[b * nt + i()] : T(0);
    ta[i()] = (in && tag) ? tag[b * nt + i()] : T(0);
  });
  sliding_row<A, T, TAB>(arm, P, q, dq, tgt, tv, ta, u, s, sctab);
  store_row<N>(ug, b, u);
  if (sg) store_row<N>(sg, b, s);
}

// ---- Joint / Damping / RestingConfig for B states
template <class A, class T>
ABRK_INL void joint_body(long b, const A& arm, const JointP<T>& P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
             const T* __restrict__ tg, const T* __restrict__ tvg, T* __restrict__ ug) {
  constexpr int N = A::N;
  T q[N], dq[N], tgt[N], tv[N], u[N];
  load_row<N>(qg, b, q);
  load_row<N>(dqg, b, dq);
  if (tg) load_row<N>(tg, b, tgt);
  else sfor<N>([&](auto i) ABRK_LAMBDA { tgt[i()] = T(0); });
  if (tvg) load_row<N>(tvg, b, tv);
  else sfor<N>([&](auto i) ABRK_LAMBDA { tv[i()] = T(0); });
  joint_row<A, T>(arm, P, q, dq, tgt, tv, u);
  store_row<N>(ug, b, u);
}

// ---- AvoidJointLimits / Floating / AvoidObstacles for B states; acc: u += instead of u =
template <int N, class T>
ABRK_INL void put_row(T* __restrict__ ug, long b, const T (&u)[N], int acc) {
  if (acc) {
    T old[N];
    load_row<N>(ug, b, old);
    sfor<N>([&](auto i) ABRK_LAMBDA { old[i()] += u[i()]; });
    store_row<N>(ug, b, old);
  } else {
    store_row<N>(ug, b, u);
  }
}

template <int N, class T>
ABRK_INL void limits_body(long b, const LimitsP<T>& P, const T* __restrict__ qg, T* __restrict__ ug, int acc) {
  T q[N], u[N];
  load_row<N>(qg, b, q);
  limits_row<N, T>(P, q, u);
  put_row<N>(ug, b, u, acc);
}

template <class A, class T>
ABRK_INL void floating_body(long b, const A& arm, int dynamic, int task_space, const T* __restrict__ qg,
                            const T* __restrict__ dqg, T* __restrict__ ug, int acc) {
  constexpr int N = A::N;
  T q[N], dq[N], u[N];
  load_row<N>(qg, b, q);
  if (dynamic) load_row<N>(dqg, b, dq);
  else sfor<N>([&](auto i) ABRK_LAMBDA { dq[i()] = T(0); });
  floating_row<A, T>(arm, dynamic, task_space, q, dq, u);
  put_row<N>(ug, b, u, acc);
}

template <class A, class T>
ABRK_INL void obstacles_body(long b, const A& arm, const ObsP<T>& P, const T* __restrict__ qg, T* __restrict__ ug,
                             int acc) {
  constexpr int N = A::N;
  T q[N], u[N];
  load_row<N>(qg, b, q);
  obstacles_row<A, T>(arm, P, q, u);
  put_row<N>(ug, b, u, acc);
}

// the same through the split row program (phase A, pairs, finish - abrk_ctrl.h obstacles_row_split): what the host check
// build runs where the GPU runs obstacles_lds_kernel
template <class A, class T>
ABRK_INL void obstacles_split_body(long b, const A& arm, const ObsP<T>& P, const T* __restrict__ qg, T* __restrict__ ug,
                                   int acc) {
  constexpr int N = A::N;
  T q[N], u[N];
  load_row<N>(qg, b, q);
  if constexpr (A::kOrtho && N >= 3) obstacles_row_split<A, T>(arm, P, q, u);
  else obstacles_row<A, T>(arm, P, q, u);
  put_row<N>(ug, b, u, acc);
}

// ---- closed loop: n_steps x { OSC.generate ; ArmSim._step } with the state kept in registers
// (examples/PyGame/force_osc_xy.py:57-78).  Two-joint arms only.
template <class A, class T, bool USE_C, int KM>
ABRK_INL void rollout_body(long b, const A& arm, const OscP<T>& P, const TwoLinkP<T>& K, long B, int n_steps, int every,
                           T* __restrict__ qg, T* __restrict__ dqg, const T* __restrict__ tg, T* __restrict__ ierrg,
                           T* __restrict__ qt, T* __restrict__ dqt, T* __restrict__ ut) {
  static_assert(A::N == 2, "the reference's Python plant is the two-link arm");
  T q[2], dq[2], tgt[6], tv[6], ierr[6], une[2], u[2], ts[2];
  load_row<2>(qg, b, q);
  load_row<2>(dqg, b, dq);
  load_row<6>(tg, b, tgt);
  const bool have_ierr = ierrg != nullptr;
  if (have_ierr) load_row<6>(ierrg, b, ierr);
  else sfor<6>([&](auto r) ABRK_LAMBDA { ierr[r()] = T(0); });
  sfor<6>([&](auto r) ABRK_LAMBDA { tv[r()] = T(0); });
  une[0] = une[1] = T(0);
  const int n_chk = every > 0 ? n_steps / every : 0;
  int chk = 0, until = every;
  RegScratch<T, 2> scr;
  bool singular = false;
  for (int t = 0; t < n_steps; t++) {
    osc_row<A, T, KM, USE_C, 2>(arm, P, q, dq, tgt, false, tv, have_ierr, ierr, false, une, u, ts, []() {}, scr);
    singular = singular || scr.singular;
    twolink_step(K, q, dq, u);
    if (every > 0 && --until == 0) {
      until = every;
      if (chk < n_chk) {
        const long o = (b * n_chk + chk) * 2;
        if (qt) { qt[o] = q[0]; qt[o + 1] = q[1]; }
        if (dqt) { dqt[o] = dq[0]; dqt[o + 1] = dq[1]; }
        if (ut) { ut[o] = u[0]; ut[o + 1] = u[1]; }
      }
      chk++;
    }
  }
  store_row<2>(qg, b, q);
  store_row<2>(dqg, b, dq);
  if (have_ierr) store_row<6>(ierrg, b, ierr);
  if (singular && P.status) *P.status = 1;
}

// ---- InverseKinematics.generate_path for B independent paths (inverse_kinematics.py:28-135)
template <class A, class T>
ABRK_INL void ik_body(long b, const A& arm, const IkP<T>& P, long B, const T* __restrict__ qg,
                      const T* __restrict__ tg, T* __restrict__ pp, T* __restrict__ vp) {
  constexpr int N = A::N;
  T q[N], tgt[6];
  load_row<N>(qg, b, q);
  load_row<6>(tg, b, tgt);
  ik_row<A, T>(arm, P, q, tgt, pp + b * (long)P.n_steps * N, vp + b * (long)P.n_steps * N);
}

template <class T>
ABRK_INL void twolink_step_body(long b, const TwoLinkP<T>& K, T* __restrict__ qg, T* __restrict__ dqg,
                                const T* __restrict__ ug) {
  T q[2], dq[2], u[2];
  load_row<2>(qg, b, q);
  load_row<2>(dqg, b, dq);
  load_row<2>(ug, b, u);
  twolink_step(K, q, dq, u);
  store_row<2>(qg, b, q);
  store_row<2>(dqg, b, dq);
}

// ---- OSC._Mx / ._velocity_limiting / ._calc_orientation_forces for B rows (osc.py:120-215)
template <int N, class T>
ABRK_INL void mx_body(long b, int k, T thr, const T* __restrict__ Mg, const T* __restrict__ Jg, T* __restrict__ Mxg,
                      T* __restrict__ Minvg) {
  T Ms[N * (N + 1) / 2], Jr[N][6], Mx[21], Minv[N * (N + 1) / 2];
  const T* Mp = Mg + b * (N * N);
  sfor<N>([&](auto i) ABRK_LAMBDA { sfor<i() + 1>([&](auto j) ABRK_LAMBDA { Ms[tri(i(), j())] = Mp[i() * N + j()]; }); });
  const T* Jp = Jg + b * (long)k * N;
  sfor<6>([&](auto r) ABRK_LAMBDA {
    sfor<N>([&](auto i) ABRK_LAMBDA { Jr[i()][r()] = (r() < k) ? Jp[r() * N + i()] : T(0); });
  });
  mx_row<N, T>(Ms, Jr, k, thr, Mx, Minv);
  T* Xp = Mxg + b * (long)k * k;
  sfor<6>([&](auto r) ABRK_LAMBDA {
    sfor<6>([&](auto c) ABRK_LAMBDA {
      if (r() < k && c() < k) Xp[r() * k + c()] = Mx[tri(r(), c())];
    });
  });
  if (Minvg) {
    T* Ip = Minvg + b * (N * N);
    sfor<N>([&](auto i) ABRK_LAMBDA { sfor<N>([&](auto j) ABRK_LAMBDA { Ip[i() * N + j()] = Minv[tri(i(), j())]; }); });
  }
}

template <class T>
ABRK_INL void velocity_limiting_body(long b, T kp, T ko, T kv, T vmax0, T vmax1, const T* __restrict__ ing,
                                     T* __restrict__ outg) {
  T ut[6];
  load_row<6>(ing, b, ut);
  velocity_limiting_row<T>(kp, ko, kv, vmax0, vmax1, ut);
  store_row<6>(outg, b, ut);
}

template <class T>
ABRK_INL void orientation_forces_body(long b, int alg, const T* __restrict__ Rg, const T* __restrict__ ag,
                                      T* __restrict__ outg) {
  T R[9], abg[3], uo[3];
  load_row<9>(Rg, b, R);
  load_row<3>(ag, b, abg);
  orientation_forces(alg, R, abg, uo);
  store_row<3>(outg, b, uo);
}

// ---- the six functions of abr_control/utils/transformations.py that the control path uses (:973 euler_matrix,
// :1096 quaternion_from_euler, :1192 quaternion_from_matrix, :1274 quaternion_multiply, :1293 quaternion_conjugate,
// :1632 unit_vector), one row each.  Euler axes: 'rxyz' (osc.py:164,178) and 'sxyz' (inverse_kinematics.py:73-82).
enum { TF_QUAT_FROM_EULER_RXYZ = 0, TF_QUAT_FROM_EULER_SXYZ = 1, TF_QUAT_FROM_MATRIX = 2, TF_QUAT_MULTIPLY = 3,
       TF_QUAT_CONJUGATE = 4, TF_UNIT_VECTOR4 = 5, TF_UNIT_VECTOR3 = 6, TF_EULER_MATRIX_RXYZ = 7 };
template <class T>
ABRK_INL void transformations_body(long b, int op, const T* __restrict__ ag, const T* __restrict__ bg,
                                   T* __restrict__ outg) {
  if (op == TF_QUAT_FROM_EULER_RXYZ || op == TF_QUAT_FROM_EULER_SXYZ) {
    T a[3], q[4];
    load_row<3>(ag, b, a);
    if (op == TF_QUAT_FROM_EULER_RXYZ) quat_from_euler_rxyz(a[0], a[1], a[2], q);
    else quat_from_euler_sxyz(a[0], a[1], a[2], q);
    store_row<4>(outg, b, q);
  } else if (op == TF_QUAT_FROM_MATRIX) {
    T R[9], q[4];
    load_row<9>(ag, b, R);
    quat_from_R(R, q);
    store_row<4>(outg, b, q);
  } else if (op == TF_QUAT_MULTIPLY) {
    T q1[4], q0[4], r[4];
    load_row<4>(ag, b, q1);
    load_row<4>(bg, b, q0);
    quat_mul(q1, q0, r);
    store_row<4>(outg, b, r);
  } else if (op == TF_QUAT_CONJUGATE) {
    T q[4];
    load_row<4>(ag, b, q);
    q[1] = -q[1];
    q[2] = -q[2];
    q[3] = -q[3];
    store_row<4>(outg, b, q);
  } else if (op == TF_UNIT_VECTOR4) {
    T v[4];
    load_row<4>(ag, b, v);
    const T inv = T(1) / Rm<T>::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    sfor<4>([&](auto i) ABRK_LAMBDA { v[i()] *= inv; });
    store_row<4>(outg, b, v);
  } else if (op == TF_UNIT_VECTOR3) {
    T v[3];
    load_row<3>(ag, b, v);
    const T inv = T(1) / Rm<T>::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    sfor<3>([&](auto i) ABRK_LAMBDA { v[i()] *= inv; });
    store_row<3>(outg, b, v);
  } else {
    T a[3], M[9];
    load_row<3>(ag, b, a);
    euler_matrix_rxyz(a[0], a[1], a[2], M);
    store_row<9>(outg, b, M);
  }
}

}  // namespace abrk
