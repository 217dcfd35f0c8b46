// abrk_ctrl.h - per-row control laws on top of abrk_device.h: OSC.generate
// (abr_control/controllers/osc.py:217-320), Sliding.generate (sliding.py:34-99),
// Joint / Damping / RestingConfig (joint.py:104-131, damping.py:21-32,
// resting_config.py:18-42) and the small fixed-size dense algebra they need, all unrolled
// into registers (no LAPACK on a GPU lane: inv -> Cholesky, pinv -> Jacobi).
#pragma once
#include "abrk_device.h"

namespace abrk {

// ---------------------------------------------------------------- device-side parameter blocks
// (filled on the host from abrk_osc_params etc., in the kernel's arithmetic type)
template <class T>
struct NullP {
  int kind;
  int mask[7];
  T kp, kv;
  T rest[7];
};

template <class T>
struct OscP {
  T kp, ko, kv, ki;
  T vmax0, vmax1;
  // velocity limiting (osc.py:110-115, 198-215), formed once on the host in T: vmax0 / kp * kv, vmax1 / ko * kv, kp / kv,
  // ko / kv.  (Computed in the kernels they were loop invariants: the six-row kernel's persistent loop kept the four
  // quotients - divisions of a branch most launches never take - in vector registers it does not have.)
  T sat_xyz, sat_abg, lamb_xyz, lamb_abg;
  T off[3];
  int use_vmax, use_g, alg;
  int dof[6];
  int pos_on, ori_on;
  int ref_frame, m_joints, has_off;
  int n_null;
  NullP<T> nul[4];
  // where a row whose joint-space inertia matrix has a non-positive Cholesky pivot raises the flag (the reference's
  // numpy.linalg.inv(M) raises LinAlgError there, osc.py:136): a host-visible word the host layer reads after the stream
  // is drained and reports as ABRK_ESINGULAR; nullptr = nobody listens
  int* status;
};
// (a plain store of 1 by however many rows: benign race, rare branch).  A non-finite M - NaN / Inf joint angles - is not
// "singular": numpy.linalg.inv returns NaNs for it without raising, and so do the kernels.  `minpiv`: the smallest
// Cholesky pivot of M that is a number (chol above): NaN pivots never enter it, so the test needs no look at M itself
// (summing M's entries in the cold branch kept all 21 of them alive across the factorisation: +6 .. 14 registers on the
// kernels that sit at the 256-register line - Jaco2's x,y,z law with secondary controllers spilled 44 B and lost 8 %).
// true on every lane of the wavefront if the condition holds on any of them (a scalar on the GPU)
ABRK_INL bool any_lane(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(c) != 0;
#else
  return c;
#endif
}
template <class T>
ABRK_INL void flag_singular(const OscP<T>& P, T minpiv) {
  if (!(minpiv > T(0)) && P.status) *P.status = 1;
}
template <class T>
struct SlidingP {
  T kd, lamb;
  T off[3];
  int cartesian, ref_frame, m_joints, has_off;
};

template <class T>
struct TwoLinkP {  // ArmSim constants (arms/twojoint/arm_sim.py:33-41)
  T K1, K2, K3, K4, dt;
};

template <class T>
struct IkP {  // InverseKinematics (inverse_kinematics.py:21-26, 65-71): step limits already scaled by dt
  T max_dx, max_dr, max_dq;
  int n_steps, method;
};

template <class T>
struct JointP {
  NullP<T> c;  // kind 0: Joint; 1: Damping; 2: RestingConfig
  int account_for_gravity;
};

template <class T>
struct LimitsP {  // AvoidJointLimits, as its constructor stores them (avoid_joint_limits.py:45-81)
  T mn[7], mx[7], mt[7];
  int cz[7], gr[7], nomin[7], nomax[7];
};

template <class T>
struct ObsP {  // AvoidObstacles (avoid_obstacles.py:26-36)
  int n;
  T threshold, gain, maximum;
  T obs[16][4];
};

// ---------------------------------------------------------------- small dense algebra
template <class T>
ABRK_INL T rcp(T x) {
  return Rm<T>::rcp(x);
}

// Cholesky of a symmetric K x K matrix in lower-triangular packed storage.
// L overwrites a copy; il[j] = 1/L_jj.  ok=false if a pivot is not positive.
// GUARD = false (fp64): the pivots are tested (the return value) but L and il are not sanitised - no two double selects
// per pivot (four v_cndmask: a double select costs as much as three FMAs on gfx950, tools/microbench/valu_rates.hip; 24
// issue slots of a 6 x 6 factor, and the compares go too where the caller ignores the result).  After a pivot <= 0, L and
// il hold non-finite values: for callers that read them only when ok (the task-space factor of the six-row law) or whose
// matrix is positive definite by construction (the joint-space inertia matrix of an arm with mass - a singular one gives
// non-finite torques where the reference raises LinAlgError).  The first failing pivot is a real number <= 0, so the
// flag does not depend on how comparisons treat the NaNs that follow it (-ffinite-math-only).
// minpiv (optional): receives the smallest pivot that is a NUMBER (fmin discards NaNs: v_min_f64 is IEEE minNum) - <= 0
// exactly when a real pivot is not positive, untouched by the NaN pivots of a non-finite matrix.
template <int K, class T, bool GUARD = true>
ABRK_INL bool chol(const T (&S)[K * (K + 1) / 2], T (&L)[K * (K + 1) / 2], T (&il)[K], T* minpiv = nullptr) {
  constexpr bool kGuard = GUARD || sizeof(T) < 8;  // fp32: a pivot of an ill-conditioned M can round to <= 0
  bool ok = true;
  sfor<K>([&](auto j) ABRK_LAMBDA {
    T dgn = S[tri(j(), j())];
    sfor<j()>([&](auto k) ABRK_LAMBDA { dgn = Rm<T>::fma(-L[tri(j(), k())], L[tri(j(), k())], dgn); });
    if (minpiv) *minpiv = Rm<T>::fmin(*minpiv, dgn);
    const bool pos = dgn > T(0);
    ok = ok && pos;
    const T dsafe = kGuard ? (pos ? dgn : T(1)) : dgn;
    const T rs = Rm<T>::rsqrt(dsafe);
    const T inv = kGuard ? (pos ? rs : T(0)) : rs;
    L[tri(j(), j())] = dsafe * rs;
    il[j()] = inv;
    sfor<K - 1 - j()>([&](auto ii) ABRK_LAMBDA {
      constexpr int i = j() + 1 + ii();
      T acc = S[tri(i, j())];
      sfor<j()>([&](auto k) ABRK_LAMBDA { acc = Rm<T>::fma(-L[tri(i, k())], L[tri(j(), k())], acc); });
      L[tri(i, j())] = acc * inv;
    });
  });
  return ok;
}
// Inverse and determinant of a symmetric 3 x 3 / 2 x 2 matrix by cofactors (packed lower: a00 a10 a11 a20 a21 a22):
// one reciprocal instead of the three reciprocal square roots of a Cholesky factor (v_rcp_f64 / v_rsq_f64 issue in 16
// cycles, four FMAs' worth), no pivot selects, and the inverse itself - which the null-space filter and the second
// certificate of the pinv branch want anyway - instead of two triangular solves: ~40 issue slots for the x,y,z law
// against ~95.  ok = positive definite (leading minors); the entries carry a relative error of cond(A) eps like the
// factor's while ONE eigenvalue is small, of cond^2 eps when two are: the caller gates on trace^K / det (osc_law).
// When !ok, Inv is the adjugate (finite, unused by the callers).
template <int K, class T>
ABRK_INL bool spd_inverse_small(const T (&A)[K * (K + 1) / 2], T (&Inv)[K * (K + 1) / 2], T& det) {
  static_assert(K == 2 || K == 3, "closed forms for 2 x 2 and 3 x 3");
  bool ok;
  if constexpr (K == 3) {
    const T a00 = A[0], a10 = A[1], a11 = A[2], a20 = A[3], a21 = A[4], a22 = A[5];
    Inv[0] = Rm<T>::fma(a11, a22, -(a21 * a21));
    Inv[1] = Rm<T>::fma(a21, a20, -(a10 * a22));
    Inv[2] = Rm<T>::fma(a00, a22, -(a20 * a20));
    Inv[3] = Rm<T>::fma(a10, a21, -(a20 * a11));
    Inv[4] = Rm<T>::fma(a10, a20, -(a00 * a21));
    Inv[5] = Rm<T>::fma(a00, a11, -(a10 * a10));
    det = Rm<T>::fma(a20, Inv[3], Rm<T>::fma(a10, Inv[1], a00 * Inv[0]));
    ok = a00 > T(0) && Inv[5] > T(0) && det > Rm<T>::tiny();
  } else {
    const T a00 = A[0], a10 = A[1], a11 = A[2];
    Inv[0] = a11;
    Inv[1] = -a10;
    Inv[2] = a00;
    det = Rm<T>::fma(a00, a11, -(a10 * a10));
    ok = a00 > T(0) && det > Rm<T>::tiny();
  }
  const T rd = ok ? rcp(ok ? det : T(1)) : T(1);
  sfor<K*(K + 1) / 2>([&](auto e) ABRK_LAMBDA { Inv[e()] *= rd; });
  return ok;
}
// x = L^-1 b
template <int K, class T>
ABRK_INL void chol_fwd(const T (&L)[K * (K + 1) / 2], const T (&il)[K], const T (&b)[K], T (&x)[K]) {
  sfor<K>([&](auto i) ABRK_LAMBDA {
    T acc = b[i()];
    sfor<i()>([&](auto k) ABRK_LAMBDA { acc = Rm<T>::fma(-L[tri(i(), k())], x[k()], acc); });
    x[i()] = acc * il[i()];
  });
}
// x = L^-T b
template <int K, class T>
ABRK_INL void chol_bwd(const T (&L)[K * (K + 1) / 2], const T (&il)[K], const T (&b)[K], T (&x)[K]) {
  sfor<K>([&](auto ir) ABRK_LAMBDA {
    constexpr int i = K - 1 - ir();
    T acc = b[i];
    sfor<K - 1 - i>([&](auto kk) ABRK_LAMBDA {
      constexpr int k = i + 1 + kk();
      acc = Rm<T>::fma(-L[tri(k, i)], x[k], acc);
    });
    x[i] = acc * il[i];
  });
}
// symmetric matvec from packed lower triangle
template <int K, class T>
ABRK_INL void symv(const T (&S)[K * (K + 1) / 2], const T (&v)[K], T (&o)[K]) {
  sfor<K>([&](auto i) ABRK_LAMBDA {
    T acc = S[tri(i(), 0)] * v[0];
    sfor<K - 1>([&](auto jj) ABRK_LAMBDA { acc = Rm<T>::fma(S[tri(i(), jj() + 1)], v[jj() + 1], acc); });
    o[i()] = acc;
  });
}
// L^-1 (lower, packed) of a Cholesky factor
template <int K, class T>
ABRK_INL void chol_factor_inverse(const T (&L)[K * (K + 1) / 2], const T (&il)[K], T (&Li)[K * (K + 1) / 2]) {
  sfor<K>([&](auto j) ABRK_LAMBDA {
    Li[tri(j(), j())] = il[j()];
    sfor<K - 1 - j()>([&](auto ii) ABRK_LAMBDA {
      constexpr int i = j() + 1 + ii();
      T acc = T(-0.0);
      sfor<i - j()>([&](auto kk) ABRK_LAMBDA {
        constexpr int k = j() + kk();
        acc -= L[tri(i, k)] * Li[tri(k, j())];
      });
      Li[tri(i, j())] = acc * il[i];
    });
  });
}
// inverse of an SPD matrix from its Cholesky factor: Sinv = L^-T L^-1 (packed)
template <int K, class T>
ABRK_INL void chol_inverse(const T (&L)[K * (K + 1) / 2], const T (&il)[K], T (&Sinv)[K * (K + 1) / 2]) {
  T Li[K * (K + 1) / 2];  // L^-1, lower
  sfor<K>([&](auto j) ABRK_LAMBDA {
    Li[tri(j(), j())] = il[j()];
    sfor<K - 1 - j()>([&](auto ii) ABRK_LAMBDA {
      constexpr int i = j() + 1 + ii();
      T acc = T(-0.0);
      sfor<i - j()>([&](auto kk) ABRK_LAMBDA {
        constexpr int k = j() + kk();
        acc -= L[tri(i, k)] * Li[tri(k, j())];
      });
      Li[tri(i, j())] = acc * il[i];
    });
  });
  sfor<K>([&](auto i) ABRK_LAMBDA {
    sfor<i() + 1>([&](auto j) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<K - i()>([&](auto kk) ABRK_LAMBDA {
        constexpr int k = i() + kk();
        acc += Li[tri(k, i())] * Li[tri(k, j())];
      });
      Sinv[tri(i(), j())] = acc;
    });
  });
}

// tan of the Jacobi rotation angle, t = sign(theta) / (|theta| + sqrt(theta^2 + 1)) with theta = num / den (den != 0),
// multiplied through by |den|: t = +-|den| / (|num| + hypot(num, den)) - one rsqrt and one rcp seed with their Newton
// corrections instead of two IEEE divisions and a square root (22 instead of ~50 vector instructions).
template <class T>
ABRK_INL T jacobi_tan(T num, T den) {
  const T an = Rm<T>::fabs(num), ad = Rm<T>::fabs(den);
  const T w = Rm<T>::fmax(Rm<T>::fma(num, num, den * den), Rm<T>::tiny());
  const T h = w * Rm<T>::rsqrt(w);
  const T t = ad * Rm<T>::rcp(an + h);
  return ((num >= T(0)) == (den >= T(0))) ? t : -t;
}

// Cyclic two-sided Jacobi eigen-decomposition of a symmetric K x K matrix (packed lower
// in, destroyed).  V columns = eigenvectors, lam = eigenvalues.  Stands in for the SVD
// inside numpy.linalg.pinv(Mx_inv) (osc.py:145) - for a symmetric matrix |eigenvalues|
// are the singular values.
template <int K, class T>
ABRK_INL void jacobi_eig(T (&S)[K * (K + 1) / 2], T (&V)[K][K], T (&lam)[K]) {
  sfor<K>([&](auto i) ABRK_LAMBDA { sfor<K>([&](auto j) ABRK_LAMBDA { V[i()][j()] = (i() == j()) ? T(1) : T(0); }); });
  // A rotation is applied while |a_pq| > eps sqrt(|a_pp a_qq|) (the relative criterion of Demmel & Veselic: what is
  // left below it perturbs every eigenvalue - small ones included - by a few eps of itself) or, where a diagonal entry
  // is itself rounding noise of a singular matrix, while |a_pq| > eps^2 |A|; a sweep that applies none ends the
  // iteration.  (Until round 3 every non-zero a_pq was rotated until the off-diagonal mass fell below 1e-18 |A|: one to
  // two sweeps of rotations by angles ~1e-17 that change nothing - on a lone lane, i.e. at the price of the whole
  // wavefront's time.)
  T dg0 = T(0);
  sfor<K>([&](auto p) ABRK_LAMBDA { dg0 += S[tri(p(), p())] * S[tri(p(), p())]; });
  const T floor2 = Rm<T>::eps() * Rm<T>::eps() * Rm<T>::eps() * Rm<T>::eps() * dg0;  // (eps^2 |A|)^2
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
    sfor<K>([&](auto qq) ABRK_LAMBDA {
      sfor<qq()>([&](auto pp) ABRK_LAMBDA {
        constexpr int p = pp(), q = qq();  // p < q
        T apq = S[tri(q, p)];
        const T a2 = apq * apq;
        if (a2 > Rm<T>::eps() * Rm<T>::eps() * Rm<T>::fabs(S[tri(p, p)] * S[tri(q, q)]) && a2 > floor2) {
          rotated = true;
          T t = jacobi_tan(S[tri(q, q)] - S[tri(p, p)], T(2) * apq);
          T c = Rm<T>::rsqrt(t * t + T(1));
          T s = t * c;
          sfor<K>([&](auto k) ABRK_LAMBDA {
            if constexpr (k() != p && k() != q) {
              T skp = S[tri(k(), p)], skq = S[tri(k(), q)];
              S[tri(k(), p)] = c * skp - s * skq;
              S[tri(k(), q)] = s * skp + c * skq;
            }
          });
          S[tri(p, p)] -= t * apq;
          S[tri(q, q)] += t * apq;
          S[tri(q, p)] = T(0);
          sfor<K>([&](auto k) ABRK_LAMBDA {
            T vkp = V[k()][p], vkq = V[k()][q];
            V[k()][p] = c * vkp - s * vkq;
            V[k()][q] = s * vkp + c * vkq;
          });
        }
      });
    });
    if (!rotated) break;
  }
  sfor<K>([&](auto i) ABRK_LAMBDA { lam[i()] = S[tri(i(), i())]; });
}

// Eigen-decomposition of a symmetric K x K matrix by Householder tridiagonalisation and the implicit QL iteration
// (the EISPACK tred2 / tql2 pair, restated with every index a compile-time constant: the matrices live in registers,
// and run-time indexing would send them to scratch memory).  Same interface as jacobi_eig; the eigenpairs come out in
// no particular order.  Why it is here: the truncating pinv of the six-row OSC law (osc.py:145) is rare per row (4.6 %
// of random UR5 states) but runs on a lone lane while its 63 neighbours wait, so its LATENCY is the config-sized
// step's critical path.  Cyclic Jacobi on a 6 x 6 matrix is ~6 sweeps x 15 rotations x ~75 instructions; this is ~400
// for the reduction and ~65 per QL rotation, ~2 iterations per eigenvalue on blocks that shrink as eigenvalues deflate.
// The predicated static unrolling executes every rotation slot of the active block (i >= m slots are skipped by the
// execution mask), so the cost is (K - 1 - l) slots per iteration for eigenvalue l.
// Accuracy: backward stable - eigenvalues to a few eps |A|, eigenvectors orthonormal to a few eps (jacobi_eig's
// RELATIVE accuracy on tiny eigenvalues is not needed by its callers: eigenvalues below rcond * max are dropped, the
// kept ones are at least 1e-4 |A|).
// a condition that holds on every lane of the wavefront alike (the wave-cooperative second pass: all lanes carry the
// same matrix): as a scalar, so that the branch on it is a scalar branch
ABRK_INL bool uni(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(c) != 0;  // v_cmp into a scalar pair, s_cmp, s_cbranch (a readfirstlane costs six more)
#else
  return c;
#endif
}
// the rotation slots I, I - 1, .. L of one implicit-QL pass for the uniform (one matrix per wavefront) form of ql_core.
// `m` is a scalar (the slots that exist are behind scalar compares, off the vector pipeline); a rotation radius that
// underflows (tql2's recovery path: never on matrices of sane scale) only raises `bad` - the caller then repeats the
// decomposition with the predicated form - so that no data-dependent branch sits on the recurrence.
template <int K, class T, int NR, int L, int I>
struct QlChain {
  static ABRK_INL void run(T (&d)[K], T (&e)[K], T (&V)[NR][K], int m, T& sn, T& cs, T& pp, T& g, bool& bad) {
#pragma clang fp contract(off)  // (see ql_core)
    if constexpr (I >= L) {
      if (I < m) {
        const T f = sn * e[I], b = cs * e[I];
        const T r2 = Rm<T>::fma(f, f, g * g);
        bad = bad || !(r2 > Rm<T>::tiny());
        const T r2g = Rm<T>::fmax(r2, Rm<T>::tiny());
        const T ir = Rm<T>::rsqrt(r2g), r = r2g * ir;
        e[I + 1] = r;
        sn = f * ir;
        cs = g * ir;
        g = d[I + 1] - pp;
        const T rr = Rm<T>::fma(d[I] - g, sn, T(2) * cs * b);
        pp = sn * rr;
        d[I + 1] = g + pp;
        g = Rm<T>::fma(cs, rr, -b);
        sfor<NR>([&](auto kk) ABRK_LAMBDA {
          const T fz = V[kk()][I + 1];
          V[kk()][I + 1] = Rm<T>::fma(sn, V[kk()][I], cs * fz);
          V[kk()][I] = Rm<T>::fma(cs, V[kk()][I], -(sn * fz));
        });
      }
      QlChain<K, T, NR, L, I - 1>::run(d, e, V, m, sn, cs, pp, g, bad);
    }
  }
};
// an int that is equal on every lane, as a scalar
ABRK_INL int uni_int(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}
// The solver proper: S = Z diag(lam) Z^T, applied to NR row vectors: V <- V Z.  IDENT: V comes in as the identity
// (NR = K; its row 0 is e_0 and no reflector touches it) - V goes out as Z, the eigenvectors in its columns.  Otherwise
// the NR rows are any vectors x^T and go out as x^T Z = (Z^T x)^T: what the wave-cooperative second pass of the six-row
// law wants (each lane carries ONE vector and all lanes the same matrix).  UNI: every lane of the wavefront runs this
// on the SAME matrix, so data-dependent branches are uniform - the rotation slots are real (scalar) branches and only
// the rotations that exist are executed; without it (one matrix per lane) the predicated forms below.
// -> false (UNI only): a rotation radius underflowed and the result is not to be used - repeat with UNI = false.
// EARLY (round 5; the truncating pseudo-inverse of the six-row law): the iteration may STOP once the part of the
// tridiagonal matrix that is still coupled is certainly kept by `pinv(rcond)` - all pivots of LDL^T(T' - cut_hi I)
// positive, cut_hi = rcond x a Gershgorin bound of lam_max - and every eigenvalue found so far is decided either way
// whatever lam_max is within its bounds (none in (cut_lo, cut_hi], cut_lo from the largest diagonal entry).  T' is then
// applied as T'^-1 by a tridiagonal solve (ql_pinv_factor / ql_pinv_solve): the SAME pseudo-inverse, no approximation.
// A zero shift on the first pass of every eigenvalue steers QL to the smallest one first: a truncating Mx_inv has one
// (83 % of random UR5 states) or two eigenvalues below the cut-off, well separated from the rest, so the iteration
// typically ends after the first - 15.8 rotations instead of 29.5 in the mean, 26 instead of 38 at the most
// (tools/ql_early_exit_prototype.py, 400 matrices, all within 1.5e-12 of numpy.linalg.pinv).
// What is left when it stops: tridiag(lam, off) in the transformed basis.
template <int K, class T>
struct QlTail {
  T rcond;    // in: relative cut-off of the pseudo-inverse
  T off[K];   // off[i] couples i and i + 1: 0 among the found eigenvalues and between them and the block that is left
  T cut;      // a FOUND eigenvalue (index <= lexit) at or below it is dropped; the block that is left is kept entirely
  int lexit;  // index of the last eigenvalue that was isolated (K - 1: the decomposition is complete)
};
template <int K, class T, int NR, bool IDENT, bool UNI, bool EARLY = false>
ABRK_INL bool ql_core(const T (&S)[K * (K + 1) / 2], T (&V)[NR][K], T (&lam)[K], QlTail<K, T>* tail = nullptr) {
  // No contraction of a * b + c beyond the fmas that are written out: the predicated one-matrix-per-lane form and the
  // uniform one-matrix-per-wavefront form are different instantiations, and the finish kernel picks between them by the
  // length of a sub-list - the result of a row must not depend on which one ran (bit for bit: a batch and its chunks).
#pragma clang fp contract(off)
  static_assert(K >= 3, "two rows: one Jacobi rotation is exact");
  static_assert(!IDENT || NR == K, "the identity has K rows");
  T a[K * (K + 1) / 2];  // packed lower triangle (the reduction works on a copy)
  sfor<K*(K + 1) / 2>([&](auto e) ABRK_LAMBDA { a[e()] = S[e()]; });
  T d[K], e[K];
  bool bad = false;
  sfor<K>([&](auto i) ABRK_LAMBDA { e[i()] = T(0); });
  // ---- Householder: A <- H_k A H_k, H_k = I - beta v v^T acting on rows / columns k+1 .. K-1;  V <- V H_k
  sfor<K - 2>([&](auto kc) ABRK_LAMBDA {
    constexpr int k = kc(), m0 = k + 1, M = K - m0;  // the block that is transformed: indices m0 .. K-1 (M of them)
    T x0 = a[tri(m0, k)];
    T sigma2 = T(0);
    sfor<M - 1>([&](auto j) ABRK_LAMBDA { sigma2 = Rm<T>::fma(a[tri(m0 + 1 + j(), k)], a[tri(m0 + 1 + j(), k)], sigma2); });
    T alpha = x0;  // the new sub-diagonal entry
    // nothing below the sub-diagonal (to rounding): the column is already tridiagonal
    bool reflect = sigma2 > Rm<T>::eps() * Rm<T>::eps() * Rm<T>::eps() * Rm<T>::eps() * (x0 * x0 + sigma2) && sigma2 > Rm<T>::tiny();
    if constexpr (UNI) reflect = uni(reflect);
    if (reflect) {
      const T n2 = Rm<T>::fma(x0, x0, sigma2);
      const T nrm = n2 * Rm<T>::rsqrt(n2);
      alpha = x0 >= T(0) ? -nrm : nrm;
      T v[M];
      v[0] = x0 - alpha;
      sfor<M - 1>([&](auto j) ABRK_LAMBDA { v[1 + j()] = a[tri(m0 + 1 + j(), k)]; });
      const T vtv = Rm<T>::fma(v[0], v[0], sigma2);
      const T beta = T(2) * Rm<T>::rcp(vtv);
      T pv[M], kk = T(0);
      sfor<M>([&](auto i) ABRK_LAMBDA {
        T acc = T(-0.0);
        sfor<M>([&](auto j) ABRK_LAMBDA { acc = Rm<T>::fma(a[tri(m0 + i(), m0 + j())], v[j()], acc); });
        pv[i()] = beta * acc;
        kk = Rm<T>::fma(pv[i()], v[i()], kk);
      });
      kk *= T(0.5) * beta;
      sfor<M>([&](auto i) ABRK_LAMBDA { pv[i()] = Rm<T>::fma(-kk, v[i()], pv[i()]); });  // q = p - (beta/2)(p.v) v
      sfor<M>([&](auto i) ABRK_LAMBDA {
        sfor<i() + 1>([&](auto j) ABRK_LAMBDA {
          a[tri(m0 + i(), m0 + j())] = Rm<T>::fma(-pv[i()], v[j()], Rm<T>::fma(-v[i()], pv[j()], a[tri(m0 + i(), m0 + j())]));
        });
      });
      // V <- V H (IDENT: row 0 of V is e_0 throughout: the reflectors never touch index 0)
      sfor<NR - (IDENT ? 1 : 0)>([&](auto rr) ABRK_LAMBDA {
        constexpr int r = rr() + (IDENT ? 1 : 0);
        T sdot = T(-0.0);
        sfor<M>([&](auto j) ABRK_LAMBDA { sdot = Rm<T>::fma(V[r][m0 + j()], v[j()], sdot); });
        sdot *= beta;
        sfor<M>([&](auto j) ABRK_LAMBDA { V[r][m0 + j()] = Rm<T>::fma(-sdot, v[j()], V[r][m0 + j()]); });
      });
    }
    e[k] = alpha;
  });
  e[K - 2] = a[tri(K - 1, K - 2)];
  sfor<K>([&](auto i) ABRK_LAMBDA { d[i()] = a[tri(i(), i())]; });
  // ---- implicit QL with Wilkinson shift on (d, e): e[i] couples i and i + 1
  bool done = false;  // EARLY: what is still coupled is certainly kept - nothing more to isolate
  int lexit = K - 1;
  T cut_exit = T(0);
  // eigenvalues 0 .. L are final (L = -1: none yet).  May the rest, indices L + 1 .. K - 1, stay as it is?  Yes if every
  // found eigenvalue is decided whatever lam_max is within [max diagonal, Gershgorin bound], and the rest is certainly
  // kept: all pivots of LDL^T(T' - hi I) positive.  An index of the rest that is decoupled on both sides and at or
  // below the lower cut-off is a decided, dropped eigenvalue of its own (a masked LAST task row - ctrlr_dof =
  // [1,1,1,1,1,0], the reference benchmark's Jaco2 setting - is an exact zero that the reflectors never touch: it
  // sits at the bottom of the tridiagonal matrix, where QL would only reach it last).
  auto exit_test = [&](auto Lc) ABRK_LAMBDA {
    constexpr int L = Lc();
    T fmx = T(0);
    sfor<L + 1>([&](auto i) ABRK_LAMBDA { fmx = Rm<T>::fmax(fmx, Rm<T>::fabs(d[i()])); });
    T dmx = d[L + 1], ger = d[L + 1] + Rm<T>::fabs(e[L + 1]);
    sfor<K - 2 - L>([&](auto ii) ABRK_LAMBDA {
      constexpr int i = L + 2 + ii();
      dmx = Rm<T>::fmax(dmx, d[i]);
      ger = Rm<T>::fmax(ger, d[i] + Rm<T>::fabs(e[i - 1]) + (i < K - 1 ? Rm<T>::fabs(e[i < K - 1 ? i : 0]) : T(0)));
    });
    const T lo = tail->rcond * Rm<T>::fmax(fmx, dmx), hi = tail->rcond * Rm<T>::fmax(fmx, ger);
    bool ok = true;
    sfor<L + 1>([&](auto i) ABRK_LAMBDA {
      const T av = Rm<T>::fabs(d[i()]);
      ok = ok && !(av > lo && !(av > hi));
    });
    const T flo = Rm<T>::eps() * ger;
    T qv = T(1);
    sfor<K - 1 - L>([&](auto ii) ABRK_LAMBDA {
      constexpr int i = L + 1 + ii();
      const T el = (i > L + 1) ? e[i > 0 ? i - 1 : 0] : T(0);  // coupling to i - 1 within the rest
      const T er = (i < K - 1) ? e[i < K - 1 ? i : 0] : T(0);
      const bool dropped = el == T(0) && er == T(0) && !(Rm<T>::fabs(d[i]) > lo);  // isolated and certainly dropped
      const T qn = (d[i] - hi) - el * el * Rm<T>::rcp(ok ? qv : T(1));
      ok = ok && (dropped || qn > flo);
      qv = dropped ? T(1) : qn;
    });
    if constexpr (UNI) ok = uni(ok);
    lexit = ok ? L : lexit;
    cut_exit = ok ? hi : cut_exit;
    done = ok;
  };
  if constexpr (EARLY) exit_test(ic<-1>{});  // nothing to isolate at all?  (then the "pseudo-inverse" is one LDL^T solve)
  sfor<K>([&](auto lc) ABRK_LAMBDA {
    constexpr int l = lc();
    if constexpr (l < K - 1) {
      if (!(EARLY && done)) {
      for (int iter = 0; iter < 40; iter++) {
        // smallest m >= l whose coupling e[m] is negligible (m = K - 1: none), and d[m] with it (picked up here, under
        // the same conditions: selecting d[m] by comparing m with every index afterwards is turned into a table
        // look-up in scratch memory by the compiler)
        int m = K - 1;
        T dm = d[K - 1];
        sfor<K - 1 - l>([&](auto jj) ABRK_LAMBDA {
          constexpr int j = K - 2 - jj();  // K-2 .. l
          // (with an absolute floor: rounding noise between two diagonal entries that are exactly 0 - masked task rows
          //  after the reflectors, in fp32 - can be a denormal, which v_rcp / v_rsq read as 0: NaNs)
          const T dd = Rm<T>::fabs(d[j]) + Rm<T>::fabs(d[j + 1]);
          const bool small = !(Rm<T>::fabs(e[j]) > Rm<T>::fmax(Rm<T>::eps() * dd, Rm<T>::tiny()));
          m = small ? j : m;
          dm = small ? d[j] : dm;
        });
        if constexpr (UNI) m = uni_int(m);
        if (m == l) break;
        T g = (d[l + 1] - d[l]) * T(0.5) * Rm<T>::rcp(e[l] == T(0) ? T(1) : e[l]);
        {
          const T r2 = Rm<T>::fma(g, g, T(1));
          const T r = r2 * Rm<T>::rsqrt(r2);
          const T den = g + (g >= T(0) ? r : -r);
          g = dm - d[l] + e[l] * Rm<T>::rcp(den);  // d[m] - d[l] + e[l] / (g + sign(r, g))
        }
        if constexpr (EARLY) g = iter == 0 ? dm : g;  // zero shift on an eigenvalue's first pass: towards the smallest
        T sn = T(1), cs = T(1), pp = T(0);
        bool stop = false;  // a zero rotation radius ends the pass early (tql2's recovery from underflow)
        if constexpr (UNI) {
          // one matrix per wavefront: the slots that exist, behind scalar compares (QlChain)
          QlChain<K, T, NR, l, K - 2>::run(d, e, V, m, sn, cs, pp, g, bad);
        } else {
        // Every slot K-2 .. l is executed by every lane, inactive ones (i >= m, or after a stop) as the identity
        // rotation: the pass is ONE basic block, so the six independent eigenvector updates of a slot overlap with the
        // next slot's scalar recurrence (a lone lane is latency-bound: with a branch per slot nothing overlapped)
        sfor<K - 1 - l>([&](auto ii) ABRK_LAMBDA {
          constexpr int i = K - 2 - ii();  // K-2 .. l
          const bool act = (i < m) && !stop;
          const T f = sn * e[i], b = cs * e[i];
          const T r2 = Rm<T>::fma(f, f, g * g);
          const bool zero = !(r2 > Rm<T>::tiny());  // (a denormal radius counts as zero: v_rsq would make it infinite)
          const bool go = act && !zero;
          d[i + 1] = (act && zero) ? d[i + 1] - pp : d[i + 1];
          stop = stop || (act && zero);
          const T ir = Rm<T>::rsqrt(go ? r2 : T(1)), r = r2 * ir;
          const T sn_n = f * ir, cs_n = g * ir;
          const T g1 = d[i + 1] - pp;
          const T rr = Rm<T>::fma(d[i] - g1, sn_n, T(2) * cs_n * b);
          const T pp_n = sn_n * rr;
          e[i + 1] = go ? r : e[i + 1];
          d[i + 1] = go ? g1 + pp_n : d[i + 1];
          g = go ? Rm<T>::fma(cs_n, rr, -b) : g;
          sn = go ? sn_n : sn;
          cs = go ? cs_n : cs;
          pp = go ? pp_n : pp;
          const T se = go ? sn_n : T(0), ce = go ? cs_n : T(1);
          sfor<NR>([&](auto kk) ABRK_LAMBDA {
            const T fz = V[kk()][i + 1];
            V[kk()][i + 1] = Rm<T>::fma(se, V[kk()][i], ce * fz);
            V[kk()][i] = Rm<T>::fma(ce, V[kk()][i], -(se * fz));
          });
        });
        }
        // e[m] = 0 in either case (as value selects: an `if (m == j) e[j] = 0` chain becomes ONE store through a
        // selected pointer, which sends e[] to scratch memory)
        sfor<K - 1 - l>([&](auto jj) ABRK_LAMBDA {
          constexpr int j = l + jj();
          e[j] = (m == j) ? T(0) : e[j];
        });
        if (UNI || !stop) {
          d[l] -= pp;
          e[l] = g;
        }
      }
      if constexpr (EARLY && l < K - 2) exit_test(ic<l>{});
      }
    }
  });
  sfor<K>([&](auto i) ABRK_LAMBDA { lam[i()] = d[i()]; });
  if constexpr (EARLY) {
    // complete decomposition: the cut-off from the largest eigenvalue itself (numpy.linalg.pinv)
    T smax = T(0);
    sfor<K>([&](auto i) ABRK_LAMBDA { smax = Rm<T>::fmax(smax, Rm<T>::fabs(d[i()])); });
    tail->cut = done ? cut_exit : tail->rcond * smax;
    tail->lexit = lexit;
    sfor<K>([&](auto i) ABRK_LAMBDA { tail->off[i()] = (done && i() > lexit && i() < K - 1) ? e[i() < K - 1 ? i() : 0] : T(0); });
  }
  return UNI ? !uni(bad) : true;
}
// The pseudo-inverse of what ql_core<EARLY> left - diag over the found eigenvalues (dropped ones act as 0), T'^-1 over
// the block that is still coupled - as one LDL^T recurrence over all K indices: couplings are 0 where deflated, so the
// found part comes out as 1 / lam.  factor: li[i] = off[i-1] / q[i-1], iq[i] = 1 / q[i] (0 for a dropped eigenvalue);
// solve: y = pinv x.  Contraction pinned off: the lane and wave forms of the finish kernel call the same two routines.
template <int K, class T>
ABRK_INL void ql_pinv_factor(const T (&lam)[K], const QlTail<K, T>& t, T (&li)[K], T (&iq)[K]) {
#pragma clang fp contract(off)
  T q = lam[0];
  li[0] = T(0);
  sfor<K>([&](auto ic_) ABRK_LAMBDA {
    constexpr int i = ic_();
    if constexpr (i > 0) {
      li[i] = t.off[i - 1] * iq[i - 1];
      q = lam[i] - li[i] * t.off[i - 1];
    }
    // a found eigenvalue by its magnitude (numpy.linalg.pinv: singular values); an index of the part that was left as
    // it is by its pivot: the pivots of a kept block are >= its smallest eigenvalue > cut, an isolated dropped
    // index (see ql_core's exit test) has the pivot lam[i] <= cut
    const bool keep = i > t.lexit ? q > t.cut : Rm<T>::fabs(lam[i]) > t.cut;
    iq[i] = keep ? rcp(keep ? q : T(1)) : T(0);
  });
}
template <int K, class T>
ABRK_INL void ql_pinv_solve(const T (&li)[K], const T (&iq)[K], const T (&x)[K], T (&y)[K]) {
#pragma clang fp contract(off)
  T z[K];
  z[0] = x[0];
  sfor<K - 1>([&](auto ii) ABRK_LAMBDA { z[ii() + 1] = x[ii() + 1] - li[ii() + 1] * z[ii()]; });
  sfor<K>([&](auto i) ABRK_LAMBDA { z[i()] = z[i()] * iq[i()]; });
  y[K - 1] = z[K - 1];
  sfor<K - 1>([&](auto ii) ABRK_LAMBDA {
    constexpr int i = K - 2 - ii();
    y[i] = z[i] - li[i + 1] * y[i + 1];
  });
}
template <int K, class T>
ABRK_INL void ql_eig(const T (&S)[K * (K + 1) / 2], T (&V)[K][K], T (&lam)[K]) {
  sfor<K>([&](auto i) ABRK_LAMBDA { sfor<K>([&](auto j) ABRK_LAMBDA { V[i()][j()] = (i() == j()) ? T(1) : T(0); }); });
  ql_core<K, T, K, true, false>(S, V, lam);
}

// eigen-decomposition behind a truncating pinv: Jacobi up to 3 x 3 (one / three rotation pairs), QL above
template <int K, class T>
ABRK_INL void sym_eig(T (&S)[K * (K + 1) / 2], T (&V)[K][K], T (&lam)[K]) {
  if constexpr (K >= 4) ql_eig<K>(S, V, lam);  // (cyclic Jacobi for every size: rounds 1-2)
  else jacobi_eig<K>(S, V, lam);
}

// Direct eigen-decomposition of a symmetric 3 x 3 matrix (packed lower in) - no sweeps, no data-dependent trip
// count, so the lanes of a wavefront stay together (AvoidObstacles runs one truncated pinv per near segment-obstacle
// pair; with the Jacobi sweeps that was ~12.7 k vector instructions per row).  The hybrid of the closed form and one
// rotation: (1) the eigenvalue that is ISOLATED from the other two - the largest when det(A - mI) >= 0, else the
// smallest - from the trigonometric solution of the characteristic cubic, where it is well conditioned (the two
// others may coincide; their closed form loses half the digits there and is not used); (2) its eigenvector as the
// largest cross product of two rows of (A - lam I), a rank-2 matrix whose non-zero eigenvalues are at least sqrt(3)
// normalised units away from 0; (3) an orthonormal basis of the complement, the 2 x 2 block of A in it, and ONE
// Jacobi rotation, exact for two dimensions.  V is orthonormal by construction; eigenvalues carry errors of a few
// eps |A| (prototype against numpy.linalg.eigvalsh / pinv over 2e5 rank-1, rank-2, clustered and graded matrices:
// 1.6e-15 |A|, pinv(rcond=0.01) within 1.2e-13).  The order of the pairs is (isolated, pair, pair), NOT by row:
// callers that rely on Jacobi leaving a decoupled row's eigenpair in place (osc_law's masked rows) keep jacobi_eig.
template <class T>
ABRK_INL void sym3_eig(const T (&S)[6], T (&V)[3][3], T (&lam)[3]) {
  const T a00 = S[tri(0, 0)], a01 = S[tri(1, 0)], a11 = S[tri(1, 1)], a02 = S[tri(2, 0)], a12 = S[tri(2, 1)],
          a22 = S[tri(2, 2)];
  const T m = (a00 + a11 + a22) * T(1.0 / 3.0);
  const T b00 = a00 - m, b11 = a11 - m, b22 = a22 - m;
  const T p2 = (b00 * b00 + b11 * b11 + b22 * b22 + T(2) * (a01 * a01 + a02 * a02 + a12 * a12)) * T(1.0 / 6.0);
  const T lim = T(4) * Rm<T>::eps() * Rm<T>::fabs(m);
  if (!(p2 > lim * lim)) {  // a multiple of the identity to working precision: any basis serves
    sfor<3>([&](auto i) ABRK_LAMBDA { sfor<3>([&](auto j) ABRK_LAMBDA { V[i()][j()] = (i() == j()) ? T(1) : T(0); }); });
    lam[0] = a00;
    lam[1] = a11;
    lam[2] = a22;
    return;
  }
  const T ip = Rm<T>::rsqrt(p2);
  const T c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
  T hd = T(0.5) * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
  hd = Rm<T>::fmin(T(1), Rm<T>::fmax(T(-1), hd));
  const T phi = Rm<T>::acos(hd) * T(1.0 / 3.0);  // in [0, pi/3]
  T sn, cs;
  Rm<T>::sincos_fast(phi, sn, cs);
  // roots of the normalised cubic: 2 cos(phi) >= 2 cos(phi - 2pi/3) >= 2 cos(phi + 2pi/3)
  const T s = hd >= T(0) ? T(2) * cs : -cs - T(1.7320508075688772935) * sn;
  const T r0[3] = {c00 - s, c01, c02}, r1[3] = {c01, c11 - s, c12}, r2[3] = {c02, c12, c22 - s};
  T x0[3], x1[3], x2[3];
  cross3(r0, r1, x0);
  cross3(r0, r2, x1);
  cross3(r1, r2, x2);
  const T n0 = dot3(x0, x0), n1 = dot3(x1, x1), n2 = dot3(x2, x2);
  const bool u1 = n1 > n0;
  T nb = u1 ? n1 : n0, xb[3];
  sfor<3>([&](auto r) ABRK_LAMBDA { xb[r()] = u1 ? x1[r()] : x0[r()]; });
  const bool u2 = n2 > nb;
  nb = u2 ? n2 : nb;
  const T inb = Rm<T>::rsqrt(nb);
  T vs[3];
  sfor<3>([&](auto r) ABRK_LAMBDA { vs[r()] = (u2 ? x2[r()] : xb[r()]) * inb; });
  // u perpendicular to vs from its two larger components (their squares sum to at least 1/2), w = vs x u
  const bool xbig = Rm<T>::fabs(vs[0]) > Rm<T>::fabs(vs[1]);
  const T ih = Rm<T>::rsqrt(xbig ? vs[0] * vs[0] + vs[2] * vs[2] : vs[1] * vs[1] + vs[2] * vs[2]);
  const T u[3] = {xbig ? -vs[2] * ih : T(0), xbig ? T(0) : vs[2] * ih, xbig ? vs[0] * ih : -vs[1] * ih};
  T w[3], Au[3], Aw[3], Av[3];
  cross3(vs, u, w);
  symv<3>(S, u, Au);
  symv<3>(S, w, Aw);
  symv<3>(S, vs, Av);
  const T k00 = dot3(u, Au), k01 = dot3(u, Aw), k11 = dot3(w, Aw);
  const T t = jacobi_tan(k11 - k00, T(2) * k01);  // 0 when k01 = 0
  const T c = Rm<T>::rsqrt(t * t + T(1)), sr = t * c;
  lam[0] = dot3(vs, Av);
  lam[1] = k00 - t * k01;
  lam[2] = k11 + t * k01;
  sfor<3>([&](auto r) ABRK_LAMBDA {
    V[r()][0] = vs[r()];
    V[r()][1] = c * u[r()] - sr * w[r()];
    V[r()][2] = sr * u[r()] + c * w[r()];
  });
}

// One-sided (Hestenes) Jacobi SVD: orthogonalise the K columns G[:,0..K-1] (length N each, G = J^T),
// accumulating the rotations in V (K x K).  Afterwards G = U diag(sig), so for J = G^T (K x N):
//   pinv(J)[i][r] = sum_{sig_j > rcond * sig_max} G[i][j] V[r][j] / sig_j^2
// (numpy.linalg.pinv, default rcond 1e-15: sliding.py:72, inverse_kinematics.py:104-121).  Small singular
// values keep their relative accuracy, which an eigen-decomposition of J J^T would lose.
template <int K, int N, class T>
ABRK_INL void pinv_KxN(const T (&J)[N][K] /* J[i][r] = J(r,i) */, T rcond, T (&P)[N][K] /* P[i][r] */) {
  if constexpr (K <= N) {
    // Well-conditioned full row rank (the common case along an IK path or a sliding-mode step): pinv(J) =
    // J^T (J J^T)^-1 through a K x K Cholesky factor.  cond(J J^T) <= trace^K / det; below 1e6 (fp32: 1e2) the
    // squared conditioning costs at most ~1e-10 (1e-5) relative and nothing is truncated at rcond = 1e-15.
    T A[K * (K + 1) / 2], L[K * (K + 1) / 2], il[K], trace = T(0);
    sfor<K>([&](auto r) ABRK_LAMBDA {
      sfor<r() + 1>([&](auto c) ABRK_LAMBDA {
        T acc = T(-0.0);
        sfor<N>([&](auto i) ABRK_LAMBDA { acc += J[i()][r()] * J[i()][c()]; });
        A[tri(r(), c())] = acc;
      });
      trace += A[tri(r(), r())];
    });
    bool ok = chol<K>(A, L, il);
    T det = T(1), tk = T(1);
    sfor<K>([&](auto r) ABRK_LAMBDA {
      det *= L[tri(r(), r())] * L[tri(r(), r())];
      tk *= trace;
    });
    if (ok && tk < det * (sizeof(T) == 8 ? T(1e6) : T(1e2))) {
      T Ai[K * (K + 1) / 2];
      chol_inverse<K>(L, il, Ai);
      sfor<N>([&](auto i) ABRK_LAMBDA {
        sfor<K>([&](auto r) ABRK_LAMBDA {
          T acc = T(-0.0);
          sfor<K>([&](auto c) ABRK_LAMBDA { acc += J[i()][c()] * Ai[tri(c(), r())]; });
          P[i()][r()] = acc;
        });
      });
      return;
    }
  }
  T G[N][K];
  T V[K][K];
  sfor<N>([&](auto i) ABRK_LAMBDA { sfor<K>([&](auto r) ABRK_LAMBDA { G[i()][r()] = J[i()][r()]; }); });
  sfor<K>([&](auto a) ABRK_LAMBDA { sfor<K>([&](auto b) ABRK_LAMBDA { V[a()][b()] = (a() == b()) ? T(1) : T(0); }); });
  for (int sweep = 0; sweep < 40; sweep++) {
    T worst = T(0);
    sfor<K>([&](auto qq) ABRK_LAMBDA {
      sfor<qq()>([&](auto pp) ABRK_LAMBDA {
        constexpr int p = pp(), q = qq();
        T alpha = T(-0.0), beta = T(-0.0), gamma = T(-0.0);
        sfor<N>([&](auto i) ABRK_LAMBDA {
          alpha += G[i()][p] * G[i()][p];
          beta += G[i()][q] * G[i()][q];
          gamma += G[i()][p] * G[i()][q];
        });
        // |gamma| > eps sqrt(alpha beta), without the square root
        if (gamma * gamma > Rm<T>::eps() * Rm<T>::eps() * alpha * beta && Rm<T>::fabs(gamma) > Rm<T>::tiny()) {
          worst = T(1);  // a rotation was applied in this sweep
          T t = jacobi_tan(beta - alpha, T(2) * gamma);
          T c = Rm<T>::rsqrt(T(1) + t * t);
          T s = c * t;
          sfor<N>([&](auto i) ABRK_LAMBDA {
            T gp = G[i()][p], gq = G[i()][q];
            G[i()][p] = c * gp - s * gq;
            G[i()][q] = s * gp + c * gq;
          });
          sfor<K>([&](auto a) ABRK_LAMBDA {
            T vp = V[a()][p], vq = V[a()][q];
            V[a()][p] = c * vp - s * vq;
            V[a()][q] = s * vp + c * vq;
          });
        }
      });
    });
    if (!(worst > T(0))) break;
  }
  T sig2[K], smax2 = T(0);
  sfor<K>([&](auto j) ABRK_LAMBDA {
    T acc = T(-0.0);
    sfor<N>([&](auto i) ABRK_LAMBDA { acc += G[i()][j()] * G[i()][j()]; });
    sig2[j()] = acc;
    smax2 = Rm<T>::fmax(smax2, acc);
  });
  T w[K];
  // sigma_j > rcond * sigma_max, compared on the squares
  T cut2 = rcond * rcond * smax2;
  sfor<K>([&](auto j) ABRK_LAMBDA {
    bool keep = sig2[j()] > cut2;
    w[j()] = keep ? rcp(keep ? sig2[j()] : T(1)) : T(0);
  });
  sfor<N>([&](auto i) ABRK_LAMBDA {
    sfor<K>([&](auto r) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<K>([&](auto j) ABRK_LAMBDA { acc += G[i()][j()] * V[r()][j()] * w[j()]; });
      P[i()][r()] = acc;
    });
  });
}
template <int N, class T>
ABRK_INL void pinv_3xN(const T (&J)[N][3], T rcond, T (&P)[N][3]) {
  pinv_KxN<3, N, T>(J, rcond, P);
}

// ---------------------------------------------------------------- quaternions (utils/transformations.py)
// quaternion_from_matrix(isprecise=False) (transformations.py:1192-1271): dominant
// eigenvector of the symmetric 4x4 K/3 built from R.  For a (near-)rotation K/3 has
// eigenvalues {1, -1/3, -1/3, -1/3}, so B = K/3 + I/3 is rank one up to the
// non-orthogonality of R and three power steps from its heaviest column converge to
// rounding (error ratio^3 with ratio ~ |R^T R - I|).  Returns unit (w,x,y,z), w >= 0
// (sign rule transformations.py:1269; unit_vector base_config.py:315).
// The iterate is normalised once, at the end: the dominant eigenvalue is 4/3, so STEPS steps scale it by (4/3)^STEPS
// and nothing can overflow (a reciprocal square root per step was a sixth of the orientation error's cost).
// STEPS = 1 where R is a product of exact joint rotations (orthogonal to rounding: the frame rotation of an orthogonal
// chain inside the fused kernels) - the heaviest column is then already the eigenvector to ~1e-16.
// (Every sum of products in this section is written as an explicit fma chain under `fp contract(off)`: which multiply the
//  compiler fuses with which add otherwise depends on the code around the inlined call, and the two kernels that run
//  the six-row law on a row - first pass and complete row program - must round alike: a row's bits do not depend on the
//  batch it arrives in, G::test_gpu_six_row_bits_do_not_depend_on_the_batch_size.)
template <class T>
ABRK_INL T dot4_fma(T a0, T b0, T a1, T b1, T a2, T b2, T a3, T b3) {
  return Rm<T>::fma(a3, b3, Rm<T>::fma(a2, b2, Rm<T>::fma(a1, b1, a0 * b0)));
}
template <class T>
ABRK_INL T dot3_fma(T a0, T b0, T a1, T b1, T a2, T b2) {
  return Rm<T>::fma(a2, b2, Rm<T>::fma(a1, b1, a0 * b0));
}
template <class T, int STEPS = 3>
ABRK_INL void quat_from_R(const T (&R)[9], T (&qo)[4]) {
#pragma clang fp contract(off)
  const T m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m21 = R[7],
          m22 = R[8];
  const T th = T(1) / T(3);
  T Bm[4][4];
  Bm[0][0] = Rm<T>::fma(m00 - m11 - m22, th, th);
  Bm[1][1] = Rm<T>::fma(m11 - m00 - m22, th, th);
  Bm[2][2] = Rm<T>::fma(m22 - m00 - m11, th, th);
  Bm[3][3] = Rm<T>::fma(m00 + m11 + m22, th, th);
  Bm[1][0] = Bm[0][1] = (m01 + m10) * th;
  Bm[2][0] = Bm[0][2] = (m02 + m20) * th;
  Bm[2][1] = Bm[1][2] = (m12 + m21) * th;
  Bm[3][0] = Bm[0][3] = (m21 - m12) * th;
  Bm[3][1] = Bm[1][3] = (m02 - m20) * th;
  Bm[3][2] = Bm[2][3] = (m10 - m01) * th;
  // heaviest diagonal -> start vector = that column
  int best = 0;
  T bd = Bm[0][0];
  sfor<3>([&](auto jj) ABRK_LAMBDA {
    constexpr int j = jj() + 1;
    bool gt = Bm[j][j] > bd;
    best = gt ? j : best;
    bd = gt ? Bm[j][j] : bd;
  });
  T v[4];
  sfor<4>([&](auto i) ABRK_LAMBDA {
    v[i()] = best == 0 ? Bm[i()][0] : best == 1 ? Bm[i()][1] : best == 2 ? Bm[i()][2] : Bm[i()][3];
  });
  sfor<STEPS>([&](auto it) ABRK_LAMBDA {
    T w[4];
    sfor<4>([&](auto i) ABRK_LAMBDA {
      w[i()] = dot4_fma(Bm[i()][0], v[0], Bm[i()][1], v[1], Bm[i()][2], v[2], Bm[i()][3], v[3]);
    });
    sfor<4>([&](auto i) ABRK_LAMBDA { v[i()] = w[i()]; });
  });
  T nn = dot4_fma(v[0], v[0], v[1], v[1], v[2], v[2], v[3], v[3]);
  T inv = Rm<T>::rsqrt(nn);
  T sg = (v[3] < T(0)) ? -inv : inv;
  qo[0] = v[3] * sg;
  qo[1] = v[0] * sg;
  qo[2] = v[1] * sg;
  qo[3] = v[2] * sg;
}

// sin / cos of three angles: through the calling kernel's LDS table where it keeps one (`tab`; sincos_all_tab: one
// straight-line block, the library behind one rarely taken branch), else the polynomial / library routine
template <class T, bool TAB>
ABRK_INL void sincos3(T a0, T a1, T a2, T (&sv)[3][2], const void* tab) {
  if constexpr (TAB) {
    const T x[3] = {a0, a1, a2};
    sincos_all_tab<3>(x, sv, tab);
  } else {
    Rm<T>::sincos(a0, sv[0][0], sv[0][1]);
    Rm<T>::sincos(a1, sv[1][0], sv[1][1]);
    Rm<T>::sincos(a2, sv[2][0], sv[2][1]);
  }
}
// quaternion_from_euler(ai, aj, ak, 'rxyz') (transformations.py:1096-1150), then unit_vector
template <class T, bool TAB = false>
ABRK_INL void quat_from_euler_rxyz(T ai, T aj, T ak, T (&q)[4], const void* tab = nullptr) {
#pragma clang fp contract(off)
  // axes 'rxyz' -> (firstaxis, parity, repetition, frame) = (2, 1, 0, 1): i=3, j=2, k=1
  T t = ai;
  ai = ak;
  ak = t;
  aj = -aj;
  T sv[3][2];
  sincos3<T, TAB>(ai / T(2), aj / T(2), ak / T(2), sv, tab);
  const T si = sv[0][0], ci = sv[0][1], sj = sv[1][0], cj = sv[1][1], sk = sv[2][0], ck = sv[2][1];
  T cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  q[0] = Rm<T>::fma(cj, cc, sj * ss);
  q[3] = Rm<T>::fma(cj, sc, -(sj * cs));
  q[2] = -Rm<T>::fma(cj, ss, sj * cc);
  q[1] = Rm<T>::fma(cj, cs, -(sj * sc));
  T inv = Rm<T>::rsqrt(dot4_fma(q[0], q[0], q[1], q[1], q[2], q[2], q[3], q[3]));
  sfor<4>([&](auto i) ABRK_LAMBDA { q[i()] *= inv; });
}

// quaternion_from_euler(ai, aj, ak, 'sxyz') (transformations.py:1096-1150; axes tuple (0,0,0,0): i=1,j=2,k=3),
// then unit_vector - the target orientation of inverse_kinematics.py:73-82
template <class T>
ABRK_INL void quat_from_euler_sxyz(T ai, T aj, T ak, T (&q)[4]) {
#pragma clang fp contract(off)
  T si, ci, sj, cj, sk, ck;
  Rm<T>::sincos(ai / T(2), si, ci);
  Rm<T>::sincos(aj / T(2), sj, cj);
  Rm<T>::sincos(ak / T(2), sk, ck);
  T cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  q[0] = Rm<T>::fma(cj, cc, sj * ss);
  q[1] = Rm<T>::fma(cj, sc, -(sj * cs));
  q[2] = Rm<T>::fma(cj, ss, sj * cc);
  q[3] = Rm<T>::fma(cj, cs, -(sj * sc));
  T inv = Rm<T>::rsqrt(dot4_fma(q[0], q[0], q[1], q[1], q[2], q[2], q[3], q[3]));
  sfor<4>([&](auto i) ABRK_LAMBDA { q[i()] *= inv; });
}

// euler_matrix(ai, aj, ak, 'rxyz')[:3,:3] (transformations.py:973-1032)
template <class T, bool TAB = false>
ABRK_INL void euler_matrix_rxyz(T ai, T aj, T ak, T (&M)[9], const void* tab = nullptr) {
#pragma clang fp contract(off)
  // (2,1,0,1): i=2, j=1, k=0; swap ai/ak; negate all
  T t = ai;
  ai = -ak;
  ak = -t;
  aj = -aj;
  T sv[3][2];
  sincos3<T, TAB>(ai, aj, ak, sv, tab);
  const T si = sv[0][0], ci = sv[0][1], sj = sv[1][0], cj = sv[1][1], sk = sv[2][0], ck = sv[2][1];
  T cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  constexpr int i = 2, j = 1, k = 0;
  M[i * 3 + i] = cj * ck;
  M[i * 3 + j] = Rm<T>::fma(sj, sc, -cs);
  M[i * 3 + k] = Rm<T>::fma(sj, cc, ss);
  M[j * 3 + i] = cj * sk;
  M[j * 3 + j] = Rm<T>::fma(sj, ss, cc);
  M[j * 3 + k] = Rm<T>::fma(sj, cs, -sc);
  M[k * 3 + i] = -sj;
  M[k * 3 + j] = cj * si;
  M[k * 3 + k] = cj * ci;
}

// quaternion_multiply(q1, q0) (transformations.py:1274-1290)
template <class T>
ABRK_INL void quat_mul(const T (&q1)[4], const T (&q0)[4], T (&r)[4]) {
#pragma clang fp contract(off)
  T w0 = q0[0], x0 = q0[1], y0 = q0[2], z0 = q0[3];
  T w1 = q1[0], x1 = q1[1], y1 = q1[2], z1 = q1[3];
  r[0] = Rm<T>::fma(w1, w0, -dot3_fma(x1, x0, y1, y0, z1, z0));
  r[1] = Rm<T>::fma(w1, x0, Rm<T>::fma(-z1, y0, Rm<T>::fma(y1, z0, x1 * w0)));
  r[2] = Rm<T>::fma(w1, y0, Rm<T>::fma(z1, x0, Rm<T>::fma(y1, w0, -(x1 * z0))));
  r[3] = Rm<T>::fma(w1, z0, Rm<T>::fma(z1, w0, Rm<T>::fma(-y1, x0, x1 * y0)));
}

// _calc_orientation_forces (osc.py:149-196)
// QSTEPS: power steps of quat_from_R (1 where Re is orthogonal to rounding)
// TAB: sin / cos of the target's Euler angles through the calling kernel's LDS table (`tab`)
template <class T, int QSTEPS = 3, bool TAB = false>
ABRK_INL void orientation_forces(int alg, const T (&Re)[9], const T (&abg)[3], T (&uo)[3], const void* tab = nullptr) {
#pragma clang fp contract(off)
  if (alg == 0) {
    T qd[4], qe[4], qec[4], qr[4];
    quat_from_euler_rxyz<T, TAB>(abg[0], abg[1], abg[2], qd, tab);
    quat_from_R<T, QSTEPS>(Re, qe);
    qec[0] = qe[0];
    qec[1] = -qe[1];
    qec[2] = -qe[2];
    qec[3] = -qe[3];
    quat_mul(qd, qec, qr);
    T sg = qr[0] > T(0) ? T(1) : (qr[0] < T(0) ? T(-1) : T(0));
    sfor<3>([&](auto r) ABRK_LAMBDA { uo[r()] = -qr[1 + r()] * sg; });
  } else {
    T Rd[9], Red[9], qed[4];
    euler_matrix_rxyz<T, TAB>(abg[0], abg[1], abg[2], Rd, tab);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      sfor<3>([&](auto c) ABRK_LAMBDA {
        Red[r() * 3 + c()] = dot3_fma(Re[0 * 3 + r()], Rd[0 * 3 + c()], Re[1 * 3 + r()], Rd[1 * 3 + c()],
                                      Re[2 * 3 + r()], Rd[2 * 3 + c()]);
      });
    });
    quat_from_R<T, QSTEPS>(Red, qed);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      uo[r()] = -dot3_fma(Re[r() * 3 + 0], qed[1], Re[r() * 3 + 1], qed[2], Re[r() * 3 + 2], qed[3]);
    });
  }
}

// Python float modulo: result takes the sign of the (positive) divisor
template <class T>
ABRK_INL T pymod_pos(T a, T b) {
  T r = Rm<T>::fmod(a, b);
  return (r < T(0)) ? r + b : r;
}

// joint-space command of a secondary controller: u_null = M v   (damping.py:31-32,
// resting_config.py:25-42 + joint.py:42-46,118-123)
template <int N, class T>
ABRK_INL void null_command(const NullP<T>& c, const T (&q)[N], const T (&dq)[N], T (&v)[N]) {
  const T pi = T(3.141592653589793238462643383279502884);
  if (c.kind == 1) {
    sfor<N>([&](auto i) ABRK_LAMBDA { v[i()] += -c.kv * dq[i()]; });
  } else {
    sfor<N>([&](auto i) ABRK_LAMBDA {
      T qt = c.mask[i()] ? pymod_pos(c.rest[i()] - q[i()] + pi, pi * T(2)) - pi : T(0);
      v[i()] += c.kp * qt + c.kv * (T(0) - dq[i()]);
    });
  }
}

// value the optimiser must treat as unknown from here on (device code only; the host check build does not care)
template <class T>
ABRK_INL void opaque(T& x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
}

// ---------------------------------------------------------------- OSC.generate, one row
// KM = 3 or 2 (FAST: task rows are exactly x,y,z / x,y of the EE) or 6 (all six task rows, unselected
// rows masked: their Jacobian row is zeroed and Mx_inv gets a unit diagonal there, which
// leaves det, the inverse and the singular values of the selected block unchanged).
//
// ---- the control law proper (osc.py:244-318), fed with M, g, C dq, the task Jacobian (Jv/Jw
// columns), the task point p and the frame rotation RF.  Called by osc_row after the fused
// kinematics, and by the law-only kernel for robot_configs whose J/M/g come from elsewhere
// (the reference's duck-typed boundary, e.g. MujocoConfig).  The gravity torque is gscale * gz
// (fused path: 9.81 * gz accumulators; law-only path: -1 * g).
template <int N, class T, int KM, bool USE_C, int FEAT>
ABRK_INL void osc_law(const OscP<T>& P, const T (&Ms)[N * (N + 1) / 2], const T (&gz)[N], T gscale,
                      const T (&cvec)[N], const T (&Jv)[N][3], const T (&Jw)[N][3], const T (&p)[3],
                      const T (&RF)[9], const T (&q)[N], const T (&dq)[N], const T (&tgt)[6], bool tv_given,
                      const T (&tvin)[6], bool have_ierr, T (&ierr)[6], bool have_ext, const T (&une)[N],
                      T (&u)[N], T (&ts)[N], bool* defer = nullptr, bool* singular = nullptr) {
  constexpr bool FAST = (KM <= 3);
  T Jr[N][KM];
  bool sel[KM];
  sfor<KM>([&](auto r) ABRK_LAMBDA {
    sel[r()] = FAST ? true : (P.dof[r()] != 0);
    sfor<N>([&](auto i) ABRK_LAMBDA {
      T val = (r() < 3) ? Jv[i()][r() % 3] : Jw[i()][r() % 3];
      Jr[i()][r()] = sel[r()] ? val : T(0);
    });
  });

  ABRK_MARK("law3:chol_M");
  // _Mx (osc.py:120-147): Mx_inv = J M^-1 J^T through the Cholesky factor of M
  T L[N * (N + 1) / 2], il[N];
  T minpiv = T(1);
  chol<N, T, false>(Ms, L, il, &minpiv);
  // (the kernels take the verdict along and store the flag after the row's outputs: a store behind a branch HERE
  //  splits the law's one scheduling region - +14 registers on Jaco2's x,y,z kernels, 44 B of scratch and 8 % on
  //  BASELINE config 3.  And they take it along as a WAVE-level scalar - "some row of this wavefront": the flag says no
  //  more than that anyway - because a per-lane bool rides in vector registers: +10 .. 12 on the six-row kernels)
  if (singular) *singular = any_lane(!(minpiv > T(0)));
  else flag_singular(P, minpiv);
  ABRK_MARK("law3:Y");
  T Y[N][KM];
  sfor<KM>([&](auto r) ABRK_LAMBDA {
    T b[N], x[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { b[i()] = Jr[i()][r()]; });
    chol_fwd<N>(L, il, b, x);
    sfor<N>([&](auto i) ABRK_LAMBDA { Y[i()][r()] = x[i()]; });
  });
  ABRK_MARK("law3:Am");
  T Am[KM * (KM + 1) / 2];
  T trace = T(0);
  sfor<KM>([&](auto r) ABRK_LAMBDA {
    sfor<r() + 1>([&](auto c) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { acc += Y[i()][r()] * Y[i()][c()]; });
      Am[tri(r(), c())] = acc;
    });
    trace += sel[r()] ? Am[tri(r(), r())] : T(0);
    if (!sel[r()]) Am[tri(r(), r())] = T(1);
  });
  ABRK_MARK("law3:cholA_cert");
  T LA[KM * (KM + 1) / 2], ila[KM], Mx[KM * (KM + 1) / 2];
  // x,y,z / x,y: Mx by cofactors (spd_inverse_small); otherwise through the Cholesky factor of Mx_inv
  // (fp64 only: the cofactors carry a relative error of eps * trace^K / det - see below -, which single precision
  //  cannot afford on the everyday rows)
  constexpr bool kClosed = FAST && sizeof(T) == 8;
  bool okA;
  T det = T(1);
  if constexpr (kClosed) {
    okA = spd_inverse_small<KM>(Am, Mx, det);
    // Accuracy gate.  Cofactors are differences of products of entries of size |A|, so adj(A) and det carry an absolute
    // error of eps |A|^(K-1) and eps |A|^K: relative to det that is eps * trace^K / det - as good as a factorisation
    // (eps * cond) while one eigenvalue is small, eps * cond^2 when two are.  Up to trace^K / det = 1e6 (error <=
    // 1e-9; none of 12 288 golden rows of the built-in arms is above 1e5, and whatever passes the first certificate is
    // below 1e4) the cofactor result stands; beyond it - rows deep in the pinv branch, nearly all of which go on to the
    // eigen-decomposition anyway - the Cholesky factor is taken after all (cold code).
    T tk = trace;
    sfor<KM - 1>([&](auto) ABRK_LAMBDA { tk *= trace; });
    if (!(okA && det * T(1e6) >= tk)) {
      okA = chol<KM>(Am, LA, ila);
      det = T(1);
      sfor<KM>([&](auto r) ABRK_LAMBDA { det *= LA[tri(r(), r())] * LA[tri(r(), r())]; });
      chol_inverse<KM>(LA, ila, Mx);
    }
  } else {
    okA = chol<KM>(Am, LA, ila);
    sfor<KM>([&](auto r) ABRK_LAMBDA { det *= LA[tri(r(), r())] * LA[tri(r(), r())]; });
  }
  // (factor form) FEAT = 0 needs Mx only once (Mx u_task): solved with the factor instead of forming the inverse
  bool mx_explicit = kClosed || FEAT != 0;
  if constexpr (!kClosed && FEAT != 0) chol_inverse<KM>(LA, ila, Mx);
  const T thr = T(1e-3), rcond = T(1e-3) * T(0.1);
  if (!(okA && det >= thr)) {
    // pinv branch (osc.py:142-145).  pinv == inv unless some singular value is below
    // rcond*max; det >= rcond*trace^K proves none is (lam_min >= det/lam_max^(K-1)).
    T bound = rcond;
    sfor<KM>([&](auto r) ABRK_LAMBDA { bound *= sel[r()] ? trace : T(1); });
    bool truncates = !(okA && det > bound);
    if (truncates && okA) {
      // second, much tighter certificate (the determinant bound is hopeless for six rows of mixed units):
      // lam_max(A) <= trace(A) and 1/lam_min(A) <= trace(A^-1), so trace(A) trace(A^-1) < 1/rcond proves
      // every singular value is above rcond * max - pinv is the inverse the factor already gives
      if constexpr (!kClosed && FEAT == 0) chol_inverse<KM>(LA, ila, Mx);
      T tinv = T(0);
      sfor<KM>([&](auto r) ABRK_LAMBDA { tinv += sel[r()] ? Mx[tri(r(), r())] : T(0); });
      if (trace * tinv * rcond < T(1)) {
        truncates = false;
        mx_explicit = true;
      }
    }
    if (truncates && defer) {
      // the caller works such rows off in a second, dense pass (u, ts, ierr are left untouched here)
      *defer = true;
      return;
    }
    if (truncates) {
      T S[KM * (KM + 1) / 2], V[KM][KM], lam[KM];
      sfor<KM*(KM + 1) / 2>([&](auto e) ABRK_LAMBDA { S[e()] = Am[e()]; });
      // x,y,z in fp64: the direct 3 x 3 decomposition (no sweeps, ~250 instructions).  The rows that really truncate are
      // < 0.01 % of random states, but each runs on a lone lane while its wavefront waits: with the Jacobi sweeps such a
      // wavefront lived 4.4 us instead of 2.5 (UR5, 4096 rows) and 9.2 instead of <= 7.9 us in the 131072-row shard of
      // BASELINE config 4, where the ~13 of them ARE the kernel's tail (profiles/round5/shard_step_timeline.md).
      // (No row is masked in the FAST forms, so the order sym3_eig leaves the pairs in does not matter.)
      if constexpr (FAST && KM == 3 && sizeof(T) == 8) sym3_eig(S, V, lam);
      else jacobi_eig<KM>(S, V, lam);
      T smax = T(0);
      sfor<KM>([&](auto r) ABRK_LAMBDA {
        // a masked row r is an isolated unit diagonal: Jacobi never rotates it, so eigenpair r
        // stays (1, e_r) and belongs to no controlled DOF - drop it
        bool mine = sel[r()];
        lam[r()] = mine ? lam[r()] : T(0);
        smax = Rm<T>::fmax(smax, Rm<T>::fabs(lam[r()]));
      });
      T cut = rcond * smax;
      T wv[KM];
      sfor<KM>([&](auto r) ABRK_LAMBDA { wv[r()] = (Rm<T>::fabs(lam[r()]) > cut) ? rcp(lam[r()]) : T(0); });
      sfor<KM>([&](auto a) ABRK_LAMBDA {
        sfor<a() + 1>([&](auto b) ABRK_LAMBDA {
          T acc = T(-0.0);
          sfor<KM>([&](auto r) ABRK_LAMBDA { acc += V[a()][r()] * V[b()][r()] * wv[r()]; });
          Mx[tri(a(), b())] = acc;
        });
      });
      mx_explicit = true;
    }
  }

  ABRK_MARK("law3:forces");
  // desired task-space forces (osc.py:250-259)
  T ut[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  if (FAST || P.pos_on) sfor<3>([&](auto r) ABRK_LAMBDA { ut[r()] = p[r()] - tgt[r()]; });
  if constexpr (!FAST) {
    if (P.ori_on) {
      T abg[3] = {tgt[3], tgt[4], tgt[5]}, uo[3];
      orientation_forces(P.alg, RF, abg, uo);
      sfor<3>([&](auto r) ABRK_LAMBDA { ut[3 + r()] = uo[r()]; });
    }
  }
  // integral term (osc.py:262-264)
  if (FEAT >= 2 && have_ierr) {
    sfor<6>([&](auto r) ABRK_LAMBDA {
      ierr[r()] += ut[r()];
      ut[r()] += P.ki * ierr[r()];
    });
  }
  // gains / velocity limiting (osc.py:266-272, 198-215; constants osc.py:89-115)
  if (P.use_vmax) {
    T sat_xyz = P.sat_xyz, sat_abg = P.sat_abg;
    T nx = Rm<T>::sqrt(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]);
    T na = Rm<T>::sqrt(ut[3] * ut[3] + ut[4] * ut[4] + ut[5] * ut[5]);
    T sx = (nx > sat_xyz) ? sat_xyz / nx : T(1);
    T sa = (na > sat_abg) ? sat_abg / na : T(1);
    T lx = P.lamb_xyz, la = P.lamb_abg;
    sfor<3>([&](auto r) ABRK_LAMBDA {
      ut[r()] = P.kv * sx * lx * ut[r()];
      ut[3 + r()] = P.kv * sa * la * ut[3 + r()];
    });
  } else {
    sfor<3>([&](auto r) ABRK_LAMBDA {
      ut[r()] *= P.kp;
      ut[3 + r()] *= P.ko;
    });
  }
  ABRK_MARK("law3:vel_comp");
  // velocity compensation (osc.py:274-282)
  bool tv_zero = true;
  if (FEAT >= 2 && tv_given) sfor<6>([&](auto r) ABRK_LAMBDA { tv_zero = tv_zero && (tvin[r()] == T(0)); });
  T Mdq[N];
  symv<N>(Ms, dq, Mdq);
  if (tv_zero) {
    sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = T(-1) * P.kv * Mdq[i()]; });
  } else {
    sfor<6>([&](auto r) ABRK_LAMBDA {
      T dx = T(-0.0);
      if constexpr (r() < KM) {
        sfor<N>([&](auto i) ABRK_LAMBDA { dx += Jr[i()][r()] * dq[i()]; });
      }
      ut[r()] += P.kv * (dx - tvin[r()]);
    });
    sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = T(0); });
  }
  ABRK_MARK("law3:f_JTf");
  // u -= J^T (Mx u_task[ctrlr_dof]) (osc.py:285-288)
  T uts[KM], f[KM];
  sfor<KM>([&](auto r) ABRK_LAMBDA { uts[r()] = sel[r()] ? ut[r()] : T(0); });
  if constexpr (kClosed) {
    symv<KM>(Mx, uts, f);
  } else if (mx_explicit) {
    symv<KM>(Mx, uts, f);
  } else {
    T y[KM];
    chol_fwd<KM>(LA, ila, uts, y);
    chol_bwd<KM>(LA, ila, y, f);
  }
  sfor<N>([&](auto i) ABRK_LAMBDA {
    T acc = T(-0.0);
    sfor<KM>([&](auto r) ABRK_LAMBDA { acc += Jr[i()][r()] * f[r()]; });
    u[i()] -= acc;
  });
  if constexpr (USE_C) sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] -= cvec[i()]; });  // osc.py:291-292
  sfor<N>([&](auto i) ABRK_LAMBDA { ts[i()] = u[i()]; });                          // osc.py:297
  // osc.py:300-301.  use_g is uniform over the launch: a zero gain (a scalar select) instead of the N double selects
  // the compiler makes of `if (P.use_g)` - u + 0 g is u
  const T gsc = P.use_g ? gscale : T(0);
  sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = Rm<T>::fma(gsc, gz[i()], u[i()]); });

  ABRK_MARK("law3:null");
  // secondary controllers through the null-space filter I - J^T Jbar^T (osc.py:310-318).
  // With u_null = M v the filtered signal is M v - J^T Mx (J v) (Jbar^T M = Mx J).
  if (FEAT >= 1 && (P.n_null > 0 || (FEAT >= 2 && have_ext))) {
    T v[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { v[i()] = T(0); });
    for (int c = 0; c < P.n_null; c++) null_command<N>(P.nul[c], q, dq, v);
    T un[N];
    // Damping only (BASELINE config 3): v = -kv dq, so M v = -kv (M dq) - and M dq is at hand (wave-uniform branch)
    bool damping_only = P.n_null > 0;
    T kvs = T(0);
    for (int c = 0; c < P.n_null; c++) {
      damping_only = damping_only && (P.nul[c].kind == 1);
      kvs += P.nul[c].kv;
    }
    if (damping_only) sfor<N>([&](auto i) ABRK_LAMBDA { un[i()] = -kvs * Mdq[i()]; });
    else symv<N>(Ms, v, un);
    if (FEAT >= 2 && have_ext) {
      // caller-evaluated u_null: v_ext = M^-1 u_ext
      T y[N], w[N];
      chol_fwd<N>(L, il, une, y);
      chol_bwd<N>(L, il, y, w);
      sfor<N>([&](auto i) ABRK_LAMBDA {
        v[i()] += w[i()];
        un[i()] += une[i()];
      });
    }
    T jv[KM], f2[KM];
    sfor<KM>([&](auto r) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { acc += Jr[i()][r()] * v[i()]; });
      jv[r()] = acc;
    });
    symv<KM>(Mx, jv, f2);
    sfor<N>([&](auto i) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<KM>([&](auto r) ABRK_LAMBDA { acc += Jr[i()][r()] * f2[r()]; });
      u[i()] += un[i()] - acc;
    });
  }
}

// LEN values (LEN even) to a 2-element-aligned address as two-element pieces (16-byte stores in fp64)
template <int LEN, class T>
ABRK_INL void store_pairs(T* dst, const T (&v)[LEN]) {
  static_assert(LEN % 2 == 0, "pairs");
  typedef T V2 __attribute__((ext_vector_type(2)));
  V2* d2 = reinterpret_cast<V2*>(dst);
  sfor<LEN / 2>([&](auto k) ABRK_LAMBDA { d2[k()] = V2{v[2 * k()], v[2 * k() + 1]}; });
}
// weights of numpy.linalg.pinv(Mx_inv, rcond) on the eigenvalues (osc.py:145): 1 / lam where |lam| > rcond max|lam|
template <int K, class T>
ABRK_INL void pinv_weights(const T (&lam)[K], T rcond, T (&wv)[K]) {
#pragma clang fp contract(off)
  T smax = T(0);
  sfor<K>([&](auto r) ABRK_LAMBDA { smax = Rm<T>::fmax(smax, Rm<T>::fabs(lam[r()])); });
  const T cut = rcond * smax;
  sfor<K>([&](auto r) ABRK_LAMBDA {
    const bool keep = Rm<T>::fabs(lam[r()]) > cut;
    wv[r()] = keep ? rcp(keep ? lam[r()] : T(1)) : T(0);
  });
}
// ---- second pass of the six-row law on a hand-over record (the first pass left it: osc_law6's deferral branch).
// With Mx_inv = Z diag(lam) Z^T the pseudo-inverse is Z W Z^T, and what the law needs of it are J^T Mx u_task and
// J^T Mx (J v): with G = Z^T [J | u_task | J v] - the SAME orthogonal transformation applied to every column -
//   (J^T Mx x)_c = sum_i G[i][c] W_i G[i][x],
// so Z is never formed: the solver (ql_core) carries the columns along, one per lane in the wave-cooperative kernel
// (abrk_law.hip osc6_finish_kernel: NV = 1, all lanes of the wavefront decompose the same matrix, each transforms its
// own column; lanes N / N + 1 hand their transformed column to the others at the end) or all of them on one lane (the
// host check build: NV = N + 2).  `col`: index of the first column held.  -> a1[k] = (J^T Mx u_task)_(col+k),
// a2[k] = (J^T Mx J v)_(col+k) given the transformed u_task / J v columns gu, gw.
// (in two steps, so that a caller can ask for a record before it knows that there is one: osc6_finish_kernel)
template <int N, class T, int NV>
ABRK_INL void osc6_rec_load(const T* __restrict__ rec, int col, T (&S)[21], T (&G)[NV][6]) {
  constexpr int XS = rec_xs(N);
  sfor<21>([&](auto e) ABRK_LAMBDA { S[e()] = rec[e()]; });
  sfor<NV>([&](auto k) ABRK_LAMBDA {
    sfor<6>([&](auto r) ABRK_LAMBDA { G[k()][r()] = rec[rec_off_x() + r() * XS + col + k()]; });
  });
}
// -> the factor of the truncating pseudo-inverse in the transformed basis (ql_pinv_factor): y = pinv x is
// ql_pinv_solve(li, iq, x, y) for any transformed vector x = a column of G
template <int N, class T, int NV, bool UNI>
ABRK_INL void osc6_rec_solve(const T* __restrict__ rec, int col, T (&S)[21], T (&G)[NV][6], T (&li)[6], T (&iq)[6]) {
  constexpr int XS = rec_xs(N);
  T lam[6];
  QlTail<6, T> tail;
  tail.rcond = T(1e-3) * T(0.1);
  if (!ql_core<6, T, NV, false, UNI, true>(S, G, lam, &tail)) {
    if constexpr (UNI) {  // (cold: a rotation radius underflowed - the predicated form has tql2's recovery path)
      sfor<NV>([&](auto k) ABRK_LAMBDA {
        sfor<6>([&](auto r) ABRK_LAMBDA { G[k()][r()] = rec[rec_off_x() + r() * XS + col + k()]; });
      });
      ql_core<6, T, NV, false, false, true>(S, G, lam, &tail);
    }
  }
  ql_pinv_factor<6>(lam, tail, li, iq);
}
// The truncating pseudo-inverse and the tail of the six-row law on ONE lane, from the values a hand-over record holds:
// S = Mx_inv (masked rows as isolated zeros), G[c] = column c of [J | u_task | J v], the two joint-space sums b1, b2.
// ONE routine for every place a row can meet it - the finish kernel's one-record-per-lane form (from a record), the
// complete row program (from its registers: batches below one wavefront of rows, the recompute pass of large batches,
// sharded and fused-output calls) - so that a row's bits do not depend on the batch it arrives in; the finish
// kernel's wave-cooperative form is the same arithmetic with the columns spread over lanes (contraction pinned off).
template <int N, class T>
ABRK_INL void osc6_tail(const T (&S)[21], T (&G)[N + 2][6], const T (&b1)[N], const T (&b2)[N], bool nulls, T (&u)[N],
                        T (&ts)[N]) {
#pragma clang fp contract(off)  // the same bits as the wave-cooperative form (abrk_kernels.h osc6_finish_kernel)
  T lam[6], li[6], iq[6], yu[6], yw[6];
  QlTail<6, T> tail;
  tail.rcond = T(1e-3) * T(0.1);
  ql_core<6, T, N + 2, false, false, true>(S, G, lam, &tail);
  ql_pinv_factor<6>(lam, tail, li, iq);
  ql_pinv_solve<6>(li, iq, G[N], yu);      // pinv (Z^T u_task)
  ql_pinv_solve<6>(li, iq, G[N + 1], yw);  // pinv (Z^T J v)
  sfor<N>([&](auto c) ABRK_LAMBDA {
    T a1 = T(-0.0), a2 = T(-0.0);
    sfor<6>([&](auto i) ABRK_LAMBDA {
      a1 = Rm<T>::fma(G[c()][i()], yu[i()], a1);
      a2 = Rm<T>::fma(G[c()][i()], yw[i()], a2);
    });
    ts[c()] = b1[c()] - a1;
    u[c()] = ts[c()] + b2[c()] - (nulls ? a2 : T(0));
  });
}
// the whole second pass of one row on one lane, from its record
template <int N, class T>
ABRK_INL void osc6_finish_row(const T* __restrict__ rec, bool nulls, T (&u)[N], T (&ts)[N]) {
  T S[21], G[N + 2][6], b1[N], b2[N];
  osc6_rec_load<N, T, N + 2>(rec, 0, S, G);
  sfor<N>([&](auto c) ABRK_LAMBDA {
    b1[c()] = rec[rec_off_b1(N) + c()];
    b2[c()] = rec[rec_off_b1(N) + N + c()];
  });
  osc6_tail<N, T>(S, G, b1, b2, nulls, u, ts);
}
// What a deferring row hands over besides Mx_inv and its Jacobian rows - (J v)[r] of the secondary controllers and the two
// joint-space sums around J^T f - with contraction pinned off: the first pass (into the record) and the complete row
// program (into registers) are different instantiations and must arrive at the same bits.
template <int N, class T>
ABRK_INL T osc6_jv(const T (&row)[N], const T (&v)[N]) {
#pragma clang fp contract(off)
  T jv = T(-0.0);
  sfor<N>([&](auto i) ABRK_LAMBDA { jv += row[i()] * v[i()]; });
  return jv;
}
template <int N, bool USE_C, bool NOTS, class T, int NU>
ABRK_INL void osc6_sums(const OscP<T>& P, T gscale, const T (&gz)[N], const T (&u0)[N], const T (&cvec)[N], bool nulls,
                        const T (&un)[NU], T (&b1)[N], T (&b2)[N]) {
#pragma clang fp contract(off)
  const T gsc = (!NOTS && P.use_g) ? gscale : T(0);
  sfor<N>([&](auto i) ABRK_LAMBDA {
    b1[i()] = USE_C ? u0[i()] - cvec[i()] : u0[i()];
    b2[i()] = gsc * gz[i()];
  });
  if constexpr (NU == N) {
    if (nulls) sfor<N>([&](auto i) ABRK_LAMBDA { b2[i()] += un[i()]; });
  }
}

// ---- the same law for all six task rows (any ctrlr_dof, ref_frame, orientation control), restructured around its
// register peak.  osc_law above keeps Jr (6 N values), Y = L^-1 J^T (6 N), M, its factor and two 6 x 6 factors alive
// together: 390-490 registers, ONE wave per SIMD.  Here
//   * the six (masked) rows of the task Jacobian sit in `js` - the wavefront's LDS slab on the GPU (LdsScratch), plain
//     arrays in the host check build and the law-only kernel - and are read back a row at a time: for Y, for J^T f,
//     for J v and J dq;
//   * everything that only needs the kinematic outputs (the task-space forces incl. the orientation error, M dq, the
//     secondary controllers' M v) is reduced to its N- or 6-vector BEFORE the factorisations, so p, RF, the target and M
//     are dead by then;
//   * Y is held three rows at a time: Mx_inv = Y Y^T in two row blocks of three (+3 forward solves, -18 live values).
// Same arithmetic per entry as osc_law (Gram form of Mx_inv, Cholesky, the two certificates, Jacobi behind them);
// only the order of independent steps differs.  Two waves per SIMD on the six-joint arms.
constexpr int kLaw6Yb = 3;  // rows of Y held at a time (2 was measured: 12 instead of 9 solves cost more than 16 B of scratch)
// ... in the first pass (compiled without the eigen-decomposition: 202 - 210 registers on the UR5, room to spare under the
// 256 of two waves per SIMD) all six rows at once: 6 forward solves instead of 9, 224 - 246 registers, still no scratch.
// Same box, 8 M UR5 rows: 762 / 763 us against 786 / 794 us with three rows (and 815 / 817 with two); the 4096-row step
// 16.9 - 17.0 against 17.1 - 17.4 us.
constexpr int kLaw6YbFirst = 6;
// rows of the task Jacobian read from the row store one AHEAD of their use (first pass: Y and J^T f).  Same box, three
// interleaved repetitions: the 4096-row step 16.65 / 16.71 / 16.75 us against 17.09 / 16.94 / 16.95; 8 M rows 766.9 / 767.1 /
// 770.8 against 769.9 / 769.8 / 771.7 us; no register more (224 - 248).
template <int N, class T, bool USE_C, int FEAT, class Rows, int QSTEPS = 3>
ABRK_INL void osc_law6(const OscP<T>& P, const T (&Ms)[N * (N + 1) / 2], const T (&gz)[N], T gscale,
                       const T (&cvec)[N], Rows& js, const T (&p)[3], const T (&RF)[9], const T (&q)[N],
                       const T (&dq)[N], const T (&tgt)[6], bool tv_given, const T (&tvin)[6], bool have_ierr,
                       T (&ierr)[6], bool have_ext, const T (&une)[N], T (&u)[N], T (&ts)[N], bool* defer = nullptr,
                       bool* singular = nullptr) {
  constexpr int KM = 6;
  static_assert(rec_len(N) >= rec_off_b1(N) + 2 * N, "hand-over record layout");
  // ctrlr_dof is uniform over the launch.  With all six rows selected (the reference benchmark's UR5 setting) nothing
  // is masked; the masked forms sit in blocks of their own behind scalar branches (ABRK_UNIFORM_BLOCK) instead of
  // costing every launch two v_cndmask per `sel ? x : y`.
  bool sel[KM], all_sel = true;
  sfor<KM>([&](auto r) ABRK_LAMBDA {
    sel[r()] = P.dof[r()] != 0;
    all_sel = all_sel && sel[r()];
  });

  ABRK_MARK("law6:task_forces");
  // desired task-space forces (osc.py:250-259)
  T ut[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  if (P.pos_on) sfor<3>([&](auto r) ABRK_LAMBDA { ut[r()] = p[r()] - tgt[r()]; });
  if (P.ori_on) {
    T abg[3] = {tgt[3], tgt[4], tgt[5]}, uo[3];
    // (the target's Euler angles: through the kernel's sin / cos table in LDS where there is one - 16 + 3 instructions
    //  per angle against 25 + 10 of the polynomial routine)
    if constexpr (Rows::kHasTab) orientation_forces<T, QSTEPS, true>(P.alg, RF, abg, uo, js.sctab);
    else orientation_forces<T, QSTEPS>(P.alg, RF, abg, uo);
    sfor<3>([&](auto r) ABRK_LAMBDA { ut[3 + r()] = uo[r()]; });
  }
  // integral term (osc.py:262-264).  (A deferred row returns below without its state being stored.)
  if (FEAT >= 2 && have_ierr) {
    sfor<6>([&](auto r) ABRK_LAMBDA {
      ierr[r()] += ut[r()];
      ut[r()] += P.ki * ierr[r()];
    });
  }
  // gains / velocity limiting (osc.py:266-272, 198-215; constants osc.py:89-115)
  if (P.use_vmax) {
    T sat_xyz = P.sat_xyz, sat_abg = P.sat_abg;
    T nx = Rm<T>::sqrt(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]);
    T na = Rm<T>::sqrt(ut[3] * ut[3] + ut[4] * ut[4] + ut[5] * ut[5]);
    T sx = (nx > sat_xyz) ? sat_xyz / nx : T(1);
    T sa = (na > sat_abg) ? sat_abg / na : T(1);
    T lx = P.lamb_xyz, la = P.lamb_abg;
    sfor<3>([&](auto r) ABRK_LAMBDA {
      ut[r()] = P.kv * sx * lx * ut[r()];
      ut[3 + r()] = P.kv * sa * la * ut[3 + r()];
    });
  } else {
    sfor<3>([&](auto r) ABRK_LAMBDA {
      ut[r()] *= P.kp;
      ut[3 + r()] *= P.ko;
    });
  }
  ABRK_MARK("law6:vel_comp");
  // velocity compensation (osc.py:274-282)
  bool tv_zero = true;
  if (FEAT >= 2 && tv_given) sfor<6>([&](auto r) ABRK_LAMBDA { tv_zero = tv_zero && (tvin[r()] == T(0)); });
  T Mdq[N], u0[N];
  symv<N>(Ms, dq, Mdq);
  if (tv_zero) {
    sfor<N>([&](auto i) ABRK_LAMBDA { u0[i()] = T(-1) * P.kv * Mdq[i()]; });
  } else {
    sfor<6>([&](auto r) ABRK_LAMBDA {
      T row[N];
      js.get_row(r, row);
      T dx = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { dx += row[i()] * dq[i()]; });
      ut[r()] += P.kv * (dx - tvin[r()]);
    });
    sfor<N>([&](auto i) ABRK_LAMBDA { u0[i()] = T(0); });
  }
  T uts[KM];
  sfor<KM>([&](auto r) ABRK_LAMBDA { uts[r()] = ut[r()]; });
  if (!all_sel) {  // (a select, not a factor: an unselected target entry may be anything - osc.py:285 indexes it away)
    ABRK_UNIFORM_BLOCK();
    sfor<KM>([&](auto r) ABRK_LAMBDA { uts[r()] = sel[r()] ? ut[r()] : T(0); });
  }
  // no training signal wanted (osc.py:297 is the only reader of u before gravity): u0 takes the gravity term now
  if constexpr (Rows::kNoTs) {
    const T gsc = P.use_g ? gscale : T(0);  // uniform: a zero gain instead of N double selects
    sfor<N>([&](auto i) ABRK_LAMBDA { u0[i()] = Rm<T>::fma(gsc, gz[i()], u0[i()]); });
  }

  // secondary controllers (osc.py:310-318): v with u_null = M v, and un = M v while M is at hand
  const bool nulls = FEAT >= 1 && (P.n_null > 0 || (FEAT >= 2 && have_ext));
  T v[FEAT >= 1 ? N : 1], un[FEAT >= 1 ? N : 1];
  if constexpr (FEAT >= 1) {
    if (nulls) {
      sfor<N>([&](auto i) ABRK_LAMBDA { v[i()] = T(0); });
      for (int c = 0; c < P.n_null; c++) null_command<N>(P.nul[c], q, dq, v);
      bool damping_only = P.n_null > 0;
      T kvs = T(0);
      for (int c = 0; c < P.n_null; c++) {
        damping_only = damping_only && (P.nul[c].kind == 1);
        kvs += P.nul[c].kv;
      }
      if (damping_only) sfor<N>([&](auto i) ABRK_LAMBDA { un[i()] = -kvs * Mdq[i()]; });
      else symv<N>(Ms, v, un);
    }
  }

  ABRK_MARK("law6:chol_M");
  ABRK_STAMP(js, 8, false);
  // _Mx (osc.py:120-147): Mx_inv = J M^-1 J^T = Y Y^T, Y = J L^-T (rows y_r = L^-1 j_r)
  T L[N * (N + 1) / 2], il[N];
  T minpiv = T(1);
  chol<N, T, false>(Ms, L, il, &minpiv);
  // (the kernels take the verdict along and store the flag after the row's outputs: a store behind a branch HERE
  //  splits the law's one scheduling region - +14 registers on Jaco2's x,y,z kernels, 44 B of scratch and 8 % on
  //  BASELINE config 3.  And they take it along as a WAVE-level scalar - "some row of this wavefront": the flag says no
  //  more than that anyway - because a per-lane bool rides in vector registers: +10 .. 12 on the six-row kernels)
  if (singular) *singular = any_lane(!(minpiv > T(0)));
  else flag_singular(P, minpiv);
  if constexpr (FEAT >= 2) {
    if (have_ext) {  // caller-evaluated u_null: v_ext = M^-1 u_ext
      T y[N], w[N];
      chol_fwd<N>(L, il, une, y);
      chol_bwd<N>(L, il, y, w);
      sfor<N>([&](auto i) ABRK_LAMBDA {
        v[i()] += w[i()];
        un[i()] += une[i()];
      });
    }
  }
  ABRK_MARK("law6:Y_Am");
  ABRK_STAMP(js, 9, false);
  T Am[KM * (KM + 1) / 2];
  auto yrow = [&](auto r, T(&y)[N]) ABRK_LAMBDA {
    T b[N];
    js.get_row(r, b);
    chol_fwd<N>(L, il, b, y);
  };
  auto ydot = [&](const T(&a)[N], const T(&b)[N]) ABRK_LAMBDA -> T {
    T acc = T(-0.0);
    sfor<N>([&](auto i) ABRK_LAMBDA { acc += a[i()] * b[i()]; });
    return acc;
  };
  {
    // Y is held YB rows at a time: the diagonal block of Mx_inv from the rows at hand, the blocks below it from one
    // more row at a time (recomputed when its own block comes up).  Three rows per block cost 9 forward solves for the
    // six rows, two rows 12 (measured with the training signal among the outputs, where u0 AND the gravity sums stay
    // live across the law: 957 us against 937 us at 8 M UR5 rows - the 16 B of scratch it saves do not pay for 3 solves)
    constexpr int YB = Rows::kDeferOnly ? kLaw6YbFirst : kLaw6Yb;
    static_assert(KM % YB == 0, "block size divides the six rows");
    sfor<KM / YB>([&](auto bi) ABRK_LAMBDA {
      constexpr int r0 = bi() * YB;
      T Ya[YB][N];
      if constexpr (YB == KM) {
        // the row store is read one row AHEAD of the forward solve that consumes it: at small batches one wavefront
        // per SIMD has nothing else to hide the LDS round trip behind (12 more registers: the first pass has them)
        T rb[2][N];
        js.get_row(ic<0>{}, rb[0]);
        sfor<YB>([&](auto r) ABRK_LAMBDA {
          if constexpr (r() + 1 < YB) js.get_row(ic<r() + 1>{}, rb[(r() + 1) & 1]);
          chol_fwd<N>(L, il, rb[r() & 1], Ya[r()]);
        });
      } else {
        sfor<YB>([&](auto r) ABRK_LAMBDA { yrow(ic<r0 + r()>{}, Ya[r()]); });
      }
      sfor<YB>([&](auto r) ABRK_LAMBDA {
        sfor<r() + 1>([&](auto c) ABRK_LAMBDA { Am[tri(r0 + r(), r0 + c())] = ydot(Ya[r()], Ya[c()]); });
      });
      sfor<KM - r0 - YB>([&](auto rr) ABRK_LAMBDA {
        constexpr int r = r0 + YB + rr();
        T yb[N];
        yrow(ic<r>{}, yb);
        sfor<YB>([&](auto c) ABRK_LAMBDA { Am[tri(r, r0 + c())] = ydot(yb, Ya[c()]); });
      });
    });
  }
  ABRK_STAMP(js, 10, false);
  T trace = T(0);
  // a masked row of J is zero: so is its diagonal entry, which becomes the unit diagonal.  (Selects on purpose: a block
  // behind a scalar branch at this spot - the register peak of the law - makes the allocator spill 70 values.)
  sfor<KM>([&](auto r) ABRK_LAMBDA {
    trace += Am[tri(r(), r())];
    Am[tri(r(), r())] = sel[r()] ? Am[tri(r(), r())] : T(1);
  });
  T LA[KM * (KM + 1) / 2], ila[KM], Mx[KM * (KM + 1) / 2];
  // unsanitised factor: LA / ila are only read where okA holds (every use below sits behind it or behind `truncates`)
  bool okA = chol<KM, T, false>(Am, LA, ila);
  T det = T(1);
  sfor<KM>([&](auto r) ABRK_LAMBDA { det *= LA[tri(r(), r())] * LA[tri(r(), r())]; });
  bool mx_explicit = FEAT != 0;
  T f[KM], f2[FEAT >= 1 ? KM : 1];
  // (only where the factor exists: after a non-positive pivot LA / ila hold non-finite values, which -ffinite-math-only
  //  makes unspecified to compute with; every reader of Mx below sits behind okA or overwrites it)
  if constexpr (FEAT != 0) {
    if (okA) chol_inverse<KM>(LA, ila, Mx);
  }
  const T thr = T(1e-3), rcond = T(1e-3) * T(0.1);
  if (!(okA && det >= thr)) {
    // pinv branch (osc.py:142-145); the two certificates of osc_law
    T bound = rcond;
    if (all_sel) {
      sfor<KM>([&](auto r) ABRK_LAMBDA { bound *= trace; });
    } else {
      ABRK_UNIFORM_BLOCK();
      sfor<KM>([&](auto r) ABRK_LAMBDA { bound *= sel[r()] ? trace : T(1); });
    }
    bool truncates = !(okA && det > bound);
    if (truncates && okA) {
      // second certificate: trace(A) trace(A^-1) < 1 / rcond.  Without secondary controllers the inverse itself is
      // not needed (f comes from the factor): trace(A^-1) = |L^-1|_F^2 over the selected columns, half the work
      T tinv = T(0);
      auto add_diag = [&](auto r, T cs) ABRK_LAMBDA {  // + (A^-1)_rr of a selected row
        if (all_sel) {
          tinv += cs;
        } else {
          ABRK_UNIFORM_BLOCK();
          tinv += sel[r()] ? cs : T(0);
        }
      };
      if constexpr (FEAT == 0) {
        T Li[KM * (KM + 1) / 2];
        chol_factor_inverse<KM>(LA, ila, Li);
        sfor<KM>([&](auto r) ABRK_LAMBDA {
          T cs = T(-0.0);
          sfor<KM - r()>([&](auto kk) ABRK_LAMBDA { cs = Rm<T>::fma(Li[tri(r() + kk(), r())], Li[tri(r() + kk(), r())], cs); });
          add_diag(r, cs);
        });
      } else {
        sfor<KM>([&](auto r) ABRK_LAMBDA { add_diag(r, Mx[tri(r(), r())]); });
      }
      if (trace * tinv * rcond < T(1)) {
        truncates = false;
        if constexpr (FEAT != 0) mx_explicit = true;
      }
    }
    ABRK_STAMP(js, 11, false);
    if (truncates && (Rows::kDeferOnly || defer)) {
      if (defer) *defer = true;  // worked off in the second pass; u / the training signal of this row are not written yet
      // the row joins its wavefront's sub-list; in hand-over mode it also leaves everything the second pass needs -
      // Mx_inv, the task Jacobian rows, u_task and the two joint-space sums around J^T f - so that the pass runs the
      // eigen-decomposition and nothing else (abrk_device.h rec_*; osc6_finish below reads it back)
      if (js.claim()) {
        T* rec = js.template record<T>(rec_len(N));
        constexpr int XS = rec_xs(N);
        T sv[22];
        sfor<21>([&](auto e) ABRK_LAMBDA { sv[e()] = Am[e()]; });
        sfor<KM>([&](auto r) ABRK_LAMBDA { sv[tri(r(), r())] = sel[r()] ? Am[tri(r(), r())] : T(0); });
        sv[21] = T(js.row);  // the row's index: the records of a chunk are packed (ScratchBase::record)
        store_pairs<22>(rec, sv);
        // (the rows come back from the row store one ahead of the stores that copy them: few lanes are in this branch,
        //  but their wavefront waits for every LDS round trip in full)
        T rb[2][N];
        js.get_row(ic<0>{}, rb[0]);
        sfor<KM>([&](auto r) ABRK_LAMBDA {
          T xr[XS];
          if constexpr (r() + 1 < KM) js.get_row(ic<r() + 1>{}, rb[(r() + 1) & 1]);
          const T(&row)[N] = rb[r() & 1];
          T jv = T(-0.0);
          sfor<N>([&](auto i) ABRK_LAMBDA { xr[i()] = row[i()]; });
          if constexpr (FEAT >= 1) {
            if (nulls) jv = osc6_jv<N>(row, v);
          }
          xr[N] = uts[r()];
          xr[N + 1] = jv;
          if constexpr (XS > N + 2) xr[XS - 1] = T(0);
          store_pairs<XS>(rec + rec_off_x() + r() * XS, xr);
        });
        T bb[2 * N], b1[N], b2[N];
        osc6_sums<N, USE_C, Rows::kNoTs>(P, gscale, gz, u0, cvec, nulls, un, b1, b2);
        sfor<N>([&](auto i) ABRK_LAMBDA {
          bb[i()] = b1[i()];
          bb[N + i()] = b2[i()];
        });
        store_pairs<2 * N>(rec + rec_off_b1(N), bb);
      }
      ABRK_STAMP(js, 12, false);
      return;
    }
    if constexpr (!Rows::kDeferOnly) if (truncates) {
      // the truncating pseudo-inverse, here and now: exactly what a hand-over record would carry, into osc6_tail - the
      // routine the finish kernel runs on records - so that this row's bits are those of every other form of the law.
      // (A masked row is an isolated diagonal entry: with 0 there its eigenpair is (0, e_r) - below every cut-off, so it
      //  drops out of the pseudo-inverse whatever position the solver leaves it in.)
      T S[KM * (KM + 1) / 2], G[N + 2][6], b1[N], b2[N];
      sfor<KM*(KM + 1) / 2>([&](auto e) ABRK_LAMBDA { S[e()] = Am[e()]; });
      sfor<KM>([&](auto r) ABRK_LAMBDA { S[tri(r(), r())] = sel[r()] ? Am[tri(r(), r())] : T(0); });
      sfor<KM>([&](auto r) ABRK_LAMBDA {
        T row[N];
        js.get_row(r, row);
        T jv = T(-0.0);
        if constexpr (FEAT >= 1) {
          if (nulls) jv = osc6_jv<N>(row, v);
        }
        sfor<N>([&](auto i) ABRK_LAMBDA { G[i()][r()] = row[i()]; });
        G[N][r()] = uts[r()];
        G[N + 1][r()] = jv;
      });
      osc6_sums<N, USE_C, Rows::kNoTs>(P, gscale, gz, u0, cvec, nulls, un, b1, b2);
      osc6_tail<N, T>(S, G, b1, b2, nulls, u, ts);
      return;
    }
  }

  ABRK_MARK("law6:f");
  ABRK_STAMP(js, 13, false);
  // f = Mx u_task[ctrlr_dof] (osc.py:285-288); f2 = Mx (J v) for the null-space filter
  if (mx_explicit) {
    symv<KM>(Mx, uts, f);
  } else {
    T y[KM];
    chol_fwd<KM>(LA, ila, uts, y);
    chol_bwd<KM>(LA, ila, y, f);
  }
  if constexpr (FEAT >= 1) {
    if (nulls) {
      T jv[KM];
      sfor<KM>([&](auto r) ABRK_LAMBDA {
        T row[N];
        js.get_row(r, row);
        T acc = T(-0.0);
        sfor<N>([&](auto i) ABRK_LAMBDA { acc += row[i()] * v[i()]; });
        jv[r()] = acc;
      });
      symv<KM>(Mx, jv, f2);
    }
  }
  ABRK_MARK("law6:JTf");
  // u = u0 - J^T f [- C dq]; training signal; + g; + (I - J^T Jbar^T) u_null = M v - J^T Mx (J v)
  T a1[N], a2[FEAT >= 1 ? N : 1];
  sfor<N>([&](auto i) ABRK_LAMBDA { a1[i()] = T(-0.0); });
  if constexpr (FEAT >= 1) sfor<N>([&](auto i) ABRK_LAMBDA { a2[i()] = T(-0.0); });
  if constexpr (Rows::kDeferOnly) {
    T rb[2][N];
    js.get_row(ic<0>{}, rb[0]);
    sfor<KM>([&](auto r) ABRK_LAMBDA {
      if constexpr (r() + 1 < KM) js.get_row(ic<r() + 1>{}, rb[(r() + 1) & 1]);
      sfor<N>([&](auto i) ABRK_LAMBDA { a1[i()] += rb[r() & 1][i()] * f[r()]; });
      if constexpr (FEAT >= 1) {
        if (nulls) sfor<N>([&](auto i) ABRK_LAMBDA { a2[i()] += rb[r() & 1][i()] * f2[r()]; });
      }
    });
  } else {
  sfor<KM>([&](auto r) ABRK_LAMBDA {
    T row[N];
    js.get_row(r, row);
    sfor<N>([&](auto i) ABRK_LAMBDA { a1[i()] += row[i()] * f[r()]; });
    if constexpr (FEAT >= 1) {
      if (nulls) sfor<N>([&](auto i) ABRK_LAMBDA { a2[i()] += row[i()] * f2[r()]; });
    }
  });
  }
  ABRK_STAMP(js, 14, false);
  sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = u0[i()] - a1[i()]; });
  if constexpr (USE_C) sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] -= cvec[i()]; });  // osc.py:291-292
  sfor<N>([&](auto i) ABRK_LAMBDA { ts[i()] = u[i()]; });                          // osc.py:297 (kNoTs: not stored)
  if constexpr (!Rows::kNoTs) {
    const T gsc = P.use_g ? gscale : T(0);  // osc.py:300-301 (uniform: a zero gain instead of N double selects)
    sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = Rm<T>::fma(gsc, gz[i()], u[i()]); });
  }
  if constexpr (FEAT >= 1) {
    if (nulls) sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] += un[i()] - a2[i()]; });
  }
}

// `late()` loads the inputs that are not needed by the kinematics (target, velocities, state; dq
// too unless the Coriolis term is on) - it is invoked after the register-pressure peak.
// FEAT selects which optional inputs are compiled in: 0 = none (the plain law: ~70 registers fewer,
// two waves per SIMD), 1 = fused secondary controllers only (Damping / RestingConfig, BASELINE config 3),
// 2 = everything (target velocity, integral state, caller-evaluated null signal).
struct NoEmit {
  template <class... X>
  ABRK_INL void operator()(X&&...) const {}
};
// `emit(p, Jv, Jw, d, jt, m)` sees the task point, its Jacobian columns, the dynamics accumulators (M packed lower, the
// gravity sums, with MAT the Christoffel matrix) and the joint frames the law is about to consume - the fused "u +
// robot_config outputs" kernel stores them from there (SURVEY 8d Mode F).
// MAT: the dynamics pass also assembles the full C(q,dq) (base_config.py:678-727) - Mode F with `C` among the outputs;
// the law's Coriolis vector is then C dq from that matrix instead of the body recursion.
// `emit_pre(d)` runs right after the dynamics pass (MAT: the Christoffel matrix leaves for its output array there, so
// that its N^2 values are not carried through the Jacobian and the law).
template <class A, class T, int KM, bool USE_C, int FEAT, bool MAT = false, class Late, class Scr, class Emit = NoEmit,
          class EmitPre = NoEmit>
ABRK_INL void osc_row(const A& arm, const OscP<T>& P, const T (&q)[A::N], const T (&dq)[A::N], const T (&tgt)[6],
                      bool tv_given, const T (&tvin)[6], bool have_ierr, T (&ierr)[6], bool have_ext,
                      const T (&une)[A::N], T (&u)[A::N], T (&ts)[A::N], Late&& late, Scr& scr,
                      Emit&& emit = Emit{}, EmitPre&& emit_pre = EmitPre{}) {
  constexpr int N = A::N;
  constexpr bool FAST = (KM <= 3);
  // OSC(use_C) on orthogonal chains (the two-pass form): the Coriolis vector rides on the dynamics pass.  The link
  // visitor of the forward kinematics also advances a body recursion (rne_forward_step) and parks each link's wrench
  // in `scr` - LDS on the GPU, so the 12 N registers they would take stay free and the kernel keeps two waves per
  // SIMD -; one backward sweep (rne_backward) then projects the sums onto the joint axes.  (Until round 2 the
  // recursion ran as a pass of its own with a second forward kinematics: +128 instructions per row.)
  constexpr bool TWO_PASS = USE_C && !MAT && A::kOrtho;
  // the frame rotation of an orthogonal chain is a product of exact rotations: one power step in quat_from_R
  constexpr int kQSteps = A::kOrthoFrames ? 1 : 3;
  Joints<A, T> jt;
  Dyn<A, T, MAT ? CMODE_MAT : (USE_C && !TWO_PASS) ? CMODE_VEC : CMODE_NONE> d;
  T cvm[(MAT && USE_C) ? N : 1];  // C dq from the matrix
  T XR[9], xo[3];
  T p[3], RF[9];
  T cv2[TWO_PASS ? N : 1];
  RneState<T> rne;
  int m = N;
  // sin/cos of the joint angles: through the kernel's LDS table where it keeps one (fp64 OSC kernels on the GPU)
  auto sincos_policy = [&]() ABRK_LAMBDA {
    if constexpr (std::remove_reference<Scr>::type::kHasTab) return ScTab{scr.sctab};
    else return ScCompute{};
  };
  auto dynamics_pass = [&](auto& cap_) ABRK_LAMBDA {
    if constexpr (TWO_PASS) {
      rne_init(rne);
      kin_dyn_hook(arm, q, dq, jt, d, XR, xo, cap_, [&](auto L, const T(&pl)[3]) ABRK_LAMBDA {
        ABRK_SCHED_FENCE();  // the link's Jacobian columns (M, g) are retired before the recursion's transients start
        rne_forward_step<L()>(arm, jt, pl, dq, rne, scr);
        ABRK_SCHED_FENCE();
      }, sincos_policy());
      rne_backward(jt, scr, cv2);
      // the result is only consumed at the very end of the law: without this the scheduler reads the slab here and
      // carries (spills) the 6 N values until then
      sfor<N>([&](auto i) ABRK_LAMBDA { opaque(cv2[i()]); });
      ABRK_SCHED_FENCE();
    } else {
      kin_dyn_hook(arm, q, dq, jt, d, XR, xo, cap_, [](auto, const T(&)[3]) ABRK_LAMBDA {}, sincos_policy());
    }
  };
  ABRK_MARK("row:dynamics");
  if constexpr (FAST) {
    NoCap nc;
    dynamics_pass(nc);
    if (P.has_off) {
      T oe[3];
      mulBE<A, T>(arm, XR, xo, RF, oe);
      sfor<3>([&](auto r) ABRK_LAMBDA {
        p[r()] = oe[r()] + RF[r() * 3] * P.off[0] + RF[r() * 3 + 1] * P.off[1] + RF[r() * 3 + 2] * P.off[2];
      });
    } else {
      mulBE_pt<A, T>(arm, XR, xo, p);
    }
  } else if constexpr (std::decay<Scr>::type::kEeFrame) {
    // the end effector's frame, known at compile time: no capture in the chain (ScratchBase::kEeFrame); the same
    // expressions as the capture path below, so the same bits
    NoCap nc;
    dynamics_pass(nc);
    T oe[3];
    mulBE<A, T>(arm, XR, xo, RF, oe);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      p[r()] = oe[r()] + RF[r() * 3] * P.off[0] + RF[r() * 3 + 1] * P.off[1] + RF[r() * 3 + 2] * P.off[2];
    });
    m = N;
  } else {
    FrameCap<T> cap;
    cap.frame = P.ref_frame;
    sfor<9>([&](auto e) ABRK_LAMBDA { cap.R[e()] = T(0); });
    sfor<3>([&](auto r) ABRK_LAMBDA { cap.o[r()] = T(0); });
    dynamics_pass(cap);
    sfor<9>([&](auto e) ABRK_LAMBDA { RF[e()] = cap.R[e()]; });
    sfor<3>([&](auto r) ABRK_LAMBDA {
      p[r()] = cap.o[r()] + RF[r() * 3] * P.off[0] + RF[r() * 3 + 1] * P.off[1] + RF[r() * 3 + 2] * P.off[2];
    });
    m = P.m_joints;
  }
  ABRK_MARK("row:jacobian");
  if constexpr (MAT && USE_C) {
    sfor<N>([&](auto i) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto j) ABRK_LAMBDA { acc += d.Cm[i() * N + j()] * dq[j()]; });
      cvm[i()] = acc;
    });
  }
  emit_pre(d);
  ABRK_SCHED_FENCE();
  // task Jacobian, rows masked (osc.py:242-244)
  if constexpr (FAST) {
    T Jv[N][3], Jw[N][3];
    jacobian(jt, p, m, Jv, Jw);
    emit(p, Jv, Jw, d, jt, m);
    ABRK_SCHED_FENCE();
    late();
    ABRK_STAMP(scr, 3, false);  // (timeline build) kinematics, dynamics, Coriolis sweep and Jacobian done
    if constexpr (MAT && USE_C)
      osc_law<N, T, KM, true, FEAT>(P, d.Ms, d.gz, T(9.81), cvm, Jv, Jw, p, RF, q, dq, tgt, tv_given, tvin, have_ierr,
                                    ierr, have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
    else if constexpr (TWO_PASS)
      osc_law<N, T, KM, true, FEAT>(P, d.Ms, d.gz, T(9.81), cv2, Jv, Jw, p, RF, q, dq, tgt, tv_given, tvin, have_ierr,
                                    ierr, have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
    else if constexpr (USE_C)
      osc_law<N, T, KM, true, FEAT>(P, d.Ms, d.gz, T(9.81), d.cv, Jv, Jw, p, RF, q, dq, tgt, tv_given, tvin, have_ierr,
                                    ierr, have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
    else  // no Coriolis vector: the slot is not read (d.gz stands in for the array type)
      osc_law<N, T, KM, false, FEAT>(P, d.Ms, d.gz, T(9.81), d.gz, Jv, Jw, p, RF, q, dq, tgt, tv_given, tvin,
                                     have_ierr, ierr, have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
  } else {
    {
      // the six rows go to the row store (the LDS slab on the GPU - free again: rne_backward has read the wrenches)
      // ctrlr_dof (rows) and the frame's joint count (columns) are uniform over the launch: scalar branches decide what
      // is stored (vector selects `on ? val : 0` cost two v_cndmask per entry, twice: 116 of them on the UR5, 5 % of
      // the first pass).  Only a kernel that also emits J (Mode F) needs the column mask on Jv / Jw themselves.
      constexpr bool kEmits = !std::is_same<typename std::decay<Emit>::type, NoEmit>::value;
      T Jv[N][3], Jw[N][3];
      if constexpr (kEmits) jacobian(jt, p, m, Jv, Jw);
      else jacobian(jt, p, N, Jv, Jw);
      emit(p, Jv, Jw, d, jt, m);
      const bool cut_cols = !kEmits && m != N;
      sfor<6>([&](auto r) ABRK_LAMBDA {
        T row[N];
        if (P.dof[r()] != 0) {
          sfor<N>([&](auto i) ABRK_LAMBDA { row[i()] = (r() < 3) ? Jv[i()][r() % 3] : Jw[i()][r() % 3]; });
          if (cut_cols) {  // a frame below the last joint: columns m.. are zero
            ABRK_UNIFORM_BLOCK();
            sfor<N>([&](auto i) ABRK_LAMBDA { row[i()] = i() < m ? row[i()] : T(0); });
          }
          scr.put_row(r, row);
        } else {
          sfor<N>([&](auto i) ABRK_LAMBDA { row[i()] = T(0); });
          scr.put_row(r, row);
          ABRK_UNIFORM_BLOCK();  // after the stores: the two arms share no tail the optimiser could merge into selects
        }
      });
      scr.seal();
    }
    ABRK_SCHED_FENCE();
    late();
    ABRK_STAMP(scr, 3, false);  // (timeline build) kinematics, dynamics, Jacobian done, task rows in the row store
    if constexpr (MAT && USE_C)
      osc_law6<N, T, true, FEAT, typename std::decay<Scr>::type, kQSteps>(P, d.Ms, d.gz, T(9.81), cvm, scr, p, RF, q, dq, tgt, tv_given, tvin, have_ierr, ierr,
                                 have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
    else if constexpr (TWO_PASS)
      osc_law6<N, T, true, FEAT, typename std::decay<Scr>::type, kQSteps>(P, d.Ms, d.gz, T(9.81), cv2, scr, p, RF, q, dq, tgt, tv_given, tvin, have_ierr, ierr,
                                 have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
    else if constexpr (USE_C)
      osc_law6<N, T, true, FEAT, typename std::decay<Scr>::type, kQSteps>(P, d.Ms, d.gz, T(9.81), d.cv, scr, p, RF, q, dq, tgt, tv_given, tvin, have_ierr, ierr,
                                 have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
    else
      osc_law6<N, T, false, FEAT, typename std::decay<Scr>::type, kQSteps>(P, d.Ms, d.gz, T(9.81), d.gz, scr, p, RF, q, dq, tgt, tv_given, tvin, have_ierr, ierr,
                                  have_ext, une, u, ts, scr.defer_ptr(), &scr.singular);
  }
}

// ---------------------------------------------------------------- Sliding.generate, one row (sliding.py:34-99)
// sctab: the kernel's sin/cos table in LDS, or nullptr (host check build: polynomial / library routine)
template <class A, class T, bool TAB = false>
ABRK_INL void sliding_row(const A& arm, const SlidingP<T>& P, const T (&q)[A::N], const T (&dq)[A::N],
                          const T (&tgt)[A::N > 3 ? A::N : 3], const T (&tv)[A::N > 3 ? A::N : 3],
                          const T (&ta)[A::N > 3 ? A::N : 3], T (&u)[A::N], T (&s)[A::N], const void* sctab = nullptr) {
  constexpr int N = A::N;
  Joints<A, T> jt;
  Dyn<A, T, CMODE_MAT> d;
  T XR[9], xo[3];
  FrameCap<T> cap;
  cap.frame = P.ref_frame;
  sfor<9>([&](auto e) ABRK_LAMBDA { cap.R[e()] = T(0); });
  sfor<3>([&](auto r) ABRK_LAMBDA { cap.o[r()] = T(0); });
  if constexpr (TAB) kin_dyn_hook(arm, q, dq, jt, d, XR, xo, cap, [](auto, const T(&)[3]) ABRK_LAMBDA {}, ScTab{sctab});
  else kin_dyn(arm, q, dq, jt, d, XR, xo, cap);
  T dq_ref[N], ddq_ref[N];
  if (P.cartesian) {
    T p[3];
    sfor<3>([&](auto r) ABRK_LAMBDA {
      p[r()] = cap.o[r()] + cap.R[r() * 3] * P.off[0] + cap.R[r() * 3 + 1] * P.off[1] + cap.R[r() * 3 + 2] * P.off[2];
    });
    T Jv[N][3], Jw[N][3], dJv[N][3], dJw[N][3], Ji[N][3];
    jacobian(jt, p, P.m_joints, Jv, Jw);
    jacobian_dot(jt, dq, Jv, P.m_joints, dJv, dJw);
    if constexpr (A::kPlanar) {
      // planar arm: the z row of J[:3] is identically zero, so pinv(J) = [pinv(J[:2]) | 0] - two rows, a single
      // Jacobi pair (numpy.linalg.pinv drops the zero singular value the same way)
      T J2[N][2], P2[N][2];
      sfor<N>([&](auto i) ABRK_LAMBDA {
        J2[i()][0] = Jv[i()][0];
        J2[i()][1] = Jv[i()][1];
      });
      pinv_KxN<2, N, T>(J2, T(1e-15), P2);
      sfor<N>([&](auto i) ABRK_LAMBDA {
        Ji[i()][0] = P2[i()][0];
        Ji[i()][1] = P2[i()][1];
        Ji[i()][2] = T(0);
      });
    } else {
      pinv_3xN<N>(Jv, T(1e-15), Ji);
    }
    T dx[3], a[3], w[3];
    sfor<3>([&](auto r) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { acc += Jv[i()][r()] * dq[i()]; });
      dx[r()] = acc;
      a[r()] = tv[r()] + P.lamb * (tgt[r()] - p[r()]);
    });
    sfor<N>([&](auto i) ABRK_LAMBDA { dq_ref[i()] = Ji[i()][0] * a[0] + Ji[i()][1] * a[1] + Ji[i()][2] * a[2]; });
    sfor<3>([&](auto r) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { acc += dJv[i()][r()] * dq_ref[i()]; });
      w[r()] = ta[r()] + P.lamb * (tv[r()] - dx[r()]) - acc;
    });
    sfor<N>([&](auto i) ABRK_LAMBDA { ddq_ref[i()] = Ji[i()][0] * w[0] + Ji[i()][1] * w[1] + Ji[i()][2] * w[2]; });
  } else {
    sfor<N>([&](auto i) ABRK_LAMBDA {
      dq_ref[i()] = tv[i()] - P.lamb * (q[i()] - tgt[i()]);
      ddq_ref[i()] = ta[i()] - P.lamb * (dq[i()] - tv[i()]);
    });
  }
  T a1[N];
  symv<N>(d.Ms, ddq_ref, a1);
  sfor<N>([&](auto i) ABRK_LAMBDA {
    s[i()] = dq[i()] - dq_ref[i()];
    T a2 = T(-0.0);
    sfor<N>([&](auto j) ABRK_LAMBDA { a2 += d.Cm[i() * N + j()] * dq_ref[j()]; });
    u[i()] = a1[i()] + a2 + T(-9.81) * d.gz[i()] - P.kd * s[i()];
  });
}

// ---------------------------------------------------------------- two-link plant, one Euler step
// ArmSim._step (arms/twojoint/arm_sim.py:101-137): closed-form inverse of the 2x2 inertia matrix.
template <class T>
ABRK_INL void twolink_step(const TwoLinkP<T>& K, T (&q)[2], T (&dq)[2], const T (&u)[2]) {
  T S2, C2;
  Rm<T>::sincos(q[1], S2, C2);
  const T M11 = K.K1 + K.K2 * C2;
  const T M12 = K.K3 + K.K4 * C2;
  const T M21 = M12, M22 = K.K3;
  const T H1 = -K.K2 * S2 * dq[0] * dq[1] - T(0.5) * K.K2 * S2 * (dq[1] * dq[1]);
  const T H2 = T(0.5) * K.K2 * S2 * (dq[0] * dq[0]);
  const T ddq1 = (H2 * M11 - H1 * M21 - M11 * u[1] + M21 * u[0]) / (M12 * M12 - M11 * M22);
  const T ddq0 = (-H2 + u[1] - M22 * ddq1) / M21;
  dq[0] += ddq0 * K.dt;
  dq[1] += ddq1 * K.dt;
  q[0] += dq[0] * K.dt;
  q[1] += dq[1] * K.dt;
}

// ---------------------------------------------------------------- InverseKinematics.generate_path, one row
// (controllers/path_planners/inverse_kinematics.py:84-135): n_steps sequential iterations, q in registers.
template <class A, class T>
ABRK_INL void ik_row(const A& arm, const IkP<T>& P, T (&q)[A::N], const T (&tgt)[6], T* __restrict__ ppath,
                     T* __restrict__ vpath) {
  constexpr int N = A::N;
  T Qd[4];
  quat_from_euler_sxyz(tgt[3], tgt[4], tgt[5], Qd);
  for (int ii = 0; ii < P.n_steps; ii++) {
    Joints<A, T> jt;
    T XR[9], xo[3], RF[9], p[3];
    NoCap nc;
    fk_forward(arm, q, jt, XR, xo, nc, [](auto, const T(&)[3]) ABRK_LAMBDA {});
    mulBE<A, T>(arm, XR, xo, RF, p);
    T Jv[N][3], Jw[N][3];
    jacobian(jt, p, N, Jv, Jw);
    T dx[3], dr[3], Qe[4];
    quat_from_R(RF, Qe);
    sfor<3>([&](auto r) ABRK_LAMBDA { dx[r()] = tgt[r()] - p[r()]; });
    {  // dr = Qe0 Qd[1:] - Qd0 Qe[1:] - Qd[1:] x Qe[1:]   (inverse_kinematics.py:92)
      T a[3] = {Qd[1], Qd[2], Qd[3]}, b[3] = {Qe[1], Qe[2], Qe[3]}, c[3];
      cross3(a, b, c);
      sfor<3>([&](auto r) ABRK_LAMBDA { dr[r()] = Qe[0] * a[r()] - Qd[0] * b[r()] - c[r()]; });
    }
    // |dx| > max_dx: dx *= max_dx / |dx|  (norm comparison on the squares, 1/|dx| from the rsqrt seed)
    T n2x = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2], n2r = dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2];
    if (n2x > P.max_dx * P.max_dx) {
      T sc = P.max_dx * Rm<T>::rsqrt(n2x);
      sfor<3>([&](auto r) ABRK_LAMBDA { dx[r()] *= sc; });
    }
    if (n2r > P.max_dr * P.max_dr) {
      T sc = P.max_dr * Rm<T>::rsqrt(n2r);
      sfor<3>([&](auto r) ABRK_LAMBDA { dr[r()] *= sc; });
    }
    T dq[N];
    if (P.method == 3) {
      // dq = pinv(Jx) dx + (I - pinv(Jx) Jx) pinv(Jw) dr   (inverse_kinematics.py:117-121)
      T pJx[N][3], pJw[N][3], b[N], jb[3];
      pinv_KxN<3, N, T>(Jv, T(1e-15), pJx);
      pinv_KxN<3, N, T>(Jw, T(1e-15), pJw);
      sfor<N>([&](auto i) ABRK_LAMBDA { b[i()] = pJw[i()][0] * dr[0] + pJw[i()][1] * dr[1] + pJw[i()][2] * dr[2]; });
      sfor<3>([&](auto r) ABRK_LAMBDA {
        T acc = T(-0.0);
        sfor<N>([&](auto i) ABRK_LAMBDA { acc += Jv[i()][r()] * b[i()]; });
        jb[r()] = acc;
      });
      sfor<N>([&](auto i) ABRK_LAMBDA {
        dq[i()] = (pJx[i()][0] * dx[0] + pJx[i()][1] * dx[1] + pJx[i()][2] * dx[2]) +
                  (b[i()] - (pJx[i()][0] * jb[0] + pJx[i()][1] * jb[1] + pJx[i()][2] * jb[2]));
      });
    } else {
      T J6[N][6];
      sfor<N>([&](auto i) ABRK_LAMBDA {
        sfor<3>([&](auto r) ABRK_LAMBDA {
          J6[i()][r()] = Jv[i()][r()];
          J6[i()][3 + r()] = Jw[i()][r()];
        });
      });
      if (P.method == 1) {  // dq = pinv(J) [dx, dr]   (inverse_kinematics.py:106-107)
        T pJ[N][6];
        pinv_KxN<6, N, T>(J6, T(1e-15), pJ);
        sfor<N>([&](auto i) ABRK_LAMBDA {
          dq[i()] = pJ[i()][0] * dx[0] + pJ[i()][1] * dx[1] + pJ[i()][2] * dx[2] + pJ[i()][3] * dr[0] +
                    pJ[i()][4] * dr[1] + pJ[i()][5] * dr[2];
        });
      } else {  // dq = J^T solve(J J^T + 0.001 I, [dx, 0.3 dr])   (inverse_kinematics.py:108-115)
        T Am[21], L[21], il[6], rhs[6] = {dx[0], dx[1], dx[2], dr[0] * T(0.3), dr[1] * T(0.3), dr[2] * T(0.3)}, y[6], x[6];
        sfor<6>([&](auto r) ABRK_LAMBDA {
          sfor<r() + 1>([&](auto c) ABRK_LAMBDA {
            T acc = (r() == c()) ? T(0.001) : T(-0.0);
            sfor<N>([&](auto i) ABRK_LAMBDA { acc += J6[i()][r()] * J6[i()][c()]; });
            Am[tri(r(), c())] = acc;
          });
        });
        chol<6>(Am, L, il);
        chol_fwd<6>(L, il, rhs, y);
        chol_bwd<6>(L, il, y, x);
        sfor<N>([&](auto i) ABRK_LAMBDA {
          T acc = T(-0.0);
          sfor<6>([&](auto r) ABRK_LAMBDA { acc += J6[i()][r()] * x[r()]; });
          dq[i()] = acc;
        });
      }
    }
    T m = T(0);
    sfor<N>([&](auto i) ABRK_LAMBDA { m = Rm<T>::fmax(m, Rm<T>::fabs(dq[i()])); });
    if (m > P.max_dq) {
      T sc = P.max_dq * rcp(m);
      sfor<N>([&](auto i) ABRK_LAMBDA { dq[i()] *= sc; });
    }
    sfor<N>([&](auto i) ABRK_LAMBDA {
      ppath[ii * N + i()] = q[i()];
      vpath[ii * N + i()] = dq[i()];
      q[i()] += dq[i()];
    });
  }
}

// ---------------------------------------------------------------- the other secondary controllers
// Task-space inertia of a point: Mx = inv or pinv(rcond) of Jv M^-1 Jv^T (3x3), from the Cholesky factor
// of M.  GATED: inv when |det| > det_thr (floating.py:50-56), else pinv; !GATED: always pinv
// (avoid_obstacles.py:112).  pinv == inv unless a singular value is below rcond * max, and
// det > rcond * trace^3 proves none is - the eigen-decomposition runs only for the rest.
// `floor`: a trace of Mx_inv at or below it is rounding noise of a point sitting on the joint axes it
// depends on (the reference then inverts noise, pinv being relative to the largest singular value, and
// returns +-maximum at random); such a point is treated as having no mobility: Mx = 0.
template <int N, class T, bool GATED>
ABRK_INL void point_inertia(const T (&L)[N * (N + 1) / 2], const T (&il)[N], const T (&Jv)[N][3], T det_thr,
                            T rcond, T floor, T (&Mx)[6]) {
  T Y[N][3];
  sfor<3>([&](auto r) ABRK_LAMBDA {
    T b[N], x[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { b[i()] = Jv[i()][r()]; });
    chol_fwd<N>(L, il, b, x);
    sfor<N>([&](auto i) ABRK_LAMBDA { Y[i()][r()] = x[i()]; });
  });
  T Am[6], trace = T(0);
  sfor<3>([&](auto r) ABRK_LAMBDA {
    sfor<r() + 1>([&](auto c) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { acc += Y[i()][r()] * Y[i()][c()]; });
      Am[tri(r(), c())] = acc;
    });
    trace += Am[tri(r(), r())];
  });
  T LA[6], ila[3];
  bool okA = chol<3>(Am, LA, ila);
  T det = T(1);
  sfor<3>([&](auto r) ABRK_LAMBDA { det *= LA[tri(r(), r())] * LA[tri(r(), r())]; });
  if (!GATED && !(trace > floor)) {
    sfor<6>([&](auto e) ABRK_LAMBDA { Mx[e()] = T(0); });
    return;
  }
  bool direct = okA && (GATED ? det > det_thr : false);
  if (!direct) direct = okA && det > rcond * trace * trace * trace;
  if (okA) chol_inverse<3>(LA, ila, Mx);
  if (!direct && okA)  // lam_min/lam_max >= 1 / (trace(A) trace(A^-1)): nothing is truncated (see osc_law)
    direct = trace * (Mx[tri(0, 0)] + Mx[tri(1, 1)] + Mx[tri(2, 2)]) * rcond < T(1);
  if (!direct) {
    T V[3][3], lam[3];
    sym3_eig(Am, V, lam);
    T smax = T(0);
    sfor<3>([&](auto r) ABRK_LAMBDA { smax = Rm<T>::fmax(smax, Rm<T>::fabs(lam[r()])); });
    T cut = rcond * smax, wv[3];
    sfor<3>([&](auto r) ABRK_LAMBDA {
      bool keep = Rm<T>::fabs(lam[r()]) > cut;
      wv[r()] = keep ? rcp(keep ? lam[r()] : T(1)) : T(0);
    });
    sfor<3>([&](auto a) ABRK_LAMBDA {
      sfor<a() + 1>([&](auto b) ABRK_LAMBDA {
        T acc = T(-0.0);
        sfor<3>([&](auto r) ABRK_LAMBDA { acc += V[a()][r()] * V[b()][r()] * wv[r()]; });
        Mx[tri(a(), b())] = acc;
      });
    });
  }
}

// AvoidJointLimits.generate (avoid_joint_limits.py:83-142), one row; q only
template <int N, class T>
ABRK_INL void limits_row(const LimitsP<T>& P, const T (&qin)[N], T (&u)[N]) {
  const T pi = T(3.141592653589793238462643383279502884);
  sfor<N>([&](auto ii) ABRK_LAMBDA {
    constexpr int i = ii();
    const T q = qin[i] - pi;  // :91
    const T mn = P.mn[i], mx = P.mx[i], mt = P.mt[i];
    const bool hmin = !P.nomin[i], hmax = !P.nomax[i];  // NaN limits compare false everywhere (:74-75)
    const T dmn = q - mn, dmx = q - mx;
    const bool both = hmin && hmax;
    const bool closer_to_min = both && Rm<T>::fabs(dmn) >= Rm<T>::fabs(dmx);  // :94-99
    const bool closer_to_max = both && Rm<T>::fabs(dmn) <= Rm<T>::fabs(dmx);
    T amin = T(0), amax = T(0);
    if (P.gr[i]) {  // :108-115.  1/0 -> +-inf in the reference: exp(+inf) is capped by max_torque, exp(-inf) = 0
      if (hmin) amin = dmn == T(0) ? mt : Rm<T>::fmin(Rm<T>::exp(rcp(dmn == T(0) ? T(1) : dmn)), mt);
      if (hmax) amax = dmx == T(0) ? -mt : -Rm<T>::fmin(Rm<T>::exp(-rcp(dmx == T(0) ? T(1) : dmx)), mt);
    }
    bool min_index = hmin && dmn < T(0), max_index = hmax && dmx > T(0);  // :118-119
    if (P.cz[i]) {  // :124-134
      const bool mi = min_index && (hmax && dmx > T(0)) && closer_to_max;
      const bool xi = max_index && (hmin && dmn < T(0)) && closer_to_min;
      min_index = mi;
      max_index = xi;
    }
    if (min_index) amin = mt;  // :136-140
    if (max_index) amax = -mt;
    u[i] = amin + amax;
  });
}

// Floating.generate (floating.py:27-71), one row
template <class A, class T>
ABRK_INL void floating_row(const A& arm, int dynamic, int task_space, const T (&q)[A::N], const T (&dq)[A::N],
                           T (&u)[A::N]) {
  constexpr int N = A::N;
  Joints<A, T> jt;
  Dyn<A, T, CMODE_NONE> d;
  T XR[9], xo[3];
  NoCap nc;
  T zero[N];
  sfor<N>([&](auto i) ABRK_LAMBDA { zero[i()] = T(0); });
  kin_dyn(arm, q, zero, jt, d, XR, xo, nc);
  if (task_space) {
    // u = J^T (-(M^-1 J^T Mx)^T g) = -J^T Mx J M^-1 g  (floating.py:42-61), g = -9.81 gz
    T p[3], Jv[N][3], Jw[N][3], L[N * (N + 1) / 2], il[N], Mx[6];
    mulBE_pt<A, T>(arm, XR, xo, p);
    jacobian(jt, p, N, Jv, Jw);
    chol<N>(d.Ms, L, il);
    point_inertia<N, T, true>(L, il, Jv, T(1e-3), T(1e-4), T(0), Mx);
    T gq[N], y[N], w[N], jw[3], f[3];
    sfor<N>([&](auto i) ABRK_LAMBDA { gq[i()] = T(-9.81) * d.gz[i()]; });
    chol_fwd<N>(L, il, gq, y);
    chol_bwd<N>(L, il, y, w);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { acc += Jv[i()][r()] * w[i()]; });
      jw[r()] = acc;
    });
    symv<3>(Mx, jw, f);
    sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = -(Jv[i()][0] * f[0] + Jv[i()][1] * f[1] + Jv[i()][2] * f[2]); });
  } else {
    sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = T(9.81) * d.gz[i()]; });  // u = -g (floating.py:64)
  }
  if (dynamic) {  // floating.py:66-69
    T Mdq[N];
    symv<N>(d.Ms, dq, Mdq);
    sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] -= Mdq[i()]; });
  }
}

// AvoidObstacles.generate (avoid_obstacles.py:38-120), one row.  Segment ii runs from the origin of
// joint_ii to that of joint_ii+1 (the EE for the last one, :64-68); its closest point to the obstacle is a
// point of link ii+1, so its Jacobian columns are W_i (closest - o_i), i <= ii  (the reference gets the
// same through T_inv("link{ii+1}") and J("link{ii+1}", x=m), :107-110).
template <class A, class T>
ABRK_INL void obstacles_row(const A& arm, const ObsP<T>& P, const T (&q)[A::N], T (&u)[A::N]) {
  constexpr int N = A::N;
  Joints<A, T> jt;
  Dyn<A, T, CMODE_NONE> d;
  T XR[9], xo[3], pe[3];
  NoCap nc;
  T zero[N];
  sfor<N>([&](auto i) ABRK_LAMBDA { zero[i()] = T(0); });
  // General (non-orthogonal) chains: the reference's T_inv is [R^T | -R^T t] (base_config.py:791-837), not
  // the inverse, so the point it differentiates is o_l + R_l R_l^T (closest - o_l) in link l's frame - up to
  // 4e-4 away from `closest` on Jaco2, whose rotation constants are rounded.  Keep G_l = R_l R_l^T, o_l.
  constexpr int NG = A::kOrtho ? 1 : N;
  T G[NG][6], ol[NG][3];
  kin_dyn_hook(arm, q, zero, jt, d, XR, xo, nc, [&](auto Lc, const T(&pl)[3]) ABRK_LAMBDA {
    if constexpr (!A::kOrtho) {
      constexpr int l = Lc() - 1;
      T RL[9];
      mulB_rot<A, T, l>(arm, XR, RL);
      sfor<3>([&](auto a) ABRK_LAMBDA {
        sfor<a() + 1>([&](auto b) ABRK_LAMBDA {
          G[l][tri(a(), b())] = RL[a() * 3] * RL[b() * 3] + RL[a() * 3 + 1] * RL[b() * 3 + 1] +
                                RL[a() * 3 + 2] * RL[b() * 3 + 2];
        });
        ol[l][a()] = pl[a()];
      });
    }
  });
  mulBE_pt<A, T>(arm, XR, xo, pe);
  T L[N * (N + 1) / 2], il[N];
  chol<N>(d.Ms, L, il);
  sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = T(0); });
  const T lo = P.threshold / T(50), ithr = T(1) / P.threshold;
  // Mx_inv of a point at distance d from the axes it hangs on is ~ d^2 |M^-1|; below (1e-12 segment
  // lengths)^2 (fp32: 1e-5) it is rounding noise (see point_inertia)
  T noise = T(0);
  sfor<N>([&](auto i) ABRK_LAMBDA { noise += il[i()] * il[i()]; });
  noise *= sizeof(T) == 8 ? T(1e-24) : T(1e-10);
  // The first two segments hang on one and two joints: their Mx_inv = J_b W J_b^T has rank 1 / 2 BY STRUCTURE (W = the
  // leading 1x1 / 2x2 block of M^-1), the reference's pinv(rcond=0.01) always truncates there, and the general 3x3
  // eigen path (~760 fp64 instructions per near pair) reduces to a division / a 2x2 problem in the plane of the two
  // Jacobian columns.  W from the first two columns of L^-1, once per row:
  T w00 = T(0), w01 = T(0), w11 = T(0);
  {
    T e[N], y0[N], y1[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { e[i()] = i() == 0 ? T(1) : T(0); });
    chol_fwd<N>(L, il, e, y0);
    w00 = T(0);
    sfor<N>([&](auto i) ABRK_LAMBDA { w00 += y0[i()] * y0[i()]; });
    if constexpr (N >= 2) {
      sfor<N>([&](auto i) ABRK_LAMBDA { e[i()] = i() == 1 ? T(1) : T(0); });
      chol_fwd<N>(L, il, e, y1);
      sfor<N>([&](auto i) ABRK_LAMBDA {
        w01 += y0[i()] * y1[i()];
        w11 += y1[i()] * y1[i()];
      });
    }
  }
  for (int ob = 0; ob < P.n; ob++) {
    const T v[3] = {P.obs[ob][0], P.obs[ob][1], P.obs[ob][2]};
    const T radius = P.obs[ob][3];
    sfor<N>([&](auto iic) ABRK_LAMBDA {
      constexpr int ii = iic();
      T p1[3], p2[3], vl[3], vo[3];
      sfor<3>([&](auto r) ABRK_LAMBDA {
        p1[r()] = jt.o[ii][r()];
        if constexpr (ii == N - 1) p2[r()] = pe[r()];
        else p2[r()] = jt.o[ii + 1 < N ? ii + 1 : ii][r()];
        vl[r()] = p2[r()] - p1[r()];
        vo[r()] = v[r()] - p1[r()];
      });
      const T len2 = dot3(vl, vl);
      // a zero-length segment is 0/0 in the reference: every later comparison is false, no contribution
      if (len2 > T(0)) {
        T pr = dot3(vo, vl) * rcp(len2);  // :76
        pr = pr < T(0) ? T(0) : (pr > T(1) ? T(1) : pr);  // :77-84 (closest = p1, p2 or in between)
        T cl[3], dv[3];
        sfor<3>([&](auto r) ABRK_LAMBDA {
          cl[r()] = pr == T(1) ? p2[r()] : p1[r()] + pr * vl[r()];
          dv[r()] = v[r()] - cl[r()];
        });
        const T d2 = dot3(dv, dv);
        const T dist = d2 > T(0) ? d2 * Rm<T>::rsqrt(d2 > T(0) ? d2 : T(1)) : T(0);
        const T rho = Rm<T>::fmax(dist - radius, lo);  // :90
        if (rho < P.threshold) {
          // Fpsp = eta (1/rho - 1/threshold) / rho^1.5 * (v - closest)/rho   (:94-102)
          const T irho = rcp(rho);
          const T k = T(0.02) * (irho - ithr) * irho * irho * Rm<T>::rsqrt(rho);
          T F[3] = {k * dv[0], k * dv[1], k * dv[2]};
          T Jp[N][3];
          if constexpr (!A::kOrtho) {
            T e[3] = {cl[0] - ol[ii][0], cl[1] - ol[ii][1], cl[2] - ol[ii][2]}, ge[3];
            symv<3>(G[ii], e, ge);
            sfor<3>([&](auto r) ABRK_LAMBDA { cl[r()] = ol[ii][r()] + ge[r()]; });
          }
          sfor<N>([&](auto i) ABRK_LAMBDA {
            if constexpr (i() <= ii) {
              T dl[3] = {cl[0] - jt.o[i()][0], cl[1] - jt.o[i()][1], cl[2] - jt.o[i()][2]};
              wapply<i()>(jt, dl, Jp[i()]);
            } else {
              Jp[i()][0] = Jp[i()][1] = Jp[i()][2] = T(0);
            }
          });
          const T floor = noise * len2;
          bool done = false;
          if constexpr (ii == 0) {
            // rank 1: Mx_inv = w00 j j^T, pinv = j j^T / (w00 |j|^4), J^T Mx F = (j . F) / (w00 |j|^2)
            const T tr = w00 * dot3(Jp[0], Jp[0]);
            if (tr > floor) u[0] -= dot3(Jp[0], F) * rcp(tr);
            done = true;
          } else if constexpr (ii == 1) {
            // rank 2: J_b = Q R (Q: orthonormal pair spanning the two columns), Mx_inv = Q (R W R^T) Q^T; the 2x2 block
            // S = R W R^T has the non-zero eigenvalues of Mx_inv - closed form, one rotation; keep lam_- iff
            // lam_- > 0.01 lam_+;  J_b^T Mx F = R^T S^+ (Q^T F).  Degenerate pairs (a zero first column, parallel
            // columns to working precision) take the general path.
            const T g00 = dot3(Jp[0], Jp[0]);
            const T ig0 = Rm<T>::rsqrt(Rm<T>::fmax(g00, Rm<T>::tiny()));
            const T q0[3] = {Jp[0][0] * ig0, Jp[0][1] * ig0, Jp[0][2] * ig0};
            const T r01 = dot3(q0, Jp[1]);
            const T p1[3] = {Jp[1][0] - r01 * q0[0], Jp[1][1] - r01 * q0[1], Jp[1][2] - r01 * q0[2]};
            const T h11 = dot3(p1, p1), g11 = dot3(Jp[1], Jp[1]);
            if (g00 > T(0) && h11 > (sizeof(T) == 8 ? T(1e-20) : T(1e-8)) * g11) {
              const T ih = Rm<T>::rsqrt(h11);
              const T q1[3] = {p1[0] * ih, p1[1] * ih, p1[2] * ih};
              const T r00 = g00 * ig0, r11 = h11 * ih;
              // S = R W R^T, R = [[r00, r01], [0, r11]]
              const T a0 = r00 * w00 + r01 * w01, a1 = r00 * w01 + r01 * w11;  // first row of R W
              const T s00 = a0 * r00 + a1 * r01, s01 = a1 * r11, s11 = r11 * w11 * r11;
              const T tr = s00 + s11;
              if (tr > floor) {
                const T t = jacobi_tan(s11 - s00, T(2) * s01);
                const T c = Rm<T>::rsqrt(t * t + T(1)), sn = t * c;
                const T l0 = s00 - t * s01, l1 = s11 + t * s01;  // eigenpairs (c, -sn) and (sn, c)
                const T lmax = Rm<T>::fmax(l0, l1), cut = T(0.01) * lmax;
                const T i0 = l0 > cut ? rcp(l0 > cut ? l0 : T(1)) : T(0), i1 = l1 > cut ? rcp(l1 > cut ? l1 : T(1)) : T(0);
                const T b0 = dot3(q0, F), b1 = dot3(q1, F);
                const T z0 = (c * b0 - sn * b1) * i0, z1 = (sn * b0 + c * b1) * i1;  // S^+ b in the eigenbasis
                const T x0 = c * z0 + sn * z1, x1 = -sn * z0 + c * z1;
                u[0] -= r00 * x0;
                u[1] -= r01 * x0 + r11 * x1;
              }
              done = true;
            }
          }
          if (!done) {
            T Mx[6], f[3];
            point_inertia<N, T, false>(L, il, Jp, T(0), T(0.01), floor, Mx);  // :114-117
            symv<3>(Mx, F, f);
            sfor<N>([&](auto i) ABRK_LAMBDA {
              if constexpr (i() <= ii) u[i()] -= Jp[i()][0] * f[0] + Jp[i()][1] * f[1] + Jp[i()][2] * f[2];  // :119
            });
          }
        }
      }
    });
  }
  sfor<N>([&](auto i) ABRK_LAMBDA {  // np.clip (:121)
    T x = u[i()] * P.gain;
    u[i()] = x < -P.maximum ? -P.maximum : (x > P.maximum ? P.maximum : x);
  });
}

// ---- AvoidObstacles with the heavy pairs redistributed over the wavefront (orthogonal chains of three joints and more).
// obstacles_row runs every (obstacle, segment) slot on every lane that is near ITS obstacle - and makes the other 63
// wait: with three obstacles a random UR5 state has 2.0 near pairs on the segments that need the general 3 x 3 path
// (ii >= 2: ~900 instructions each), but the wavefront walks 8.4 of its 12 such slots.  Split in three:
//   phase A (one row per lane): kinematics, M and its factor, every near test, the cheap rank-1 / rank-2 pairs of the
//     first two segments; the row's factor and frames go to a record, its near heavy pairs to a bit mask;
//   pairs (one PAIR per lane, whoever's row it is): the wavefront's pairs are numbered through (a prefix sum of the
//     popcounts), lane j of round r takes pair 64 r + j, reads the owning row's record and leaves the pair's
//     contribution to u;
//   the owner adds its pairs' contributions in slot order, then gain and clip.
// The records live in LDS on the GPU (obstacles_lds_kernel: [field][lane], 66 values per row for six joints = 33 KiB per
// wavefront), in plain arrays in the host check build.  Same arithmetic per pair as obstacles_row; only the order in
// which a row's contributions are summed differs (light pairs first, then heavy ones by obstacle and segment).
template <int N>
constexpr int obs_rec_len() { return N * (N + 1) / 2 + N + 3 * N + 3 * N + 3; }  // L, il, o, z, pe
template <int N>
constexpr int obs_heavy_segments() { return N > 2 ? N - 2 : 0; }
// record field offsets
template <int N>
struct ObsRec {
  static constexpr int L = 0, IL = N * (N + 1) / 2, O = IL + N, Z = O + 3 * N, PE = Z + 3 * N;
};

// phase A.  `put(field, value)` stores a record field; -> the mask of near heavy slots (slot = ob * (N - 2) + ii - 2)
template <class A, class T, class Put>
ABRK_INL unsigned long long obstacles_phase_a(const A& arm, const ObsP<T>& P, const T (&q)[A::N], T (&u)[A::N], Put&& put) {
  constexpr int N = A::N;
  static_assert(A::kOrtho && N >= 3, "orthogonal chains with heavy segments");
  using R = ObsRec<N>;
  Joints<A, T> jt;
  Dyn<A, T, CMODE_NONE> d;
  T XR[9], xo[3], pe[3];
  NoCap nc;
  T zero[N];
  sfor<N>([&](auto i) ABRK_LAMBDA { zero[i()] = T(0); });
  kin_dyn(arm, q, zero, jt, d, XR, xo, nc);
  mulBE_pt<A, T>(arm, XR, xo, pe);
  T L[N * (N + 1) / 2], il[N];
  chol<N>(d.Ms, L, il);
  sfor<N*(N + 1) / 2>([&](auto e) ABRK_LAMBDA { put(R::L + e(), L[e()]); });
  sfor<N>([&](auto i) ABRK_LAMBDA {
    put(R::IL + i(), il[i()]);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      put(R::O + 3 * i() + r(), jt.o[i()][r()]);
      put(R::Z + 3 * i() + r(), jt.z[i()][r()]);
    });
  });
  sfor<3>([&](auto r) ABRK_LAMBDA { put(R::PE + r(), pe[r()]); });
  sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] = T(0); });
  const T lo = P.threshold / T(50), ithr = T(1) / P.threshold;
  T noise = T(0);
  sfor<N>([&](auto i) ABRK_LAMBDA { noise += il[i()] * il[i()]; });
  noise *= sizeof(T) == 8 ? T(1e-24) : T(1e-10);
  T w00 = T(0), w01 = T(0), w11 = T(0);
  {
    T e[N], y0[N], y1[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { e[i()] = i() == 0 ? T(1) : T(0); });
    chol_fwd<N>(L, il, e, y0);
    sfor<N>([&](auto i) ABRK_LAMBDA { w00 += y0[i()] * y0[i()]; });
    sfor<N>([&](auto i) ABRK_LAMBDA { e[i()] = i() == 1 ? T(1) : T(0); });
    chol_fwd<N>(L, il, e, y1);
    sfor<N>([&](auto i) ABRK_LAMBDA {
      w01 += y0[i()] * y1[i()];
      w11 += y1[i()] * y1[i()];
    });
  }
  unsigned long long heavy = 0ull;
  for (int ob = 0; ob < P.n; ob++) {
    const T v[3] = {P.obs[ob][0], P.obs[ob][1], P.obs[ob][2]};
    const T radius = P.obs[ob][3];
    sfor<N>([&](auto iic) ABRK_LAMBDA {
      constexpr int ii = iic();
      T p1[3], p2[3], vl[3], vo[3];
      sfor<3>([&](auto r) ABRK_LAMBDA {
        p1[r()] = jt.o[ii][r()];
        if constexpr (ii == N - 1) p2[r()] = pe[r()];
        else p2[r()] = jt.o[ii + 1 < N ? ii + 1 : ii][r()];
        vl[r()] = p2[r()] - p1[r()];
        vo[r()] = v[r()] - p1[r()];
      });
      const T len2 = dot3(vl, vl);
      if (len2 > T(0)) {
        T pr = dot3(vo, vl) * rcp(len2);
        pr = pr < T(0) ? T(0) : (pr > T(1) ? T(1) : pr);
        T cl[3], dv[3];
        sfor<3>([&](auto r) ABRK_LAMBDA {
          cl[r()] = pr == T(1) ? p2[r()] : p1[r()] + pr * vl[r()];
          dv[r()] = v[r()] - cl[r()];
        });
        const T d2 = dot3(dv, dv);
        const T dist = d2 > T(0) ? d2 * Rm<T>::rsqrt(d2 > T(0) ? d2 : T(1)) : T(0);
        const T rho = Rm<T>::fmax(dist - radius, lo);
        if (rho < P.threshold) {
          if constexpr (ii >= 2) {
            heavy |= 1ull << (ob * (N - 2) + (ii - 2));
          } else {
            const T irho = rcp(rho);
            const T k = T(0.02) * (irho - ithr) * irho * irho * Rm<T>::rsqrt(rho);
            T F[3] = {k * dv[0], k * dv[1], k * dv[2]};
            T Jp[2][3];
            sfor<2>([&](auto i) ABRK_LAMBDA {
              if constexpr (i() <= ii) {
                T dl[3] = {cl[0] - jt.o[i()][0], cl[1] - jt.o[i()][1], cl[2] - jt.o[i()][2]};
                wapply<i()>(jt, dl, Jp[i()]);
              } else {
                Jp[i()][0] = Jp[i()][1] = Jp[i()][2] = T(0);
              }
            });
            const T floor = noise * len2;
            bool done = false;
            if constexpr (ii == 0) {
              const T tr = w00 * dot3(Jp[0], Jp[0]);
              if (tr > floor) u[0] -= dot3(Jp[0], F) * rcp(tr);
              done = true;
            } else {
              const T g00 = dot3(Jp[0], Jp[0]);
              const T ig0 = Rm<T>::rsqrt(Rm<T>::fmax(g00, Rm<T>::tiny()));
              const T q0[3] = {Jp[0][0] * ig0, Jp[0][1] * ig0, Jp[0][2] * ig0};
              const T r01 = dot3(q0, Jp[1]);
              const T pp1[3] = {Jp[1][0] - r01 * q0[0], Jp[1][1] - r01 * q0[1], Jp[1][2] - r01 * q0[2]};
              const T h11 = dot3(pp1, pp1), g11 = dot3(Jp[1], Jp[1]);
              if (g00 > T(0) && h11 > (sizeof(T) == 8 ? T(1e-20) : T(1e-8)) * g11) {
                const T ih = Rm<T>::rsqrt(h11);
                const T q1[3] = {pp1[0] * ih, pp1[1] * ih, pp1[2] * ih};
                const T r00 = g00 * ig0, r11 = h11 * ih;
                const T a0 = r00 * w00 + r01 * w01, a1 = r00 * w01 + r01 * w11;
                const T s00 = a0 * r00 + a1 * r01, s01 = a1 * r11, s11 = r11 * w11 * r11;
                const T tr = s00 + s11;
                if (tr > floor) {
                  const T t = jacobi_tan(s11 - s00, T(2) * s01);
                  const T c = Rm<T>::rsqrt(t * t + T(1)), sn = t * c;
                  const T l0 = s00 - t * s01, l1 = s11 + t * s01;
                  const T lmax = Rm<T>::fmax(l0, l1), cut = T(0.01) * lmax;
                  const T i0 = l0 > cut ? rcp(l0 > cut ? l0 : T(1)) : T(0), i1 = l1 > cut ? rcp(l1 > cut ? l1 : T(1)) : T(0);
                  const T b0 = dot3(q0, F), b1 = dot3(q1, F);
                  const T z0 = (c * b0 - sn * b1) * i0, z1 = (sn * b0 + c * b1) * i1;
                  const T x0 = c * z0 + sn * z1, x1 = -sn * z0 + c * z1;
                  u[0] -= r00 * x0;
                  u[1] -= r01 * x0 + r11 * x1;
                }
                done = true;
              }
            }
            if (!done) {  // degenerate first two columns: the general path, here (rare)
              T JpN[N][3], Mx[6], f[3];
              sfor<N>([&](auto i) ABRK_LAMBDA {
                sfor<3>([&](auto r) ABRK_LAMBDA { JpN[i()][r()] = i() < 2 ? Jp[i() < 2 ? i() : 0][r()] : T(0); });
              });
              point_inertia<N, T, false>(L, il, JpN, T(0), T(0.01), floor, Mx);
              symv<3>(Mx, F, f);
              sfor<2>([&](auto i) ABRK_LAMBDA {
                if constexpr (i() <= ii) u[i()] -= Jp[i()][0] * f[0] + Jp[i()][1] * f[1] + Jp[i()][2] * f[2];
              });
            }
          }
        }
      }
    });
  }
  return heavy;
}

// one heavy pair: slot = ob * (N - 2) + ii - 2 of the row whose record `get(field)` reads -> its contribution c to u
// (already negated: u += c)
template <int N, class T, class Get>
ABRK_INL void obstacles_pair(const ObsP<T>& P, int slot, Get&& get, T (&c)[N]) {
  using R = ObsRec<N>;
  constexpr int NH = N - 2;
  const int ob = slot / NH, ii = 2 + slot % NH;
  T L[N * (N + 1) / 2], il[N], o[N][3], z[N][3];
  sfor<N*(N + 1) / 2>([&](auto e) ABRK_LAMBDA { L[e()] = get(R::L + e()); });
  sfor<N>([&](auto i) ABRK_LAMBDA {
    il[i()] = get(R::IL + i());
    sfor<3>([&](auto r) ABRK_LAMBDA {
      o[i()][r()] = get(R::O + 3 * i() + r());
      z[i()][r()] = get(R::Z + 3 * i() + r());
    });
  });
  // the segment's end points by their run-time index: straight from the record (o_ii; o_ii+1, or the EE after the last)
  T p1[3], p2[3], vl[3], vo[3], v[3];
  const int f2 = ii == N - 1 ? R::PE : R::O + 3 * (ii + 1);
  sfor<3>([&](auto r) ABRK_LAMBDA {
    p1[r()] = get(R::O + 3 * ii + r());
    p2[r()] = get(f2 + r());
    v[r()] = P.obs[ob][r()];
    vl[r()] = p2[r()] - p1[r()];
    vo[r()] = v[r()] - p1[r()];
  });
  const T radius = P.obs[ob][3];
  const T lo = P.threshold / T(50), ithr = T(1) / P.threshold;
  const T len2 = dot3(vl, vl);
  T pr = dot3(vo, vl) * rcp(len2);
  pr = pr < T(0) ? T(0) : (pr > T(1) ? T(1) : pr);
  T cl[3], dv[3];
  sfor<3>([&](auto r) ABRK_LAMBDA {
    cl[r()] = pr == T(1) ? p2[r()] : p1[r()] + pr * vl[r()];
    dv[r()] = v[r()] - cl[r()];
  });
  const T d2 = dot3(dv, dv);
  const T dist = d2 > T(0) ? d2 * Rm<T>::rsqrt(d2 > T(0) ? d2 : T(1)) : T(0);
  const T rho = Rm<T>::fmax(dist - radius, lo);
  const T irho = rcp(rho);
  const T k = T(0.02) * (irho - ithr) * irho * irho * Rm<T>::rsqrt(rho);
  T F[3] = {k * dv[0], k * dv[1], k * dv[2]};
  T Jp[N][3];
  sfor<N>([&](auto i) ABRK_LAMBDA {
    T dl[3] = {cl[0] - o[i()][0], cl[1] - o[i()][1], cl[2] - o[i()][2]}, w[3];
    cross3(z[i()], dl, w);
    const bool on = i() <= ii;
    sfor<3>([&](auto r) ABRK_LAMBDA { Jp[i()][r()] = on ? w[r()] : T(0); });
  });
  T noise = T(0);
  sfor<N>([&](auto i) ABRK_LAMBDA { noise += il[i()] * il[i()]; });
  noise *= sizeof(T) == 8 ? T(1e-24) : T(1e-10);
  T Mx[6], f[3];
  point_inertia<N, T, false>(L, il, Jp, T(0), T(0.01), noise * len2, Mx);
  symv<3>(Mx, F, f);
  sfor<N>([&](auto i) ABRK_LAMBDA { c[i()] = -(Jp[i()][0] * f[0] + Jp[i()][1] * f[1] + Jp[i()][2] * f[2]); });
}

// np.clip(u * gain) (avoid_obstacles.py:121)
template <int N, class T>
ABRK_INL void obstacles_finish(const ObsP<T>& P, T (&u)[N]) {
  sfor<N>([&](auto i) ABRK_LAMBDA {
    T x = u[i()] * P.gain;
    u[i()] = x < -P.maximum ? -P.maximum : (x > P.maximum ? P.maximum : x);
  });
}
// the three steps on one row, pair by pair (host check build; what obstacles_lds_kernel does with the pairs spread over
// the wavefront)
template <class A, class T>
ABRK_INL void obstacles_row_split(const A& arm, const ObsP<T>& P, const T (&q)[A::N], T (&u)[A::N]) {
  constexpr int N = A::N;
  T rec[obs_rec_len<N>()];
  unsigned long long heavy = obstacles_phase_a<A, T>(arm, P, q, u, [&](int f, T v) ABRK_LAMBDA { rec[f] = v; });
  while (heavy) {
    const int slot = __builtin_ctzll(heavy);
    heavy &= heavy - 1;
    T c[N];
    obstacles_pair<N, T>(P, slot, [&](int f) ABRK_LAMBDA { return rec[f]; }, c);
    sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] += c[i()]; });
  }
  obstacles_finish<N, T>(P, u);
}

// ---------------------------------------------------------------- Joint / Damping / RestingConfig, one row
template <class A, class T>
ABRK_INL void joint_row(const A& arm, const JointP<T>& P, const T (&q)[A::N], const T (&dq)[A::N],
                        const T (&tgt)[A::N], const T (&tv)[A::N], T (&u)[A::N]) {
  constexpr int N = A::N;
  const T pi = T(3.141592653589793238462643383279502884);
  Joints<A, T> jt;
  Dyn<A, T, CMODE_NONE> d;
  T XR[9], xo[3];
  NoCap nc;
  kin_dyn(arm, q, dq, jt, d, XR, xo, nc);
  T v[N];
  sfor<N>([&](auto i) ABRK_LAMBDA { v[i()] = T(0); });
  if (P.c.kind != 0) {
    null_command<N>(P.c, q, dq, v);
  } else {
    sfor<N>([&](auto i) ABRK_LAMBDA {
      T qt = pymod_pos(tgt[i()] - q[i()] + pi, pi * T(2)) - pi;
      v[i()] = P.c.kp * qt + P.c.kv * (tv[i()] - dq[i()]);
    });
  }
  symv<N>(d.Ms, v, u);
  if (P.c.kind == 0 && P.account_for_gravity) sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] += T(9.81) * d.gz[i()]; });
}

// ---- the helper methods of OSC as functions of their own (what controllers/tests/test_osc.py calls directly).
// The fused OSC kernels carry the same steps inline, specialised (packed M from the chain, certificates that
// skip the eigen-decomposition, solve-based Mx u); these are the general forms behind OSC._Mx / ._velocity_limiting /
// ._calc_orientation_forces of the Python mirror.

// OSC._Mx (osc.py:120-147): M [N,N] symmetric positive definite (a mass matrix), J = the k task rows OSC keeps
// (Jr[i][r] = J(r,i), rows >= k ignored).  Mx (k x k block of the packed 6 x 6) and M^-1 (packed).
template <int N, class T>
ABRK_INL void mx_row(const T (&Ms)[N * (N + 1) / 2], const T (&Jin)[N][6], int k, T thr, T (&Mx)[21],
                     T (&Minv)[N * (N + 1) / 2]) {
  T L[N * (N + 1) / 2], il[N];
  chol<N>(Ms, L, il);
  chol_inverse<N>(L, il, Minv);  // osc.py:136
  bool sel[6];
  T Y[N][6];
  sfor<6>([&](auto r) ABRK_LAMBDA {
    sel[r()] = r() < k;
    T b[N], x[N];
    sfor<N>([&](auto i) ABRK_LAMBDA { b[i()] = sel[r()] ? Jin[i()][r()] : T(0); });
    chol_fwd<N>(L, il, b, x);
    sfor<N>([&](auto i) ABRK_LAMBDA { Y[i()][r()] = x[i()]; });
  });
  // Mx_inv = J M^-1 J^T = Y^T Y (osc.py:137); a masked row is an isolated unit diagonal, which leaves determinant,
  // inverse and singular values of the selected block as they are
  T Am[21];
  sfor<6>([&](auto r) ABRK_LAMBDA {
    sfor<r() + 1>([&](auto c) ABRK_LAMBDA {
      T acc = T(-0.0);
      sfor<N>([&](auto i) ABRK_LAMBDA { acc += Y[i()][r()] * Y[i()][c()]; });
      Am[tri(r(), c())] = acc;
    });
    if (!sel[r()]) Am[tri(r(), r())] = T(1);
  });
  T LA[21], ila[6];
  const bool okA = chol<6>(Am, LA, ila);
  T det = T(1);
  sfor<6>([&](auto r) ABRK_LAMBDA { det *= LA[tri(r(), r())] * LA[tri(r(), r())]; });
  if (okA && det >= thr) {  // osc.py:138-141
    chol_inverse<6>(LA, ila, Mx);
  } else {  // osc.py:142-145: pinv(rcond = 0.1 threshold) of a symmetric positive semi-definite matrix
    T S[21], V[6][6], lam[6];
    sfor<21>([&](auto e) ABRK_LAMBDA { S[e()] = Am[e()]; });
    sfor<6>([&](auto r) ABRK_LAMBDA { S[tri(r(), r())] = sel[r()] ? S[tri(r(), r())] : T(0); });  // masked rows: eigenvalue 0
    sym_eig<6>(S, V, lam);
    T smax = T(0);
    sfor<6>([&](auto r) ABRK_LAMBDA { smax = Rm<T>::fmax(smax, Rm<T>::fabs(lam[r()])); });
    const T cut = T(0.1) * thr * smax;
    T wv[6];
    sfor<6>([&](auto r) ABRK_LAMBDA {
      const bool keep = Rm<T>::fabs(lam[r()]) > cut;
      wv[r()] = keep ? rcp(keep ? lam[r()] : T(1)) : T(0);
    });
    sfor<6>([&](auto a) ABRK_LAMBDA {
      sfor<a() + 1>([&](auto b) ABRK_LAMBDA {
        T acc = T(-0.0);
        sfor<6>([&](auto r) ABRK_LAMBDA { acc += V[a()][r()] * V[b()][r()] * wv[r()]; });
        Mx[tri(a(), b())] = acc;
      });
    });
  }
}

// OSC._velocity_limiting (osc.py:198-215, constants osc.py:89-115)
template <class T>
ABRK_INL void velocity_limiting_row(T kp, T ko, T kv, T vmax0, T vmax1, T (&ut)[6]) {
  const T sat_xyz = vmax0 / kp * kv, sat_abg = vmax1 / ko * kv;
  const T nx = Rm<T>::sqrt(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]);
  const T na = Rm<T>::sqrt(ut[3] * ut[3] + ut[4] * ut[4] + ut[5] * ut[5]);
  const T sx = (nx > sat_xyz) ? sat_xyz / nx : T(1);
  const T sa = (na > sat_abg) ? sat_abg / na : T(1);
  const T lx = kp / kv, la = ko / kv;
  sfor<3>([&](auto r) ABRK_LAMBDA {
    ut[r()] = kv * sx * lx * ut[r()];
    ut[3 + r()] = kv * sa * la * ut[3 + r()];
  });
}

}  // namespace abrk
