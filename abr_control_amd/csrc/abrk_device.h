// abrk_device.h - per-row device arithmetic of the batched arm engine (gfx950 / CDNA4).
//
// One lane evaluates one arm instance ("row" of the batch) entirely in registers: the
// work per row is a short serial chain (forward kinematics over <= 7 joints) followed by
// small fixed-size dense algebra, so rows map to lanes and a 64-wide wavefront advances
// 64 independent arms in lock-step with no cross-lane traffic and no divergence on the
// main path.  All loops are unrolled at compile time (`sfor`) so every small array lives
// in VGPRs, and for the built-in arms every static frame transform is a compile-time
// constant whose 0 / +-1 entries are folded away (`cfma`).
//
// What is computed follows abr_control/arms/base_config.py (reference file:line cited at
// each function) but NOT how: the reference differentiates symbolic expressions; here the
// affine joint chain is differentiated in closed form:
//   for a point p rigidly attached after joints 0..m-1, with P_i / o_i the linear part /
//   origin of T(joint_i) and  W_i = P_i Zhat P_i^-1  (= [z_i]x when P_i is orthogonal):
//       dp/dq_i            = W_i (p - o_i)                      (i < m)
//       d2p/dq_i dq_k      = W_min(i,k) W_max(i,k) (p - o_max)  (i,k < m)
//       dz_i/dq_k          = W_k z_i                            (k < i),  z_i = P_i e_z
//   exact for ANY affine static transforms (Jaco2's constants are not orthogonal).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "abrk_arms_builtin.h"
#include "abrk_sincos_table.h"

#ifndef ABRK_HD
#define ABRK_HD __device__
#endif
#define ABRK_INL ABRK_HD __forceinline__
#define ABRK_LAMBDA __attribute__((always_inline))
// Scheduling fence: keeps the late input loads (and everything after) below the forward-kinematics
// phase so their registers are not live during the register-pressure peak.  No-op on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define ABRK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ABRK_SCHED_FENCE() ((void)0)
#endif

// development aid: -DABRK_MARKS puts "; MARK <name>" comments into the ISA at phase boundaries (tools/phase_counts.py)
#if defined(ABRK_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define ABRK_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ABRK_MARK(name) ((void)0)
#endif
// development aid: -DABRK_TIMELINE stamps a wavefront's progress with the constant-rate counter (s_memrealtime, 100 MHz)
// at a handful of points of the x,y,z OSC kernel (entry, table barrier, inputs landed, dynamics done, law done, stores
// issued, stores complete); the stamps ride in scalar registers of the row's scratch object and leave through the
// kernel's (otherwise unused) worklist pointer.  tools/microbench/shard_step_timeline.hip is the one user; the product
// library is never built with it.
#if defined(ABRK_TIMELINE) && defined(__HIP_DEVICE_COMPILE__)
#if defined(ABRK_TIMELINE_LIGHT)  // no forced waits: how much do the stamps themselves cost?
#define ABRK_STAMP_WAITS 0
#else
#define ABRK_STAMP_WAITS 1
#endif
#define ABRK_STAMP(scr, id, wait)                                   \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    if ((wait) && ABRK_STAMP_WAITS) __builtin_amdgcn_s_waitcnt(0);  \
    (scr).tl[id] = __builtin_amdgcn_s_memrealtime();                \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
#else
#define ABRK_STAMP(scr, id, wait) ((void)0)
#endif
// Inside a block guarded by a condition that is UNIFORM over the launch (a controller parameter): keeps the block a real
// branch.  Left alone the compiler if-converts such blocks into vector selects - two v_cndmask per double, each as
// expensive as an FMA on gfx950 (tools/microbench/valu_rates.hip) - or, asked for 0 / 1 factors instead, parks the
// factors in scalar registers across the six-row kernel's persistent loop, which runs out of them (v_readlane per use).
// A scalar branch costs the vector pipeline nothing.
#define ABRK_UNIFORM_BLOCK() asm volatile("")

namespace abrk {

// ---------------------------------------------------------------- compile-time loops
template <int I>
using ic = std::integral_constant<int, I>;

template <class F, int... Is>
ABRK_INL void sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(ic<Is>{}), ...);
}
// sfor<N>(f): f(ic<0>), ..., f(ic<N-1>) - indices are constant expressions inside f
template <int N, class F>
ABRK_INL void sfor(F&& f) {
  sfor_impl(f, std::make_integer_sequence<int, (N > 0 ? N : 0)>{});
}

// ---------------------------------------------------------------- scalar math per type
template <class T>
struct Rm;
template <>
struct Rm<double> {
  // a*b + c with c held in scalar registers (see sincos)
  static ABRK_INL double fma_sc(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return ::fma(a, b, c);
#endif
  }
  // sin and cos of a joint angle.  Joint angles are moderate (|x| < 1e5), so the argument is
  // reduced with a 3-term Cody-Waite split of pi/2 under fma (exact products) and the two
  // kernels are degree-13/14 minimax polynomials on [-pi/4, pi/4]: ~25 fp64 instructions
  // against ~65 on the main path of the library sincos (whose double-double reduction serves
  // |x| up to 2^1024).  Absolute error <= 1.2e-16; larger arguments take the library path.
  static ABRK_INL void sincos(double x, double& s, double& c) {
    if (!sincos_in_range(x)) {
      ::sincos(x, &s, &c);
      return;
    }
    sincos_fast(x, s, c);
  }
  static ABRK_INL double acos(double x) { return ::acos(x); }
  static ABRK_INL bool sincos_in_range(double x) { return ::fabs(x) < 1.0e5; }
  static ABRK_INL bool sincos_tab_in_range(double x) { return ::fabs(x) < 1.0e5; }
  // the branch-free main path (valid for |x| < 1e5)
  static ABRK_INL void sincos_fast(double x, double& s, double& c) {
    const double n = ::rint(x * 0.6366197723675814);
    double r = ::fma(-n, 1.5707963267948966, x);
    r = ::fma(-n, 6.123233995736766e-17, r);
    r = ::fma(-n, -1.4973849048591698e-33, r);
    const double z = r * r;
    // Horner steps p*z + c with the coefficient c as a SCALAR source operand: the compiler's
    // default (v_fmac with c pre-loaded into the destination VGPR pair) costs two v_mov per
    // coefficient - a third of this routine's vector instructions.
    double ps = fma_sc(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma_sc(z, ps, 2.75573137070700676789e-06);
    ps = fma_sc(z, ps, -1.98412698298579493134e-04);
    ps = fma_sc(z, ps, 8.33333333332248946124e-03);
    ps = fma_sc(z, ps, -1.66666666666666324348e-01);
    const double sr = ::fma(z * r, ps, r);
    double pc = fma_sc(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma_sc(z, pc, -2.75573143513906633035e-07);
    pc = fma_sc(z, pc, 2.48015872894767294178e-05);
    pc = fma_sc(z, pc, -1.38888888888741095749e-03);
    pc = fma_sc(z, pc, 4.16666666666666019037e-02);
    const double cr = ::fma(z * z, pc, ::fma(z, -0.5, 1.0));
    const int k = (int)n;
    const bool swap = (k & 1) != 0;
    const double ss = swap ? cr : sr, cs = swap ? sr : cr;
    // quadrant signs straight into the sign bits: sin flips for k & 2, cos for (k + 1) & 2
    s = __builtin_bit_cast(double, __builtin_bit_cast(long long, ss) ^ ((long long)(k & 2) << 62));
    c = __builtin_bit_cast(double, __builtin_bit_cast(long long, cs) ^ ((long long)((k + 1) & 2) << 62));
  }
  // Table-driven form for the kernels that keep sin/cos of k 2pi/128 in LDS (`tab`: [128][2], abrk_sincos_table.h):
  // x = n (2pi/128) + r with |r| <= pi/128 (two-term Cody-Waite split, n D1 exact for |x| < 1e5), then
  //   sin x = S cos r + C sin r,  cos x = C cos r - S sin r
  // with Taylor polynomials of degree 7 / 6 in r (truncation < 1e-20): 16 fp64 + 3 integer instructions and one
  // 16-byte LDS read against 25 + 10 of the polynomial routine above.  Absolute error <= 3e-16.
  static ABRK_INL void sincos_tab(double x, const void* tab, double& s, double& c) {
    const double n = ::rint(x * kSinCosInvStep);
    double r = ::fma(-n, kSinCosD1, x);
    r = ::fma(-n, kSinCosD2, r);
    const int k = (int)n & (kSinCosN - 1);
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 sc = reinterpret_cast<const d2*>(tab)[k];
    const double z = r * r;
    double ps = fma_sc(z, -1.98412698412698412698e-04, 8.33333333333333333333e-03);
    ps = fma_sc(z, ps, -1.66666666666666666667e-01);
    const double ss = ::fma(z * r, ps, r);
    double pc = fma_sc(z, -1.38888888888888888889e-03, 4.16666666666666666667e-02);
    pc = fma_sc(z, pc, -0.5);
    const double cc = ::fma(z, pc, 1.0);
    s = ::fma(sc.x, cc, sc.y * ss);
    c = ::fma(sc.y, cc, -(sc.x * ss));
  }
  static ABRK_INL double sqrt(double x) { return ::sqrt(x); }
  static ABRK_INL double fabs(double x) { return ::fabs(x); }
  static ABRK_INL double fma(double a, double b, double c) { return ::fma(a, b, c); }
  static ABRK_INL double fmod(double a, double b) { return ::fmod(a, b); }
  static ABRK_INL double fmax(double a, double b) { return ::fmax(a, b); }
  static ABRK_INL double fmin(double a, double b) { return ::fmin(a, b); }
  static ABRK_INL double exp(double x) { return ::exp(x); }
  // 1/x and 1/sqrt(x) from the hardware seed (v_rcp_f64 / v_rsq_f64, ~2^-23 relative) plus two
  // Newton steps -> ~1-2 ulp in 5 / 9 instructions instead of the 11 + 12 of an IEEE divide and
  // sqrt.  x must be a normal positive number (pivots of SPD factorizations here).
  static ABRK_INL double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // seed error e0 <= 2^-23; y (1 + e + e^2) leaves e0^3: one cubic step reaches double precision
    const double y = __builtin_amdgcn_rcp(x);
    const double e = ::fma(-x, y, 1.0);
    return ::fma(y, ::fma(e, e, e), y);
#else
    return 1.0 / x;
#endif
  }
  static ABRK_INL double rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    // e = 1 - x y^2;  y (1 + e/2 + 3 e^2/8) leaves O(e^3)
    const double y = __builtin_amdgcn_rsq(x);
    const double e = ::fma(-(x * y), y, 1.0);
    return ::fma(y * e, ::fma(e, 0.375, 0.5), y);
#else
    return 1.0 / ::sqrt(x);
#endif
  }
  static constexpr double eps() { return 2.220446049250313e-16; }
  static constexpr double tiny() { return 1e-300; }
};
template <>
struct Rm<float> {
  static ABRK_INL void sincos(float x, float& s, float& c) { ::sincosf(x, &s, &c); }  // a custom routine measured no faster
  static ABRK_INL float acos(float x) { return ::acosf(x); }
  static ABRK_INL bool sincos_in_range(float) { return true; }  // the library routine serves every argument
  static ABRK_INL bool sincos_tab_in_range(float x) { return ::fabsf(x) < 1.0e5f; }
  static ABRK_INL void sincos_fast(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
  // fp32 form of the table-driven routine (tab: [128][2] floats in LDS): ~16 instructions against ~40 of the library's
  // sincosf; |x| < 1e5 (n < 2^21; the three-term split of 2 pi / 128 under fma keeps the reduced argument to ~2e-9),
  // Taylor degree 3 / 4 in r (|r| <= pi/128: truncation 7e-11).  Absolute error ~1e-7 = fp32 rounding.
  static ABRK_INL void sincos_tab(float x, const void* tab, float& s, float& c) {
    const float n = ::rintf(x * (float)kSinCosInvStep);
    constexpr float d1 = (float)(kSinCosD1 + kSinCosD2);
    constexpr float d2 = (float)((kSinCosD1 - (double)d1) + kSinCosD2);
    float r = ::fmaf(-n, d1, x);
    r = ::fmaf(-n, d2, r);
    const int k = (int)n & (kSinCosN - 1);
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 sc = reinterpret_cast<const f2*>(tab)[k];
    const float z = r * r;
    const float ss = ::fmaf(z * r, -1.66666667e-01f, r);
    const float cc = ::fmaf(z, ::fmaf(z, 4.16666667e-02f, -0.5f), 1.0f);
    s = ::fmaf(sc.x, cc, sc.y * ss);
    c = ::fmaf(sc.y, cc, -(sc.x * ss));
  }
  static ABRK_INL float sqrt(float x) { return ::sqrtf(x); }
  static ABRK_INL float fabs(float x) { return ::fabsf(x); }
  static ABRK_INL float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
  static ABRK_INL float fmod(float a, float b) { return ::fmodf(a, b); }
  static ABRK_INL float fmax(float a, float b) { return ::fmaxf(a, b); }
  static ABRK_INL float fmin(float a, float b) { return ::fminf(a, b); }
  static ABRK_INL float exp(float x) { return ::expf(x); }
  static ABRK_INL float rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float y = __builtin_amdgcn_rcpf(x);
    float e = ::fmaf(-x, y, 1.0f);
    return ::fmaf(y, e, y);
#else
    return 1.0f / x;
#endif
  }
  static ABRK_INL float rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float y = __builtin_amdgcn_rsqf(x);
    float e = ::fmaf(-0.5f * x * y, y, 0.5f);
    return ::fmaf(y, e, y);
#else
    return 1.0f / ::sqrtf(x);
#endif
  }
  static constexpr float eps() { return 1.1920929e-07f; }
  static constexpr float tiny() { return 1e-37f; }
};

// ---------------------------------------------------------------- 3-vectors
template <class T>
ABRK_INL void cross3(const T (&a)[3], const T (&b)[3], T (&o)[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
template <class T>
ABRK_INL T dot3(const T (&a)[3], const T (&b)[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// acc + a.b as an fma chain (one op fewer than dot3 + add; no reassociation is implied
// elsewhere: the kernels are compiled with strict IEEE semantics)
template <class T>
ABRK_INL T fdot3(T acc, const T (&a)[3], const T (&b)[3]) {
  return Rm<T>::fma(a[2], b[2], Rm<T>::fma(a[1], b[1], Rm<T>::fma(a[0], b[0], acc)));
}
template <class T>
ABRK_INL void matvec3(const T (&M)[9], const T (&v)[3], T (&o)[3]) {
  o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
  o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
  o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}

// ---------------------------------------------------------------- arm policies
// constexpr 3x4 affine product element: (X*Y)[e], e = r*4+c
constexpr double aff_mul(const double* X, const double* Y, int e) {
  int r = e / 4, c = e % 4;
  double s = 0.0;
  for (int k = 0; k < 3; k++) s += X[r * 4 + k] * Y[k * 4 + c];
  if (c == 3) s += X[r * 4 + 3];
  return s;
}
constexpr bool aff_is_orthogonal(const double* X) {
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += X[k * 4 + a] * X[k * 4 + b];
      if (s != (a == b ? 1.0 : 0.0)) return false;
    }
  // proper rotation (det = +1): only then is P Zhat P^-1 the cross product with the joint axis
  const double det = X[0] * (X[5] * X[10] - X[6] * X[9]) - X[1] * (X[4] * X[10] - X[6] * X[8]) +
                     X[2] * (X[4] * X[9] - X[5] * X[8]);
  return det > 0.0;
}

// Built-in arm: every constant is a constant expression.  Derived static transforms:
//   J0   = A0 * AJ[0]                joint_0 frame (independent of q)
//   S[i] = B[i] * AJ[i+1]            joint_i (after Rz(q_i)) -> joint_{i+1}
//   B[i]                             joint_i (after Rz(q_i)) -> link_{i+1}  (COM frame)
//   BE   = B[N-1] * E                joint_{N-1} (after Rz) -> EE
template <class Tab>
struct StaticArm {
  static constexpr int N = Tab::N;
  static constexpr int NL = Tab::NL;
  static constexpr bool kStatic = true;
  static constexpr bool kHasEE = Tab::kHasEE;
  static constexpr double J0(int e) { return aff_mul(Tab::A0, Tab::AJ[0], e); }
  static constexpr double S(int i, int e) { return aff_mul(Tab::B[i], Tab::AJ[i + 1 < N ? i + 1 : i], e); }
  static constexpr double Bm(int i, int e) { return Tab::B[i][e]; }
  static constexpr double BE(int e) { return kHasEE ? aff_mul(Tab::B[N - 1], Tab::E, e) : Tab::B[N - 1][e]; }
  static constexpr double A0m(int e) { return Tab::A0[e]; }
  static constexpr double MD(int l, int r) { return Tab::MD[l][r]; }
  // angular inertia seen by joint pair with max index m: sum of links l > m (l < NL)
  static constexpr double Isuf(int m, int r) {
    double s = 0.0;
    for (int l = m + 1; l < NL && l <= N; l++) s += Tab::MD[l][3 + r];
    return s;
  }
  static constexpr bool compute_ortho() {
    if (!aff_is_orthogonal(Tab::A0)) return false;
    for (int i = 0; i < N; i++)
      if (!aff_is_orthogonal(Tab::AJ[i]) || !aff_is_orthogonal(Tab::B[i])) return false;
    return true;
  }
  static constexpr bool kOrtho = compute_ortho();
  // every frame rotation the chain can report (links, joints, EE) is a product of exact rotations
  static constexpr bool kOrthoFrames = kOrtho && (!kHasEE || aff_is_orthogonal(Tab::E));
  // every static block rotates about z only: all joint axes are the world z axis and the z row of every
  // linear Jacobian is identically zero (twojoint, threejoint, onejoint)
  static constexpr bool aff_is_planar(const double* X) {
    return X[2] == 0.0 && X[6] == 0.0 && X[8] == 0.0 && X[9] == 0.0 && X[10] == 1.0;
  }
  static constexpr bool compute_planar() {
    if (!aff_is_planar(Tab::A0) || (kHasEE && !aff_is_planar(Tab::E))) return false;
    for (int i = 0; i < N; i++)
      if (!aff_is_planar(Tab::AJ[i]) || !aff_is_planar(Tab::B[i])) return false;
    return true;
  }
  static constexpr bool kPlanar = compute_planar();
  // joint i's frame is an exact rotation iff every static block before it is (Jaco2: joints 0-2; its later
  // rotation constants are rounded, arms/jaco2/config.py:189-273): W_i is then the cross product with z_i
  // and needs no 3x3 matrix.
  static constexpr bool ortho_joint(int i) {
    if (!aff_is_orthogonal(Tab::A0)) return false;
    for (int k = 0; k <= i; k++) {
      if (!aff_is_orthogonal(Tab::AJ[k])) return false;
      if (k < i && !aff_is_orthogonal(Tab::B[k])) return false;
    }
    return true;
  }
};

// User arm: same derived tables, filled on the host (abrk_host.cpp) in the kernel's
// arithmetic type and passed by value as a kernel argument (scalar loads, SGPR operands).
template <int NJ, class T>
struct RtArm {
  static constexpr int N = NJ;
  static constexpr bool kStatic = false;
  static constexpr bool kOrtho = false;  // always differentiate the general affine chain
  static constexpr bool kOrthoFrames = false;
  static constexpr bool kPlanar = false;
  static constexpr bool ortho_joint(int) { return false; }
  int NL;
  T J0v[12];
  T Sv[NJ][12];
  T Bv[NJ][12];
  T BEv[12];
  T A0v[12];
  T MDv[NJ + 1][6];
  T Isufv[NJ + 1][3];
};

// Accessors: value of a table element as type T (constant expression for static arms).
#define ABRK_ACC(NAME, STATIC_EXPR, RT_EXPR)                                 \
  template <class A, class T, int... Ix>                                     \
  struct NAME {                                                              \
    static ABRK_INL T get(const A& a) {                                      \
      if constexpr (A::kStatic) {                                            \
        constexpr double v = STATIC_EXPR;                                    \
        return T(v);                                                         \
      } else {                                                               \
        return RT_EXPR;                                                      \
      }                                                                      \
    }                                                                        \
    static constexpr double cv() {                                           \
      if constexpr (A::kStatic) return STATIC_EXPR;                          \
      else return 0.5; /* "not a foldable constant" */                       \
    }                                                                        \
  };

template <int... Ix>
struct idx_pack {
  static constexpr int v[sizeof...(Ix) > 0 ? sizeof...(Ix) : 1] = {Ix...};
};

ABRK_ACC(AccJ0, A::J0(idx_pack<Ix...>::v[0]), a.J0v[idx_pack<Ix...>::v[0]])
ABRK_ACC(AccS, A::S(idx_pack<Ix...>::v[0], idx_pack<Ix...>::v[1]), a.Sv[idx_pack<Ix...>::v[0]][idx_pack<Ix...>::v[1]])
ABRK_ACC(AccB, A::Bm(idx_pack<Ix...>::v[0], idx_pack<Ix...>::v[1]), a.Bv[idx_pack<Ix...>::v[0]][idx_pack<Ix...>::v[1]])
ABRK_ACC(AccBE, A::BE(idx_pack<Ix...>::v[0]), a.BEv[idx_pack<Ix...>::v[0]])
ABRK_ACC(AccA0, A::A0m(idx_pack<Ix...>::v[0]), a.A0v[idx_pack<Ix...>::v[0]])
ABRK_ACC(AccMD, A::MD(idx_pack<Ix...>::v[0], idx_pack<Ix...>::v[1]), a.MDv[idx_pack<Ix...>::v[0]][idx_pack<Ix...>::v[1]])
ABRK_ACC(AccIsuf, A::Isuf(idx_pack<Ix...>::v[0], idx_pack<Ix...>::v[1]), a.Isufv[idx_pack<Ix...>::v[0]][idx_pack<Ix...>::v[1]])

// acc + coef*x with compile-time folding of coef in {0, 1, -1}.  Start accumulators at
// -0.0 (additive identity that LLVM folds exactly under strict IEEE).
template <class Acc, class A, class T>
ABRK_INL T cfma(const A& a, T x, T acc) {
  if constexpr (A::kStatic) {
    constexpr double c = Acc::cv();
    if constexpr (c == 0.0)
      return acc;
    else if constexpr (c == 1.0)
      return acc + x;
    else if constexpr (c == -1.0)
      return acc - x;
    else
      return Rm<T>::fma(T(c), x, acc);
  } else {
    return Rm<T>::fma(Acc::get(a), x, acc);
  }
}
// acc + coef (pure constant add, skipped when 0)
template <class Acc, class A, class T>
ABRK_INL T cadd(const A& a, T acc) {
  if constexpr (A::kStatic) {
    constexpr double c = Acc::cv();
    if constexpr (c == 0.0)
      return acc;
    else
      return acc + T(c);
  } else {
    return acc + Acc::get(a);
  }
}

// (R, o) * static affine C  ->  (R2, o2);   C's elements come from accessor template AccT<A,T,...,e>
#define ABRK_FRAME_MUL(FN, ACC)                                                              \
  template <class A, class T, int... Pre>                                                    \
  ABRK_INL void FN(const A& a, const T (&R)[9], const T (&o)[3], T (&R2)[9], T (&o2)[3]) {   \
    sfor<3>([&](auto r) ABRK_LAMBDA {                                                        \
      sfor<3>([&](auto c) ABRK_LAMBDA {                                                      \
        T acc = T(-0.0);                                                                     \
        sfor<3>([&](auto k) ABRK_LAMBDA {                                                    \
          acc = cfma<ACC<A, T, Pre..., k() * 4 + c()>>(a, R[r() * 3 + k()], acc);            \
        });                                                                                  \
        R2[r() * 3 + c()] = acc;                                                             \
      });                                                                                    \
      T acc = o[r()];                                                                        \
      sfor<3>([&](auto k) ABRK_LAMBDA {                                                      \
        acc = cfma<ACC<A, T, Pre..., k() * 4 + 3>>(a, R[r() * 3 + k()], acc);                \
      });                                                                                    \
      o2[r()] = acc;                                                                         \
    });                                                                                      \
  }                                                                                          \
  /* translation only: o2 = o + R * C_t */                                                   \
  template <class A, class T, int... Pre>                                                    \
  ABRK_INL void FN##_pt(const A& a, const T (&R)[9], const T (&o)[3], T (&o2)[3]) {          \
    sfor<3>([&](auto r) ABRK_LAMBDA {                                                        \
      T acc = o[r()];                                                                        \
      sfor<3>([&](auto k) ABRK_LAMBDA {                                                      \
        acc = cfma<ACC<A, T, Pre..., k() * 4 + 3>>(a, R[r() * 3 + k()], acc);                \
      });                                                                                    \
      o2[r()] = acc;                                                                         \
    });                                                                                      \
  }                                                                                          \
  /* rotation only: R2 = R * C_R */                                                          \
  template <class A, class T, int... Pre>                                                    \
  ABRK_INL void FN##_rot(const A& a, const T (&R)[9], T (&R2)[9]) {                          \
    sfor<3>([&](auto r) ABRK_LAMBDA {                                                        \
      sfor<3>([&](auto c) ABRK_LAMBDA {                                                      \
        T acc = T(-0.0);                                                                     \
        sfor<3>([&](auto k) ABRK_LAMBDA {                                                    \
          acc = cfma<ACC<A, T, Pre..., k() * 4 + c()>>(a, R[r() * 3 + k()], acc);            \
        });                                                                                  \
        R2[r() * 3 + c()] = acc;                                                             \
      });                                                                                    \
    });                                                                                      \
  }

ABRK_FRAME_MUL(mulS, AccS)
ABRK_FRAME_MUL(mulB, AccB)
ABRK_FRAME_MUL(mulBE, AccBE)

// ---------------------------------------------------------------- per-row joint state
template <class A, class T>
struct Joints {
  static constexpr int N = A::N;
  T z[N][3];                      // z_i = P_i e_z            (J_orientation[i], base_config.py:565-580)
  T o[N][3];                      // origin of T(joint_i)
  T W[A::kOrtho ? 1 : N][9];      // W_i = P_i Zhat P_i^-1    (general affine chain only)
};

// out = W_I v
template <int I, class A, class T>
ABRK_INL void wapply(const Joints<A, T>& jt, const T (&v)[3], T (&out)[3]) {
  if constexpr (A::ortho_joint(I))
    cross3(jt.z[I], v, out);
  else
    matvec3(jt.W[I], v, out);
}

// Optional capture of an arbitrary frame (runtime id, uniform across the wavefront).
template <class T>
struct FrameCap {
  int frame;   // link_i -> 2i, joint_i -> 2i+1, EE -> 2N+1
  T R[9];
  T o[3];
};
struct NoCap {};

// Forward kinematics over the joint chain (arms/*/config.py `_calc_T`, e.g.
// ur5/config.py:301-339).  For i = 0..N-1: records joint_i (z, o, W), rotates by q_i,
// calls on_link(ic<i+1>, p) with the COM position of link_{i+1}.  Leaves in (XR, xo) the
// rotation of joint_{N-1} after its Rz and its origin - the EE hangs off that.
// Where sin/cos of the joint angles come from: computed in the chain (ScCompute), or handed in ready (ScUse: evaluated
// ahead of the chain by sincos_all / sincos_all_tab).
struct ScCompute {
  template <int I, class T>
  ABRK_INL void get(T q, T& s, T& c) const { Rm<T>::sincos(q, s, c); }
};
template <class T, int N>
struct ScUse {
  const T (&sv)[N][2];
  template <int I>
  ABRK_INL void get(T, T& s, T& c) const {
    s = sv[I][0];
    c = sv[I][1];
  }
};

// sin/cos through the LDS table of the calling kernel (Rm<double>::sincos_tab); fp32 rows keep the library routine
struct ScTab {
  const void* tab;  // [128][2] of the kernel's arithmetic type
};
// sin/cos of all joint angles ahead of the chain: one straight-line block of N independent polynomial
// evaluations (the per-joint range check would split the forward kinematics into N basic blocks and serialise
// them); angles beyond the fast routine's range are redone by the library afterwards (one rarely taken branch)
template <int N, class T>
ABRK_INL void sincos_all(const T (&q)[N], T (&sv)[N][2]) {
  bool all_in = true;
  sfor<N>([&](auto i) ABRK_LAMBDA {
    all_in = all_in && Rm<T>::sincos_in_range(q[i()]);
    Rm<T>::sincos_fast(q[i()], sv[i()][0], sv[i()][1]);
  });
  if (!all_in) sfor<N>([&](auto i) ABRK_LAMBDA { Rm<T>::sincos(q[i()], sv[i()][0], sv[i()][1]); });
}
template <int N, class T>
ABRK_INL void sincos_all_tab(const T (&q)[N], T (&sv)[N][2], const void* tab) {
  bool all_in = true;
  sfor<N>([&](auto i) ABRK_LAMBDA {
    all_in = all_in && Rm<T>::sincos_tab_in_range(q[i()]);
    Rm<T>::sincos_tab(q[i()], tab, sv[i()][0], sv[i()][1]);
  });
  if (!all_in) sfor<N>([&](auto i) ABRK_LAMBDA { Rm<T>::sincos(q[i()], sv[i()][0], sv[i()][1]); });
}

template <class A, class T, class Cap, class LinkFn, class Sc = ScCompute>
ABRK_INL void fk_forward(const A& arm, const T (&q)[A::N], Joints<A, T>& jt, T (&XR)[9], T (&xo)[3],
                         Cap& cap, LinkFn&& on_link, const Sc& scp = Sc{}) {
  constexpr int N = A::N;
  T Rj[9], oj[3];
  sfor<3>([&](auto r) ABRK_LAMBDA {
    sfor<3>([&](auto c) ABRK_LAMBDA { Rj[r() * 3 + c()] = AccJ0<A, T, r() * 4 + c()>::get(arm); });
    oj[r()] = AccJ0<A, T, r() * 4 + 3>::get(arm);
  });
  if constexpr (!std::is_same<Cap, NoCap>::value) {
    if (cap.frame == 0) {  // link0 = A0
      sfor<3>([&](auto r) ABRK_LAMBDA {
        sfor<3>([&](auto c) ABRK_LAMBDA { cap.R[r() * 3 + c()] = AccA0<A, T, r() * 4 + c()>::get(arm); });
        cap.o[r()] = AccA0<A, T, r() * 4 + 3>::get(arm);
      });
    }
  }
  sfor<N>([&](auto i) ABRK_LAMBDA {
    constexpr int I = i();
    // ---- record joint_I
    sfor<3>([&](auto r) ABRK_LAMBDA {
      jt.z[I][r()] = Rj[r() * 3 + 2];
      jt.o[I][r()] = oj[r()];
    });
    if constexpr (!A::ortho_joint(I)) {
      // rows 0,1 of P^-1 via the dual basis: r0 = (c1 x c2)/det, r1 = (c2 x c0)/det
      T c0[3] = {Rj[0], Rj[3], Rj[6]}, c1[3] = {Rj[1], Rj[4], Rj[7]}, c2[3] = {Rj[2], Rj[5], Rj[8]};
      T r0[3], r1[3];
      cross3(c1, c2, r0);
      cross3(c2, c0, r1);
      T idet = Rm<T>::rcp(dot3(c0, r0));
      sfor<3>([&](auto a) ABRK_LAMBDA {
        sfor<3>([&](auto b) ABRK_LAMBDA {
          jt.W[I][a() * 3 + b()] = (c1[a()] * r0[b()] - c0[a()] * r1[b()]) * idet;
        });
      });
    }
    if constexpr (!std::is_same<Cap, NoCap>::value) {
      if (cap.frame == 2 * I + 1) {
        sfor<9>([&](auto e) ABRK_LAMBDA { cap.R[e()] = Rj[e()]; });
        sfor<3>([&](auto r) ABRK_LAMBDA { cap.o[r()] = oj[r()]; });
      }
    }
    // ---- rotate about local z by q_I:  X = joint_I * Rz(q_I)
    T s, c;
    scp.template get<I>(q[I], s, c);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      T a0 = Rj[r() * 3 + 0], a1 = Rj[r() * 3 + 1];
      XR[r() * 3 + 0] = c * a0 + s * a1;
      XR[r() * 3 + 1] = c * a1 - s * a0;
      XR[r() * 3 + 2] = Rj[r() * 3 + 2];
    });
    sfor<3>([&](auto r) ABRK_LAMBDA { xo[r()] = oj[r()]; });
    // ---- COM of link_{I+1}
    T p[3];
    mulB_pt<A, T, I>(arm, XR, xo, p);
    if constexpr (!std::is_same<Cap, NoCap>::value) {
      if (cap.frame == 2 * (I + 1)) {
        mulB_rot<A, T, I>(arm, XR, cap.R);
        sfor<3>([&](auto r) ABRK_LAMBDA { cap.o[r()] = p[r()]; });
      }
    }
    on_link(ic<I + 1>{}, p);
    // ---- next joint frame
    if constexpr (I + 1 < N) {
      T R2[9], o2[3];
      mulS<A, T, I>(arm, XR, xo, R2, o2);
      sfor<9>([&](auto e) ABRK_LAMBDA { Rj[e()] = R2[e()]; });
      sfor<3>([&](auto r) ABRK_LAMBDA { oj[r()] = o2[r()]; });
    }
  });
  if constexpr (!std::is_same<Cap, NoCap>::value) {
    if (cap.frame == 2 * N + 1) mulBE<A, T>(arm, XR, xo, cap.R, cap.o);
  }
}

// ---------------------------------------------------------------- dynamics accumulation
// Lower-triangular index of a symmetric N x N matrix
constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// CMODE_VEC: C(q,dq) dq accumulated link by link in the same pass as M (general chains; orthogonal chains of the OSC
// kernels take the recursion of rne_forward_step / rne_backward instead); CMODE_MAT: the full Christoffel matrix
enum { CMODE_NONE = 0, CMODE_VEC = 1, CMODE_MAT = 2 };

template <class A, class T, int CMODE>
struct Dyn {
  static constexpr int N = A::N;
  T Ms[N * (N + 1) / 2];                  // M, lower triangle            (base_config.py:594-645)
  T gz[N];                                // sum_l m_l,z * dp_l,z/dq_i    (g = -9.81 gz, base_config.py:417-468)
  T Cm[CMODE == CMODE_MAT ? N * N : 1];   // Christoffel matrix           (base_config.py:678-727)
  T cv[CMODE == CMODE_VEC ? N : 1];  // C(q,dq) dq
  T om[CMODE != CMODE_NONE ? N : 1][3];   // omega_j = sum_{k<j} dq_k z_k  (orthogonal chains, matrix mode)
  // CMODE_VEC on orthogonal chains: kinematic state of the body the current link belongs to -
  // angular velocity, bias angular acceleration, bias acceleration of the last joint origin
  T bw[3], bal[3], bao[3];
};
// the recursive Coriolis-vector path (below) replaces the omega prefix sums
template <class A, int CM>
constexpr bool kRecursiveC = (CM == CMODE_VEC) && A::kOrtho;

// does link L carry linear / any mass?  (static arms: compile time; user arms: assume yes)
template <class A, int L>
constexpr bool link_has_linear_mass() {
  if constexpr (A::kStatic)
    return A::MD(L, 0) != 0.0 || A::MD(L, 1) != 0.0 || A::MD(L, 2) != 0.0;
  else
    return true;
}

// Omega_J v = sum_{k<J} dq_k W_k v
template <int J, class A, class T, int CM>
ABRK_INL void omega_apply(const Joints<A, T>& jt, const Dyn<A, T, CM>& d, const T (&dq)[A::N], const T (&v)[3],
                          T (&out)[3]) {
  if constexpr (J == 0) {
    out[0] = out[1] = out[2] = T(0);
  } else if constexpr (A::kOrtho) {
    cross3(d.om[J], v, out);
  } else {
    T acc[3] = {T(-0.0), T(-0.0), T(-0.0)};
    sfor<J>([&](auto k) ABRK_LAMBDA {
      T t[3];
      wapply<k()>(jt, v, t);
      sfor<3>([&](auto r) ABRK_LAMBDA { acc[r()] += dq[k()] * t[r()]; });
    });
    sfor<3>([&](auto r) ABRK_LAMBDA { out[r()] = acc[r()]; });
  }
}

// Contribution of link L (COM at p, moved by joints 0..L-1) to M, g and C.
//   e_i = dp/dq_i = W_i (p - o_i);   M_ij += sum_r m_r e_i[r] e_j[r];   gz_i += m_z e_i[2]
//   linear Coriolis:  C_lin = sum_l E_l^T D_l Edot_l  with
//       Edot[:,j] = d/dt e_j = W_j (sum_{k>=j} e_k dq_k) + Omega_j e_j
//   (the Christoffel combination 1/2(M_kj,i + M_ki,j - M_ij,k) dq_i collapses to this
//    because d2p/dq_i dq_k is symmetric - see DESIGN.md)
template <int L, class A, class T, int CM>
ABRK_INL void link_accumulate(const A& arm, const Joints<A, T>& jt, const T (&dq)[A::N], const T (&p)[3],
                              Dyn<A, T, CM>& d) {
  constexpr int NJ = (L < A::N ? L : A::N);
  bool live = true;
  if constexpr (!A::kStatic) live = (L < arm.NL);
  if constexpr (A::kStatic) {
    if constexpr (!(L < A::NL) || !link_has_linear_mass<A, L>()) return;
  }
  if (!live) return;
  T e[NJ][3];
  sfor<NJ>([&](auto i) ABRK_LAMBDA {
    T dlt[3] = {p[0] - jt.o[i()][0], p[1] - jt.o[i()][1], p[2] - jt.o[i()][2]};
    wapply<i()>(jt, dlt, e[i()]);
  });
  T m0 = AccMD<A, T, L, 0>::get(arm), m1 = AccMD<A, T, L, 1>::get(arm), m2 = AccMD<A, T, L, 2>::get(arm);
  sfor<NJ>([&](auto i) ABRK_LAMBDA {
    T me[3] = {m0 * e[i()][0], m1 * e[i()][1], m2 * e[i()][2]};  // D_l e_i (transient)
    d.gz[i()] += me[2];
    sfor<i() + 1>([&](auto j) ABRK_LAMBDA { d.Ms[tri(i(), j())] = fdot3(d.Ms[tri(i(), j())], me, e[j()]); });
  });
  if constexpr (kRecursiveC<A, CM>) {
    // C(q,dq) dq, linear part = sum_l E_l^T D_l a_l with a_l the bias (qdd = 0) acceleration of the COM,
    // from the body recursion instead of per-link suffix sums:
    //   a(p) = a(o) + alpha x (p - o) + omega x (omega x (p - o)),   o = origin of the last joint
    T dl[3] = {p[0] - jt.o[NJ - 1][0], p[1] - jt.o[NJ - 1][1], p[2] - jt.o[NJ - 1][2]};
    T t[3], ad[3], wt[3];
    cross3(d.bw, dl, t);
    cross3(d.bal, dl, ad);
    cross3(d.bw, t, wt);
    T ma[3] = {m0 * (d.bao[0] + ad[0] + wt[0]), m1 * (d.bao[1] + ad[1] + wt[1]), m2 * (d.bao[2] + ad[2] + wt[2])};
    sfor<NJ>([&](auto k) ABRK_LAMBDA { d.cv[k()] = fdot3(d.cv[k()], e[k()], ma); });
  } else if constexpr (CM != CMODE_NONE) {
    T s[3] = {T(-0.0), T(-0.0), T(-0.0)};
    T a[3] = {T(-0.0), T(-0.0), T(-0.0)};  // COM bias acceleration  Edot dq
    T ed[CM == CMODE_MAT ? NJ : 1][3];
    sfor<NJ>([&](auto jr) ABRK_LAMBDA {
      constexpr int j = NJ - 1 - jr();
      sfor<3>([&](auto r) ABRK_LAMBDA { s[r()] += e[j][r()] * dq[j]; });
      T t1[3], t2[3];
      wapply<j>(jt, s, t1);
      omega_apply<j>(jt, d, dq, e[j], t2);
      sfor<3>([&](auto r) ABRK_LAMBDA {
        T edj = t1[r()] + t2[r()];
        a[r()] += dq[j] * edj;
        if constexpr (CM == CMODE_MAT) ed[j][r()] = edj;
      });
    });
    if constexpr (CM == CMODE_VEC) {
      T ma[3] = {m0 * a[0], m1 * a[1], m2 * a[2]};
      sfor<NJ>([&](auto k) ABRK_LAMBDA { d.cv[k()] = fdot3(d.cv[k()], e[k()], ma); });
    } else {
      sfor<NJ>([&](auto j) ABRK_LAMBDA {
        T med[3] = {m0 * ed[j()][0], m1 * ed[j()][1], m2 * ed[j()][2]};
        sfor<NJ>([&](auto k) ABRK_LAMBDA { d.Cm[k() * A::N + j()] = fdot3(d.Cm[k() * A::N + j()], e[k()], med); });
      });
    }
  }
}

// Angular Coriolis vector of link L on orthogonal chains (CMODE_VEC): with the world-frame diagonal D_L
// of the reference (base_config.py:628) the Christoffel form of M^w = sum_l Jw_l^T D_l Jw_l reduces to
//   c^w_k = z_k . sum_{l>k} n_l,    n_l = D_l alpha_l + (D_l omega_l) x omega_l
// (derivation in DESIGN.md; note the sign differs from Euler's equation because D_l is not rotated).
template <int L, class A, class T, int CM>
ABRK_INL void angular_link_coriolis(const A& arm, const Joints<A, T>& jt, Dyn<A, T, CM>& d) {
  if constexpr (kRecursiveC<A, CM>) {
    constexpr int NJ = (L < A::N ? L : A::N);
    bool live = true;
    if constexpr (A::kStatic) {
      if constexpr (!(L < A::NL) || (A::MD(L, 3) == 0.0 && A::MD(L, 4) == 0.0 && A::MD(L, 5) == 0.0)) return;
    } else {
      live = (L < arm.NL);
    }
    if (!live) return;
    T I0 = AccMD<A, T, L, 3>::get(arm), I1 = AccMD<A, T, L, 4>::get(arm), I2 = AccMD<A, T, L, 5>::get(arm);
    T Lw[3] = {I0 * d.bw[0], I1 * d.bw[1], I2 * d.bw[2]};
    T n[3];
    cross3(Lw, d.bw, n);
    n[0] = Rm<T>::fma(I0, d.bal[0], n[0]);
    n[1] = Rm<T>::fma(I1, d.bal[1], n[1]);
    n[2] = Rm<T>::fma(I2, d.bal[2], n[2]);
    sfor<NJ>([&](auto k) ABRK_LAMBDA { d.cv[k()] = fdot3(d.cv[k()], jt.z[k()], n); });
  }
}

// Advance the body state across joint L-1 (just recorded): first move the bias acceleration to the new
// joint origin using the OLD body's (omega, alpha), then add the joint's contribution.
template <int L, class A, class T, int CM>
ABRK_INL void body_advance(const Joints<A, T>& jt, const T (&dq)[A::N], Dyn<A, T, CM>& d) {
  if constexpr (kRecursiveC<A, CM> && L - 1 < A::N) {
    constexpr int i = L - 1;
    if constexpr (i == 0) {
      sfor<3>([&](auto r) ABRK_LAMBDA { d.bw[r()] = d.bal[r()] = d.bao[r()] = T(0); });
    } else {
      T d2[3] = {jt.o[i][0] - jt.o[i - 1][0], jt.o[i][1] - jt.o[i - 1][1], jt.o[i][2] - jt.o[i - 1][2]};
      T t[3], ad[3], wt[3];
      cross3(d.bw, d2, t);
      cross3(d.bal, d2, ad);
      cross3(d.bw, t, wt);
      sfor<3>([&](auto r) ABRK_LAMBDA { d.bao[r()] += ad[r()] + wt[r()]; });
    }
    T zd[3];
    cross3(d.bw, jt.z[i], zd);  // d/dt z_i = omega x z_i
    sfor<3>([&](auto r) ABRK_LAMBDA {
      d.bal[r()] = Rm<T>::fma(zd[r()], dq[i], d.bal[r()]);
      d.bw[r()] = Rm<T>::fma(jt.z[i][r()], dq[i], d.bw[r()]);
    });
  }
}

template <class A, class T, int CM>
ABRK_INL void dyn_init(Dyn<A, T, CM>& d) {
  constexpr int N = A::N;
  sfor<N*(N + 1) / 2>([&](auto e) ABRK_LAMBDA { d.Ms[e()] = T(-0.0); });  // -0.0: exact additive identity, lets the first fma fold to a mul
  sfor<N>([&](auto i) ABRK_LAMBDA { d.gz[i()] = T(-0.0); });
  if constexpr (CM == CMODE_MAT) sfor<N * N>([&](auto e) ABRK_LAMBDA { d.Cm[e()] = T(-0.0); });
  if constexpr (CM == CMODE_VEC) sfor<N>([&](auto e) ABRK_LAMBDA { d.cv[e()] = T(-0.0); });
}

// weighted dot  sum_r Isuf(m,r) a[r] b[r]
template <int Mx, class A, class T>
ABRK_INL T idot(const A& arm, const T (&a)[3], const T (&b)[3], T acc = T(-0.0)) {
  sfor<3>([&](auto r) ABRK_LAMBDA { acc = cfma<AccIsuf<A, T, Mx, r()>>(arm, a[r()] * b[r()], acc); });
  return acc;
}

// Angular (world-frame inertia diagonal, base_config.py:628 quirk) parts of M and C, added
// once every z_i is known:  M^w_ij = z_i^T Ibar_max(i,j) z_j,  Ibar_m = sum_{l>m} diag(I_l).
template <class A, class T, int CM>
ABRK_INL void angular_finish(const A& arm, const Joints<A, T>& jt, const T (&dq)[A::N], Dyn<A, T, CM>& d) {
  constexpr int N = A::N;
  sfor<N>([&](auto i) ABRK_LAMBDA {
    sfor<i() + 1>([&](auto j) ABRK_LAMBDA { d.Ms[tri(i(), j())] = idot<i()>(arm, jt.z[i()], jt.z[j()], d.Ms[tri(i(), j())]); });
  });
  // (orthogonal chains in CMODE_VEC: the angular Coriolis vector is accumulated link by link in
  //  angular_link_coriolis during the forward pass)
  if constexpr (CM == CMODE_VEC && !A::kOrtho) {
    // general affine chain: c^w_k = zdot_k.y_k + z_k.y'_k - sum_{i>k} dq_i (W_k z_i).y_i
    //   y_k  = sum_i Ibar_max(k,i) o (z_i dq_i),   y'_k = sum_i Ibar_max(k,i) o (zdot_i dq_i)
    T zd[N][3];
    sfor<N>([&](auto k) ABRK_LAMBDA { omega_apply<k()>(jt, d, dq, jt.z[k()], zd[k()]); });
    T y[N][3], yp[N][3];
    sfor<N>([&](auto k) ABRK_LAMBDA {
      sfor<3>([&](auto r) ABRK_LAMBDA {
        T ay = T(-0.0), ayp = T(-0.0);
        sfor<N>([&](auto i) ABRK_LAMBDA {
          constexpr int m = (k() > i() ? k() : i());
          ay = cfma<AccIsuf<A, T, m, r()>>(arm, jt.z[i()][r()] * dq[i()], ay);
          ayp = cfma<AccIsuf<A, T, m, r()>>(arm, zd[i()][r()] * dq[i()], ayp);
        });
        y[k()][r()] = ay;
        yp[k()][r()] = ayp;
      });
    });
    sfor<N>([&](auto k) ABRK_LAMBDA {
      T acc = dot3(zd[k()], y[k()]) + dot3(jt.z[k()], yp[k()]);
      sfor<N - 1 - k()>([&](auto ii) ABRK_LAMBDA {
        constexpr int i = k() + 1 + ii();
        T x[3];
        wapply<k()>(jt, jt.z[i], x);
        acc -= dq[i] * dot3(x, y[i]);
      });
      d.cv[k()] += acc;
    });
  }
  if constexpr (CM == CMODE_MAT) {
    // Christoffel symbols of M^w with dz_a/dq_c = [c<a] W_c z_a:
    //   dM_ab/dq_c = [c<a] (W_c z_a)^T Ibar z_b + [c<b] z_a^T Ibar (W_c z_b),  Ibar = Ibar_max(a,b)
    T X[N][N][3];  // X[c][a] = W_c z_a  (c < a)
    sfor<N>([&](auto c) ABRK_LAMBDA {
      sfor<N - 1 - c()>([&](auto aa) ABRK_LAMBDA {
        constexpr int a = c() + 1 + aa();
        wapply<c()>(jt, jt.z[a], X[c()][a]);
      });
    });
    auto dM = [&](auto a, auto b, auto c) ABRK_LAMBDA -> T {
      constexpr int m = (a() > b() ? a() : b());
      T acc = T(-0.0);
      if constexpr (c() < a()) acc += idot<m>(arm, X[c()][a()], jt.z[b()]);
      if constexpr (c() < b()) acc += idot<m>(arm, jt.z[a()], X[c()][b()]);
      return acc;
    };
    sfor<N>([&](auto k) ABRK_LAMBDA {
      sfor<N>([&](auto j) ABRK_LAMBDA {
        T acc = T(-0.0);
        sfor<N>([&](auto i) ABRK_LAMBDA { acc += (dM(k, j, i) + dM(k, i, j) - dM(i, j, k)) * dq[i()]; });
        d.Cm[k() * N + j()] += T(0.5) * acc;
      });
    });
  }
}

// omega prefix sums (orthogonal chains): om[j] = sum_{k<j} dq_k z_k; must be current before
// link_accumulate<L> uses joints < L, so it is advanced inside the link visitor.
template <int L, class A, class T, int CM>
ABRK_INL void omega_advance(const Joints<A, T>& jt, const T (&dq)[A::N], Dyn<A, T, CM>& d) {
  if constexpr (CM != CMODE_NONE && !kRecursiveC<A, CM> && A::kOrtho && L - 1 < A::N) {
    constexpr int j = L - 1;  // joint that was just recorded
    if constexpr (j == 0) {
      d.om[0][0] = d.om[0][1] = d.om[0][2] = T(0);
    }
    if constexpr (j + 1 < A::N) {
      sfor<3>([&](auto r) ABRK_LAMBDA { d.om[j + 1][r()] = d.om[j][r()] + dq[j] * jt.z[j][r()]; });
    }
  }
}

// ---------------------------------------------------------------- Coriolis vector from the kept frames
// Per-lane scratch that carries the link wrenches from the forward to the backward sweep of coriolis_rne.
// RegScratch: plain arrays (host check build, small arms).  The GPU kernels use LdsScratch (abrk_kernels.h): 6 values
// per link parked in the wavefront's LDS slab, which keeps 12 N VGPRs free while the recursion runs.
// Deferral of the rare expensive branch of the law (the Jacobi eigen-decomposition behind a truncating pinv): when the
// kernel allows it, the row program raises `deferred` instead of running the sweeps; the kernel then parks the row
// index in a worklist that a second, densely packed pass works off (abrk_kernels.h osc_kernel, modes 1 / 2).  Lanes
// diverge otherwise: one such row makes its whole wavefront run the sweeps.
// Worklist of deferred rows: kWlLists sub-lists (wavefront w appends to sub-list w mod kWlLists), each with its own
// counter on its own 64-byte line - one shared counter serialises at ~90 atomics per microsecond, which at 8 M rows
// (125 k wavefronts with a deferred row) cost 1.4 ms, three times the arithmetic.
// Layout (ints): counter of sub-list s at [16 s]; row indices of sub-list s at [16 kWlLists + s cap + k], cap = wl_capacity(B).
constexpr int kWlLists = 256;
constexpr int kWlBlock = 64;  // rows per first-pass workgroup (= abrk_kernels.h kBlock)
constexpr long wl_capacity(long B) { return ((B + kWlBlock - 1) / kWlBlock / kWlLists + 1) * kWlBlock; }  // rows per sub-list
constexpr long wl_ints(long B) { return 16L * kWlLists + kWlLists * wl_capacity(B); }
// Hand-over record of a deferred row of the six-row law (osc_law6 writes it, osc6_finish reads it): everything the
// truncating pseudo-inverse and the tail of the law need, so that the second pass does no kinematics -
//   S   [22]          Mx_inv = J M^-1 J^T, packed lower (21 values), a masked task row as an isolated zero
//   X   [6][xs(N)]    row r: the masked task Jacobian row J[r][0..N), then u_task[r], then (J v)[r] of the secondary
//                     controllers (0 without them); xs(N) = N + 2 rounded up to even (16-byte row starts)
//   b1  [N]           u0 - C dq: the training signal is b1 - J^T f
//   b2  [N]           what follows the training signal: gravity (+ M v of the secondary controllers)
constexpr int rec_xs(int n) { return (n + 3) & ~1; }
constexpr int rec_off_x() { return 22; }
constexpr int rec_off_b1(int n) { return 22 + 6 * rec_xs(n); }
constexpr int rec_len(int n) { return (22 + 6 * rec_xs(n) + 2 * n + 1) & ~1; }

struct ScratchBase {
  // true in the first pass of the six-row kernels (DeferOnly below): the row program is compiled WITHOUT the eigen-
  // decomposition - a row that needs it is always deferred - so its registers do not weigh on the two-wave budget
  static constexpr bool kDeferOnly = false;
  // true where the caller asked for no training signal (NoTs below): the six-row law then folds the gravity term into
  // its velocity term BEFORE the factorisations - six values fewer to carry through them, which is what the first
  // pass's 256-register budget was short of (it spilled them: 86 B per row of scratch writes, PMC)
  static constexpr bool kNoTs = false;
  // true where the launch's reference frame is known to be the end effector when the kernel is compiled (EeFrame below;
  // the first pass of the plain six-row law - the reference benchmark's setting - is instantiated that way too): the
  // forward kinematics then carries no frame capture (14 uniform compares, 24 captured registers live down the chain)
  // and ends in the same mulBE the capture would have run - the same bits
  static constexpr bool kEeFrame = false;
  bool allow_defer = false, deferred = false;
  bool singular = false;  // the row's M has a non-positive pivot (the law's verdict; the flag is stored after the outputs)
#if defined(ABRK_TIMELINE)
  unsigned long long tl[16] = {};
#endif
  ABRK_INL bool* defer_ptr() { return allow_defer ? &deferred : nullptr; }
  // where a deferring row parks itself (set by the kernel; the host check build passes plain arrays).
  //   hand-over mode (rec_base != nullptr; batches up to 262144 rows): the row leaves a record (record() below) - no
  //     list, no atomic; the kernel notes WHICH rows deferred as one 64-bit mask per 64-row chunk (osc_kernel: a ballot
  //     after the row program), and the finish kernel works each chunk's records off;
  //   recompute mode: the row's index joins sub-list wl_sub of the worklist `wl` (an atomic slot) and the second pass
  //     runs the complete row program on it.
  int* wl = nullptr;
  void* rec_base = nullptr;
  long wl_cap = 0, row = 0;
  int wl_sub = 0;
  bool handed_over = false;
  bool handover = false;  // hand-over mode (rec_base is valid)
  // -> true: hand-over mode, the row's record is record<T>(len); false: recompute mode, the row was appended to its
  // sub-list.  (A flag, not a null test of the pointer: `rec_base` is a generic pointer here, and the gfx950 backend of
  // ROCm 7.2 fails on the aperture test a generic null check can fold into - "V_CMP_NE_U32 0, $src_shared_base".)
  ABRK_INL bool claim() {
    if (handover) {
      handed_over = true;
      return true;
    }
    if (wl_cap == 0) return false;
#if defined(__HIP_DEVICE_COMPILE__)
    const int k = atomicAdd(wl + 16 * wl_sub, 1);
#else
    const int k = wl[16 * wl_sub]++;
#endif
    wl[16 * kWlLists + (long)wl_sub * wl_cap + k] = (int)row;
    return false;
  }
  // The records of a 64-row chunk are packed at the chunk's first slots, in lane order: the deferring lanes of a
  // wavefront are inside one branch together, so a ballot there numbers them (the same set the kernel's mask holds
  // afterwards), and the finish kernel's wavefront (chunk, s) can ask for record chunk * 64 + s at the same time as for
  // the chunk's mask - one memory round trip instead of mask -> row -> record.  The row's own index travels in the
  // record (its 22nd value).  (The host check build runs rows one by one: record = row.)
  template <class T>
  ABRK_INL T* record(int len) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long act = __ballot(1);
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
    return static_cast<T*>(rec_base) + (row - lane + rank) * len;
#else
    return static_cast<T*>(rec_base) + row * len;
#endif
  }
};
template <class T, int N>
struct RegScratch : ScratchBase {
  static constexpr bool kHasTab = false;  // no sin/cos table: the polynomial routine
  T f[N][3], t[N][3];
  // rows of the task Jacobian for the six-row OSC law (osc_law6): plain arrays here, the wavefront's LDS slab on the GPU
  T jrow[6][N];
  template <int R>
  ABRK_INL void put_row(ic<R>, const T (&row)[N]) {
    sfor<N>([&](auto i) ABRK_LAMBDA { jrow[R][i()] = row[i()]; });
  }
  template <int R>
  ABRK_INL void get_row(ic<R>, T (&row)[N]) const {
    sfor<N>([&](auto i) ABRK_LAMBDA { row[i()] = jrow[R][i()]; });
  }
  template <int K>
  ABRK_INL void put(ic<K>, const T (&fv)[3], const T (&tv)[3]) {
    sfor<3>([&](auto r) ABRK_LAMBDA {
      f[K][r()] = fv[r()];
      t[K][r()] = tv[r()];
    });
  }
  ABRK_INL void seal() const {}
  template <int K>
  ABRK_INL void get(ic<K>, T (&fv)[3], T (&tv)[3]) const {
    sfor<3>([&](auto r) ABRK_LAMBDA {
      fv[r()] = f[K][r()];
      tv[r()] = t[K][r()];
    });
  }
};

template <class Scr>
struct DeferOnly : Scr {
  static constexpr bool kDeferOnly = true;
};
template <class Scr>
struct NoTs : Scr {
  static constexpr bool kNoTs = true;
};
template <class Scr>
struct EeFrame : Scr {
  static constexpr bool kEeFrame = true;
};

// a + al x d + w (w . d) - w2 d : acceleration of a point at offset d from a point of the same body whose
// acceleration is a (w = angular velocity, w2 = |w|^2, al = angular acceleration).  w x (w x d) is taken as
// w (w . d) - |w|^2 d: 15 instructions for the whole expression instead of 24.
template <class T>
ABRK_INL void point_accel(const T (&a)[3], const T (&al)[3], const T (&w)[3], T w2, const T (&d)[3], T (&out)[3]) {
  const T s = dot3(w, d);
  T x0 = Rm<T>::fma(al[1], d[2], a[0]), x1 = Rm<T>::fma(al[2], d[0], a[1]), x2 = Rm<T>::fma(al[0], d[1], a[2]);
  x0 = Rm<T>::fma(-al[2], d[1], x0);
  x1 = Rm<T>::fma(-al[0], d[2], x1);
  x2 = Rm<T>::fma(-al[1], d[0], x2);
  x0 = Rm<T>::fma(w[0], s, x0);
  x1 = Rm<T>::fma(w[1], s, x1);
  x2 = Rm<T>::fma(w[2], s, x2);
  out[0] = Rm<T>::fma(-w2, d[0], x0);
  out[1] = Rm<T>::fma(-w2, d[1], x1);
  out[2] = Rm<T>::fma(-w2, d[2], x2);
}

// C(q,dq) dq of an orthogonal chain (base_config.py:678-727 applied to dq) alongside the dynamics pass - the same
// forward kinematics serves both.  Recursive Newton-Euler with the reference's world-frame inertia diagonals
// (base_config.py:628):
//   forward, joint i:  a_o moves from o_{i-1} to o_i with the old body's (w, al); then al += (w x z_i) dq_i,
//                      w += z_i dq_i;  link i+1: f = D_lin a(p), t = p x f + D_ang al + (D_ang w) x w
//   backward, joint k: c_k = z_k . (sum_{l>k} t_l - o_k x sum_{l>k} f_l)
// (the sign of the gyroscopic term is opposite to Euler's equation because D_ang is not rotated).
template <class T>
struct RneState {  // kinematic state of the body that carries the current link, referred to the last joint origin
  T w[3], al[3], ao[3], w2;
};
template <class T>
ABRK_INL void rne_init(RneState<T>& st) {
  sfor<3>([&](auto r) ABRK_LAMBDA { st.w[r()] = st.al[r()] = st.ao[r()] = T(0); });
  st.w2 = T(0);
}
// forward step for joint i = L - 1 and link L (COM at p): called from the link visitor of the forward kinematics,
// when jt.z[i] / jt.o[i] have just been recorded
template <int L, class A, class T, class Scr>
ABRK_INL void rne_forward_step(const A& arm, const Joints<A, T>& jt, const T (&p)[3], const T (&dq)[A::N],
                               RneState<T>& st, Scr& scr) {
  static_assert(A::kOrtho, "general affine chains take the per-link form (link_accumulate, CMODE_VEC)");
  constexpr int i = L - 1;
  if constexpr (i < A::N) {
    if constexpr (i > 0) {
      T d2[3] = {jt.o[i][0] - jt.o[i - 1][0], jt.o[i][1] - jt.o[i - 1][1], jt.o[i][2] - jt.o[i - 1][2]};
      T an[3];
      point_accel(st.ao, st.al, st.w, st.w2, d2, an);
      sfor<3>([&](auto r) ABRK_LAMBDA { st.ao[r()] = an[r()]; });
      T zd[3];
      cross3(st.w, jt.z[i], zd);  // d/dt z_i = w x z_i
      sfor<3>([&](auto r) ABRK_LAMBDA { st.al[r()] = Rm<T>::fma(zd[r()], dq[i], st.al[r()]); });
    }
    sfor<3>([&](auto r) ABRK_LAMBDA { st.w[r()] = Rm<T>::fma(jt.z[i][r()], dq[i], st.w[r()]); });
    st.w2 = dot3(st.w, st.w);
    T f[3] = {T(0), T(0), T(0)}, t[3] = {T(0), T(0), T(0)};
    if constexpr (L < A::NL) {
      if constexpr (link_has_linear_mass<A, L>()) {
        T dl[3] = {p[0] - jt.o[i][0], p[1] - jt.o[i][1], p[2] - jt.o[i][2]};
        T a[3];
        point_accel(st.ao, st.al, st.w, st.w2, dl, a);
        f[0] = AccMD<A, T, L, 0>::get(arm) * a[0];
        f[1] = AccMD<A, T, L, 1>::get(arm) * a[1];
        f[2] = AccMD<A, T, L, 2>::get(arm) * a[2];
        cross3(p, f, t);
      }
      if constexpr (A::MD(L, 3) != 0.0 || A::MD(L, 4) != 0.0 || A::MD(L, 5) != 0.0) {
        const T I0 = AccMD<A, T, L, 3>::get(arm), I1 = AccMD<A, T, L, 4>::get(arm), I2 = AccMD<A, T, L, 5>::get(arm);
        if constexpr (A::MD(L, 3) == A::MD(L, 4) && A::MD(L, 4) == A::MD(L, 5)) {
          // isotropic inertia: (D w) x w vanishes
          sfor<3>([&](auto r) ABRK_LAMBDA { t[r()] = Rm<T>::fma(I0, st.al[r()], t[r()]); });
        } else {
          const T Lw[3] = {I0 * st.w[0], I1 * st.w[1], I2 * st.w[2]};
          T n[3];
          cross3(Lw, st.w, n);
          t[0] += Rm<T>::fma(I0, st.al[0], n[0]);
          t[1] += Rm<T>::fma(I1, st.al[1], n[1]);
          t[2] += Rm<T>::fma(I2, st.al[2], n[2]);
        }
      }
    }
    scr.put(ic<i>{}, f, t);
  }
}
// backward sweep: c_k = z_k . (sum_{l>k} t_l - o_k x sum_{l>k} f_l)
template <class A, class T, class Scr>
ABRK_INL void rne_backward(const Joints<A, T>& jt, Scr& scr, T (&cv)[A::N]) {
  constexpr int N = A::N;
  T F[3] = {T(0), T(0), T(0)}, Nm[3] = {T(0), T(0), T(0)};
  scr.seal();
  sfor<N>([&](auto kr) ABRK_LAMBDA {
    constexpr int k = N - 1 - kr();
    T f[3], t[3];
    scr.get(ic<k>{}, f, t);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      F[r()] += f[r()];
      Nm[r()] += t[r()];
    });
    T of[3];
    cross3(jt.o[k], F, of);
    const T wv[3] = {Nm[0] - of[0], Nm[1] - of[1], Nm[2] - of[2]};
    cv[k] = dot3(jt.z[k], wv);
  });
}

// FK + M, g (+C).  Returns joint state for Jacobians; (XR, xo) = last rotated joint frame.
// `extra(ic<l>, p)` runs once per link l = 1..N (p = origin of its frame) while XR still holds the
// rotation of joint_{l-1} after its Rz - callers that need per-link frames hook in here.
template <class A, class T, int CM, class Cap, class Extra, class Sc = ScCompute>
ABRK_INL void kin_dyn_hook(const A& arm, const T (&q)[A::N], const T (&dq)[A::N], Joints<A, T>& jt,
                           Dyn<A, T, CM>& d, T (&XR)[9], T (&xo)[3], Cap& cap, Extra&& extra, const Sc& scp = Sc{}) {
  dyn_init(d);
  auto visit = [&](auto L, const T(&p)[3]) ABRK_LAMBDA {
    omega_advance<L()>(jt, dq, d);
    body_advance<L()>(jt, dq, d);
    link_accumulate<L()>(arm, jt, dq, p, d);
    angular_link_coriolis<L()>(arm, jt, d);
    extra(L, p);
  };
  if constexpr (std::is_same<Sc, ScTab>::value) {
    T sv[A::N][2];
    sincos_all_tab<A::N>(q, sv, scp.tab);
    fk_forward(arm, q, jt, XR, xo, cap, visit, ScUse<T, A::N>{sv});
  } else if constexpr (std::is_same<Sc, ScCompute>::value) {
    T sv[A::N][2];
    sincos_all<A::N>(q, sv);
    fk_forward(arm, q, jt, XR, xo, cap, visit, ScUse<T, A::N>{sv});
  } else {
    fk_forward(arm, q, jt, XR, xo, cap, visit, scp);
  }
  angular_finish(arm, jt, dq, d);
}
template <class A, class T, int CM, class Cap>
ABRK_INL void kin_dyn(const A& arm, const T (&q)[A::N], const T (&dq)[A::N], Joints<A, T>& jt, Dyn<A, T, CM>& d,
                      T (&XR)[9], T (&xo)[3], Cap& cap) {
  kin_dyn_hook(arm, q, dq, jt, d, XR, xo, cap, [](auto, const T(&)[3]) ABRK_LAMBDA {});
}

// Jacobian of point p attached after joints 0..m-1 (m runtime, uniform):
//   Jv[:,i] = W_i (p - o_i), Jw[:,i] = z_i  for i < m, else 0        (base_config.py:522-592)
template <class A, class T>
ABRK_INL void jacobian(const Joints<A, T>& jt, const T (&p)[3], int m, T (&Jv)[A::N][3], T (&Jw)[A::N][3]) {
  sfor<A::N>([&](auto i) ABRK_LAMBDA {
    T dlt[3] = {p[0] - jt.o[i()][0], p[1] - jt.o[i()][1], p[2] - jt.o[i()][2]};
    T e[3];
    wapply<i()>(jt, dlt, e);
    // m is uniform over the launch: a 0 / 1 factor from a scalar select (one v_mul per entry) instead of a vector
    // select per entry (two v_cndmask for a double); folds away where m is the compile-time N
    const T on = i() < m ? T(1) : T(0);
    sfor<3>([&](auto r) ABRK_LAMBDA {
      Jv[i()][r()] = on * e[r()];
      Jw[i()][r()] = on * jt.z[i()][r()];
    });
  });
}

// dJ/dt of the same point (base_config.py:470-520):
//   dJv[:,i] = W_i (sum_{k>=i} Jv_k dq_k) + Omega_i Jv_i ;  dJw[:,i] = Omega_i z_i     (i < m)
template <class A, class T>
ABRK_INL void jacobian_dot(const Joints<A, T>& jt, const T (&dq)[A::N], const T (&Jv)[A::N][3], int m,
                           T (&dJv)[A::N][3], T (&dJw)[A::N][3]) {
  constexpr int N = A::N;
  Dyn<A, T, CMODE_VEC> tmp;  // only .om is used
  if constexpr (A::kOrtho) {
    tmp.om[0][0] = tmp.om[0][1] = tmp.om[0][2] = T(0);
    sfor<N - 1>([&](auto j) ABRK_LAMBDA {
      sfor<3>([&](auto r) ABRK_LAMBDA { tmp.om[j() + 1][r()] = tmp.om[j()][r()] + dq[j()] * jt.z[j()][r()]; });
    });
  }
  T s[3] = {T(-0.0), T(-0.0), T(-0.0)};
  sfor<N>([&](auto jr) ABRK_LAMBDA {
    constexpr int j = N - 1 - jr();
    sfor<3>([&](auto r) ABRK_LAMBDA { s[r()] += Jv[j][r()] * dq[j]; });  // Jv is already 0 for j >= m
    T t1[3], t2[3], t3[3];
    wapply<j>(jt, s, t1);
    omega_apply<j>(jt, tmp, dq, Jv[j], t2);
    omega_apply<j>(jt, tmp, dq, jt.z[j], t3);
    const T on = j < m ? T(1) : T(0);  // uniform: see jacobian()
    sfor<3>([&](auto r) ABRK_LAMBDA {
      dJv[j][r()] = on * (t1[r()] + t2[r()]);
      dJw[j][r()] = on * t3[r()];
    });
  });
}

// number of joints a frame depends on == the reference's `end_point` (base_config.py:565-572)
ABRK_INL int frame_joints(int frame, int N) {
  if (frame == 2 * N + 1) return N;
  int e = frame >> 1;
  return e < N ? e : N;
}

}  // namespace abrk
