/*
 * abrk_oracle.c - CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * (as the checker / the reported baseline).  The product (libabrk.so, abr_control_amd/)
 * never links, imports or falls back to it.
 *
 * Parity status: PINNED.  Every function here is checked by tests/test_oracle_golden.py
 * against (a) outputs of the reference itself run in the build container
 * (tests/golden/<arm>.npz, produced by oracle/gen_golden.py), and (b) the reference's own
 * closed-form known answers (abr_control/arms/tests/dummy_base_arm.py via
 * tests/golden/known_answers.npz).
 *
 * Style: deliberately literal and slow - full 4x4 matrix chains, derivatives by the
 * product rule on the chain (exact for any affine static transforms), dense 6xn
 * Jacobians for every link, the Christoffel triple loop as written in the reference.
 * It shares NO arithmetic with the HIP kernels (abr_control_amd/csrc), which use
 * recursive/closed forms; only the plain data structs of include/abrk.h are shared.
 *
 * All file:line citations are into /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/abrk.h"

#define NJ ABRK_MAX_JOINTS

/* ------------------------------------------------------------------ 4x4 helpers */
static void m4_ident(double T[16]) {
  memset(T, 0, 16 * sizeof(double));
  T[0] = T[5] = T[10] = T[15] = 1.0;
}
static void m4_mul(const double A[16], const double B[16], double C[16]) {
  double R[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0.0;
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      R[i * 4 + j] = s;
    }
  memcpy(C, R, sizeof(R));
}
static void m4_from_affine(const double S[12], double T[16]) {
  memcpy(T, S, 12 * sizeof(double));
  T[12] = T[13] = T[14] = 0.0;
  T[15] = 1.0;
}
/* Rz(q) and its first / second derivative wrt q (the `Tj_i l_{i+1} a` matrices,
 * e.g. ur5/config.py:111-118).  order 0,1,2.                                       */
static void m4_rz(double q, int order, double T[16]) {
  double c = cos(q), s = sin(q);
  memset(T, 0, 16 * sizeof(double));
  if (order == 0) {
    T[0] = c; T[1] = -s; T[4] = s; T[5] = c; T[10] = 1.0; T[15] = 1.0;
  } else if (order == 1) {
    T[0] = -s; T[1] = -c; T[4] = c; T[5] = -s;
  } else {
    T[0] = -c; T[1] = s; T[4] = -s; T[5] = -c;
  }
}

/* T(frame) with the Rz of joints d1 and d2 replaced by their derivative (d = -1: none;
 * d1 == d2 >= 0: second derivative).  Follows the chain of `_calc_T`
 * (ur5/config.py:301-339; jaco2/config.py:316-356; twojoint/config.py:157-181).
 * Returns 0, or -1 for an invalid frame id.                                         */
static int chain(const abrk_arm_desc* a, int frame, const double* q, int d1, int d2, double T[16]) {
  int n = a->n_joints;
  double S[16], Rz[16];
  if (frame < 0 || frame > 2 * n + 1) return -1;
  m4_from_affine(a->A0, T); /* link0 */
  for (int i = 0; i < n; i++) {
    if (frame >= 2 * i + 1) { /* joint_i = link_i * Tl_i j_i */
      m4_from_affine(a->AJ[i], S);
      m4_mul(T, S, T);
    }
    if (frame >= 2 * i + 2) { /* link_{i+1} = joint_i * Rz(q_i) * Tj_i l_{i+1} b */
      int order = (d1 == i) + (d2 == i);
      m4_rz(q[i], order, Rz);
      m4_mul(T, Rz, T);
      m4_from_affine(a->B[i], S);
      m4_mul(T, S, T);
    } else {
      /* frame does not depend on q_i: any requested derivative wrt q_i is zero */
      if (d1 == i || d2 == i) {
        memset(T, 0, 16 * sizeof(double));
        return 0;
      }
    }
  }
  if (frame == 2 * n + 1 && a->has_ee) {
    m4_from_affine(a->E, S);
    m4_mul(T, S, T);
  }
  if ((d1 >= 0 || d2 >= 0)) T[15] = 0.0; /* derivative of the constant 1 */
  return 0;
}

static int end_point(const abrk_arm_desc* a, int frame) {
  /* base_config.py:565-572 */
  int n = a->n_joints;
  if (frame == 2 * n + 1) return n;
  int e = frame / 2; /* linkN -> N, jointN -> N */
  return e < n ? e : n;
}

/* ------------------------------------------------------------------ public: kinematics */
int abrk_oracle_T(const abrk_arm_desc* a, int frame, const double* q, double T[16]) {
  return chain(a, frame, q, -1, -1, T);
}

/* base_config.py:371-392 / 739-789 : Tx = (T * [x,y,z,1])[:3] */
int abrk_oracle_Tx(const abrk_arm_desc* a, int frame, const double* q, const double* x, double p[3]) {
  double T[16];
  double xh[4] = {x ? x[0] : 0.0, x ? x[1] : 0.0, x ? x[2] : 0.0, 1.0};
  if (chain(a, frame, q, -1, -1, T)) return -1;
  for (int r = 0; r < 3; r++) {
    double s = 0.0;
    for (int k = 0; k < 4; k++) s += T[r * 4 + k] * xh[k];
    p[r] = s;
  }
  return 0;
}

/* base_config.py:522-592 : rows 0-2 d(Tx)/dq_i, rows 3-5 J_orientation[i] = R(joint_i)*[0,0,1]
 * for i < end_point, else 0.  J is [6][n] row-major.  dk >= 0: return dJ/dq_dk instead. */
static int jac(const abrk_arm_desc* a, int frame, const double* q, const double* x, int dk, double* J) {
  int n = a->n_joints;
  double T[16];
  double xh[4] = {x ? x[0] : 0.0, x ? x[1] : 0.0, x ? x[2] : 0.0, 1.0};
  int ep = end_point(a, frame);
  if (frame < 0 || frame > 2 * n + 1) return -1;
  for (int i = 0; i < n; i++) {
    chain(a, frame, q, i, dk, T);
    for (int r = 0; r < 3; r++) {
      double s = 0.0;
      for (int k = 0; k < 4; k++) s += T[r * 4 + k] * xh[k];
      J[r * n + i] = s;
    }
    if (i < ep) {
      chain(a, 2 * i + 1, q, dk, -1, T); /* T(joint_i) or its derivative wrt q_dk */
      for (int r = 0; r < 3; r++) J[(3 + r) * n + i] = T[r * 4 + 2];
    } else {
      for (int r = 0; r < 3; r++) J[(3 + r) * n + i] = 0.0;
    }
  }
  return 0;
}

int abrk_oracle_J(const abrk_arm_desc* a, int frame, const double* q, const double* x, double* J) {
  return jac(a, frame, q, x, -1, J);
}

/* base_config.py:470-520 : dJ[i,j] = sum_k dJ[i,j]/dq_k * dq_k */
int abrk_oracle_dJ(const abrk_arm_desc* a, int frame, const double* q, const double* dq,
                   const double* x, double* dJ) {
  int n = a->n_joints;
  double D[6 * NJ];
  if (frame < 0 || frame > 2 * n + 1) return -1;
  for (int e = 0; e < 6 * n; e++) dJ[e] = 0.0;
  for (int k = 0; k < n; k++) {
    jac(a, frame, q, x, k, D);
    for (int e = 0; e < 6 * n; e++) dJ[e] += D[e] * dq[k];
  }
  return 0;
}

/* base_config.py:594-645 : M = sum_{l < N_LINKS} J_l^T M_l J_l, J_l = J("link l"), world frame.
 * dk >= 0: dM/dq_dk by the product rule.                                             */
static void inertia(const abrk_arm_desc* a, const double* q, int dk, double* M) {
  int n = a->n_joints;
  double J[6 * NJ], D[6 * NJ];
  for (int e = 0; e < n * n; e++) M[e] = 0.0;
  for (int l = 0; l < a->n_links_dyn; l++) {
    jac(a, 2 * l, q, NULL, -1, J);
    if (dk >= 0) jac(a, 2 * l, q, NULL, dk, D);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = 0.0;
        for (int r = 0; r < 6; r++) {
          double w = a->mdiag[l][r];
          if (dk < 0)
            s += J[r * n + i] * w * J[r * n + j];
          else
            s += D[r * n + i] * w * J[r * n + j] + J[r * n + i] * w * D[r * n + j];
        }
        M[i * n + j] += s;
      }
  }
}

int abrk_oracle_M(const abrk_arm_desc* a, const double* q, double* M) {
  inertia(a, q, -1, M);
  return 0;
}

/* base_config.py:417-468 : g = sum_l J_l^T M_l [0,0,-9.81,0,0,0]^T */
int abrk_oracle_g(const abrk_arm_desc* a, const double* q, double* g) {
  int n = a->n_joints;
  double J[6 * NJ];
  const double grav[6] = {0, 0, -9.81, 0, 0, 0};
  for (int i = 0; i < n; i++) g[i] = 0.0;
  for (int l = 0; l < a->n_links_dyn; l++) {
    jac(a, 2 * l, q, NULL, -1, J);
    for (int i = 0; i < n; i++) {
      double s = 0.0;
      for (int r = 0; r < 6; r++) s += J[r * n + i] * a->mdiag[l][r] * grav[r];
      g[i] += s;
    }
  }
  return 0;
}

/* base_config.py:678-727 : C[k,j] = sum_i 1/2 (dM_kj/dq_i + dM_ki/dq_j - dM_ij/dq_k) dq_i */
int abrk_oracle_C(const abrk_arm_desc* a, const double* q, const double* dq, double* C) {
  int n = a->n_joints;
  double dM[NJ][NJ * NJ];
  for (int k = 0; k < n; k++) inertia(a, q, k, dM[k]);
  for (int kk = 0; kk < n; kk++)
    for (int jj = 0; jj < n; jj++) {
      double s = 0.0;
      for (int ii = 0; ii < n; ii++) {
        double dMkjdqi = dM[ii][kk * n + jj];
        double dMkidqj = dM[jj][kk * n + ii];
        double dMijdqk = dM[kk][ii * n + jj];
        s += 0.5 * (dMkjdqi + dMkidqj - dMijdqk) * dq[ii];
      }
      C[kk * n + jj] = s;
    }
  return 0;
}

/* base_config.py:647-676 : R = T[:3,:3] */
int abrk_oracle_R(const abrk_arm_desc* a, int frame, const double* q, double R[9]) {
  double T[16];
  if (chain(a, frame, q, -1, -1, T)) return -1;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[r * 3 + c] = T[r * 4 + c];
  return 0;
}

/* base_config.py:791-837 : [R^T | -R^T t] (transpose, not a true inverse) */
int abrk_oracle_Tinv(const abrk_arm_desc* a, int frame, const double* q, double Ti[16]) {
  double T[16];
  if (chain(a, frame, q, -1, -1, T)) return -1;
  m4_ident(Ti);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) Ti[r * 4 + c] = T[c * 4 + r];
  for (int r = 0; r < 3; r++) {
    double s = 0.0;
    for (int c = 0; c < 3; c++) s += Ti[r * 4 + c] * T[c * 4 + 3];
    Ti[r * 4 + 3] = -s;
  }
  return 0;
}

/* ------------------------------------------------------------------ small dense linear algebra
 * Stand-ins for the LAPACK calls numpy makes in osc.py / sliding.py (inv, det, pinv, eigh).
 * N up to 7.                                                                             */
#define LN 8

/* Gauss-Jordan inverse with partial pivoting; returns det (0 => singular). */
static double la_inv(const double* A, int n, double* Ainv) {
  double W[LN][2 * LN];
  double det = 1.0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      W[i][j] = A[i * n + j];
      W[i][n + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < n; c++) {
    int p = c;
    for (int r = c + 1; r < n; r++)
      if (fabs(W[r][c]) > fabs(W[p][c])) p = r;
    if (W[p][c] == 0.0) return 0.0;
    if (p != c) {
      for (int j = 0; j < 2 * n; j++) {
        double t = W[c][j]; W[c][j] = W[p][j]; W[p][j] = t;
      }
      det = -det;
    }
    det *= W[c][c];
    double inv = 1.0 / W[c][c];
    for (int j = 0; j < 2 * n; j++) W[c][j] *= inv;
    for (int r = 0; r < n; r++)
      if (r != c) {
        double f = W[r][c];
        if (f != 0.0)
          for (int j = 0; j < 2 * n; j++) W[r][j] -= f * W[c][j];
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ainv[i * n + j] = W[i][n + j];
  return det;
}

/* One-sided (Hestenes) Jacobi SVD of A [m x n], m,n <= LN: A = U diag(s) V^T with
 * U [m x r], V [n x r], r = min(m,n) handled by working on the thinner orientation. */
static void la_svd_cols(double G[LN][LN], int rows, int cols, double V[LN][LN], double* s) {
  /* orthogonalise the `cols` columns of G (rows x cols); V accumulates rotations */
  for (int i = 0; i < cols; i++)
    for (int j = 0; j < cols; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0;
    for (int p = 0; p < cols - 1; p++)
      for (int qq = p + 1; qq < cols; qq++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < rows; r++) {
          alpha += G[r][p] * G[r][p];
          beta += G[r][qq] * G[r][qq];
          gamma += G[r][p] * G[r][qq];
        }
        if (gamma == 0.0) continue;
        double lim = sqrt(alpha * beta);
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 2.3e-16 * lim) continue;
        off = fmax(off, fabs(gamma) / lim);
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < rows; r++) {
          double gp = G[r][p], gq = G[r][qq];
          G[r][p] = c * gp - sn * gq;
          G[r][qq] = sn * gp + c * gq;
        }
        for (int r = 0; r < cols; r++) {
          double vp = V[r][p], vq = V[r][qq];
          V[r][p] = c * vp - sn * vq;
          V[r][qq] = sn * vp + c * vq;
        }
      }
    if (off == 0.0) break;
  }
  for (int j = 0; j < cols; j++) {
    double nn = 0;
    for (int r = 0; r < rows; r++) nn += G[r][j] * G[r][j];
    s[j] = sqrt(nn);
  }
}

/* numpy.linalg.pinv(A, rcond): A [m x n] -> P [n x m]; singular values <= rcond*max -> 0. */
static void la_pinv(const double* A, int m, int n, double rcond, double* P) {
  double G[LN][LN], V[LN][LN], s[LN];
  int transposed = (m < n); /* orthogonalise the fewer columns */
  int rows = transposed ? n : m, cols = transposed ? m : n;
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) G[r][c] = transposed ? A[c * n + r] : A[r * n + c];
  la_svd_cols(G, rows, cols, V, s);
  double smax = 0;
  for (int j = 0; j < cols; j++) smax = fmax(smax, s[j]);
  double cutoff = rcond * smax;
  /* X = G V' with G = U S: X^+ = V S^-1 U^T, U_j = G_j / s_j  ->  X^+ = sum_j V_j G_j^T / s_j^2 */
  double Xp[LN][LN]; /* cols x rows */
  for (int i = 0; i < cols; i++)
    for (int r = 0; r < rows; r++) {
      double acc = 0;
      for (int j = 0; j < cols; j++)
        if (s[j] > cutoff) acc += V[i][j] * G[r][j] / (s[j] * s[j]);
      Xp[i][r] = acc;
    }
  /* X = A (not transposed): P = Xp [n x m].  X = A^T: P = (A^T)^+^T = Xp^T            */
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++) P[i * m + j] = transposed ? Xp[j][i] : Xp[i][j];
}

/* symmetric eigen (cyclic two-sided Jacobi): A [n x n] -> w, V (columns) */
static void la_eigh(const double* A, int n, double* w, double* Vout) {
  double S[LN][LN], V[LN][LN];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      S[i][j] = (i >= j) ? A[i * n + j] : A[j * n + i]; /* lower triangle, like eigh(UPLO='L') */
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) off += S[p][q] * S[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        if (S[p][q] == 0.0) continue;
        double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          double skp = S[k][p], skq = S[k][q];
          S[k][p] = c * skp - s * skq;
          S[k][q] = s * skp + c * skq;
        }
        for (int k = 0; k < n; k++) {
          double spk = S[p][k], sqk = S[q][k];
          S[p][k] = c * spk - s * sqk;
          S[q][k] = s * spk + c * sqk;
        }
        for (int k = 0; k < n; k++) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) {
    w[i] = S[i][i];
    for (int j = 0; j < n; j++) Vout[i * n + j] = V[i][j];
  }
}

/* ------------------------------------------------------------------ utils/transformations.py */
/* unit_vector, transformations.py:1632-1676 (1-D case) */
static void tf_unit(double* v, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += v[i] * v[i];
  s = sqrt(s);
  for (int i = 0; i < n; i++) v[i] /= s;
}

/* quaternion_from_matrix(isprecise=False), transformations.py:1192-1271 */
void abrk_oracle_quat_from_matrix(const double R[9], double q[4]) {
  double m00 = R[0], m01 = R[1], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6],
         m21 = R[7], m22 = R[8];
  double K[16] = {m00 - m11 - m22, 0, 0, 0,
                  m01 + m10, m11 - m00 - m22, 0, 0,
                  m02 + m20, m12 + m21, m22 - m00 - m11, 0,
                  m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22};
  double w[4], V[16];
  for (int i = 0; i < 16; i++) K[i] /= 3.0;
  la_eigh(K, 4, w, V);
  int best = 0;
  for (int i = 1; i < 4; i++)
    if (w[i] > w[best]) best = i;
  q[0] = V[3 * 4 + best];
  q[1] = V[0 * 4 + best];
  q[2] = V[1 * 4 + best];
  q[3] = V[2 * 4 + best];
  if (q[0] < 0.0)
    for (int i = 0; i < 4; i++) q[i] = -q[i];
}

/* quaternion_from_euler(ai,aj,ak,'rxyz'), transformations.py:1096-1150.
 * 'rxyz' -> (firstaxis, parity, repetition, frame) = (2, 1, 0, 1):
 *   i = 3, j = _NEXT_AXIS[3] + 1 = 2, k = _NEXT_AXIS[2] + 1 = 1; swap ai,ak; aj = -aj.   */
void abrk_oracle_quat_from_euler_rxyz(double ai, double aj, double ak, double q[4]) {
  double t = ai; ai = ak; ak = t;
  aj = -aj;
  ai /= 2.0; aj /= 2.0; ak /= 2.0;
  double ci = cos(ai), si = sin(ai), cj = cos(aj), sj = sin(aj), ck = cos(ak), sk = sin(ak);
  double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  const int i = 3, j = 2, k = 1;
  q[0] = cj * cc + sj * ss;
  q[i] = cj * sc - sj * cs;
  q[j] = cj * ss + sj * cc;
  q[k] = cj * cs - sj * sc;
  q[j] *= -1.0;
}

/* euler_matrix(ai,aj,ak,'rxyz')[:3,:3], transformations.py:973-1032.
 * (firstaxis, parity, repetition, frame) = (2,1,0,1): i=2, j=_NEXT_AXIS[3]=1, k=_NEXT_AXIS[2]=0;
 * swap ai,ak; negate all three.                                                       */
void abrk_oracle_euler_matrix_rxyz(double ai, double aj, double ak, double M[9]) {
  double t = ai; ai = ak; ak = t;
  ai = -ai; aj = -aj; ak = -ak;
  double si = sin(ai), sj = sin(aj), sk = sin(ak), ci = cos(ai), cj = cos(aj), ck = cos(ak);
  double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  const int i = 2, j = 1, k = 0;
  M[i * 3 + i] = cj * ck;
  M[i * 3 + j] = sj * sc - cs;
  M[i * 3 + k] = sj * cc + ss;
  M[j * 3 + i] = cj * sk;
  M[j * 3 + j] = sj * ss + cc;
  M[j * 3 + k] = sj * cs - sc;
  M[k * 3 + i] = -sj;
  M[k * 3 + j] = cj * si;
  M[k * 3 + k] = cj * ci;
}

/* quaternion_multiply(q1, q0), transformations.py:1274-1290 */
void abrk_oracle_quat_mul(const double q1[4], const double q0[4], double r[4]) {
  double w0 = q0[0], x0 = q0[1], y0 = q0[2], z0 = q0[3];
  double w1 = q1[0], x1 = q1[1], y1 = q1[2], z1 = q1[3];
  r[0] = -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0;
  r[1] = x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0;
  r[2] = -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0;
  r[3] = x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0;
}

/* robot_config.quaternion, base_config.py:304-318 (on the fp64 rotation = Oracle-D) */
int abrk_oracle_quaternion(const abrk_arm_desc* a, int frame, const double* q, double quat[4]) {
  double R[9];
  if (abrk_oracle_R(a, frame, q, R)) return -1;
  abrk_oracle_quat_from_matrix(R, quat);
  tf_unit(quat, 4);
  return 0;
}

/* ------------------------------------------------------------------ secondary controllers */
static void matvec(const double* M, int n, const double* v, double* out) {
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < n; j++) s += M[i * n + j] * v[j];
    out[i] = s;
  }
}

static double pymod(double a, double b) { /* Python/numpy float % : result has sign of b */
  double r = fmod(a, b);
  if (r != 0.0 && ((r < 0) != (b < 0))) r += b;
  return r;
}

/* Damping.generate (damping.py:21-32); RestingConfig.generate (resting_config.py:18-42 ->
 * joint.py:104-131 with account_for_gravity=False, target_velocity=0).                 */
static void null_generate(const abrk_arm_desc* a, const abrk_null_ctrl* c, const double* q,
                          const double* dq, const double* M, double* u) {
  int n = a->n_joints;
  double v[NJ];
  if (c->kind == ABRK_NULL_DAMPING) {
    for (int i = 0; i < n; i++) v[i] = -c->kv * dq[i];
  } else {
    for (int i = 0; i < n; i++) {
      double qt = 0.0;
      if (c->rest_mask[i]) qt = pymod(c->rest_angles[i] - q[i] + M_PI, M_PI * 2) - M_PI;
      v[i] = c->kp * qt + c->kv * (0.0 - dq[i]);
    }
  }
  matvec(M, n, v, u);
}

/* Joint.generate, joint.py:104-131 (angle states; q_tilde_angle joint.py:42-46) */
int abrk_oracle_joint_generate(const abrk_arm_desc* a, const abrk_null_ctrl* c, int account_for_gravity,
                               const double* q, const double* dq, const double* target,
                               const double* target_velocity, double* u) {
  int n = a->n_joints;
  double M[NJ * NJ], v[NJ], g[NJ];
  abrk_oracle_M(a, q, M);
  if (c->kind == ABRK_NULL_DAMPING || c->kind == ABRK_NULL_RESTING) {
    null_generate(a, c, q, dq, M, u);
    return 0;
  }
  for (int i = 0; i < n; i++) {
    double qt = pymod(target[i] - q[i] + M_PI, M_PI * 2) - M_PI;
    double tv = target_velocity ? target_velocity[i] : 0.0;
    v[i] = c->kp * qt + c->kv * (tv - dq[i]);
  }
  matvec(M, n, v, u);
  if (account_for_gravity) {
    abrk_oracle_g(a, q, g);
    for (int i = 0; i < n; i++) u[i] -= g[i];
  }
  return 0;
}

/* ------------------------------------------------------------------ OSC.generate, osc.py:217-320 */
/* OSC._Mx, osc.py:120-147.  M [n,n]; J [k,n] (the task rows OSC keeps); Mx [k,k]; Minv [n,n] */
void abrk_oracle_osc_mx(int n, int k, const double* M, const double* J, double threshold, double* Mx, double* Minv) {
  double Mxinv[36], T1[6 * NJ];
  la_inv(M, n, Minv); /* osc.py:136 */
  for (int r = 0; r < k; r++) /* T1 = J Minv  [k x n] */
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int i = 0; i < n; i++) s += J[r * n + i] * Minv[i * n + j];
      T1[r * n + j] = s;
    }
  for (int r = 0; r < k; r++) /* osc.py:137 */
    for (int c = 0; c < k; c++) {
      double s = 0;
      for (int j = 0; j < n; j++) s += T1[r * n + j] * J[c * n + j];
      Mxinv[r * k + c] = s;
    }
  double det = la_inv(Mxinv, k, Mx);                                       /* osc.py:138-141 */
  if (!(fabs(det) >= threshold)) la_pinv(Mxinv, k, k, threshold * 0.1, Mx); /* osc.py:142-145 */
}

/* OSC._calc_orientation_forces, osc.py:149-196, from R_e = robot_config.R(ref_frame, q) (robot_config.quaternion is
 * quaternion_from_matrix of the same R, base_config.py:304-318).  Returns -2 for an unknown algorithm. */
int abrk_oracle_osc_orientation_forces(int algorithm, const double R_e[9], const double target_abg[3], double u_o[3]) {
  if (algorithm == 0) {
    double q_d[4], q_e[4], q_ec[4], q_r[4];
    abrk_oracle_quat_from_euler_rxyz(target_abg[0], target_abg[1], target_abg[2], q_d);
    tf_unit(q_d, 4);
    abrk_oracle_quat_from_matrix(R_e, q_e);
    tf_unit(q_e, 4);
    q_ec[0] = q_e[0]; q_ec[1] = -q_e[1]; q_ec[2] = -q_e[2]; q_ec[3] = -q_e[3];
    abrk_oracle_quat_mul(q_d, q_ec, q_r);
    double sg = (q_r[0] > 0) - (q_r[0] < 0); /* np.sign */
    for (int r = 0; r < 3; r++) u_o[r] = -q_r[1 + r] * sg;
  } else if (algorithm == 1) {
    double R_d[9], R_ed[9], q_ed[4];
    abrk_oracle_euler_matrix_rxyz(target_abg[0], target_abg[1], target_abg[2], R_d);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int m = 0; m < 3; m++) s += R_e[m * 3 + r] * R_d[m * 3 + c];
        R_ed[r * 3 + c] = s;
      }
    abrk_oracle_quat_from_matrix(R_ed, q_ed);
    tf_unit(q_ed, 4);
    for (int r = 0; r < 3; r++) {
      double s = 0;
      for (int c = 0; c < 3; c++) s += R_e[r * 3 + c] * q_ed[1 + c];
      u_o[r] = -1 * s;
    }
  } else {
    return -2;
  }
  return 0;
}

/* OSC._velocity_limiting, osc.py:198-215 with the constants of osc.py:89-115; in place on u_task[6] */
void abrk_oracle_osc_velocity_limiting(const abrk_osc_params* P, double u_task[6]) {
  double gains[6] = {P->kp, P->kp, P->kp, P->ko, P->ko, P->ko};
  double lamb[6];
  for (int r = 0; r < 6; r++) lamb[r] = gains[r] / P->kv;
  double sat_gain_xyz = P->vmax[0] / P->kp * P->kv, sat_gain_abg = P->vmax[1] / P->ko * P->kv;
  double scale_xyz = sat_gain_xyz, scale_abg = sat_gain_abg;
  double norm_xyz = sqrt(u_task[0] * u_task[0] + u_task[1] * u_task[1] + u_task[2] * u_task[2]);
  double norm_abg = sqrt(u_task[3] * u_task[3] + u_task[4] * u_task[4] + u_task[5] * u_task[5]);
  double scale[6] = {1, 1, 1, 1, 1, 1};
  if (norm_xyz > sat_gain_xyz)
    for (int r = 0; r < 3; r++) scale[r] *= scale_xyz / norm_xyz;
  if (norm_abg > sat_gain_abg)
    for (int r = 3; r < 6; r++) scale[r] *= scale_abg / norm_abg;
  for (int r = 0; r < 6; r++) u_task[r] = P->kv * scale[r] * lamb[r] * u_task[r];
}

int abrk_oracle_osc_generate(const abrk_arm_desc* a, const abrk_osc_params* P, const double* q,
                             const double* dq, const double* target, const double* target_velocity,
                             double* integrated_error, const double* u_null_ext, double* u,
                             double* training_signal) {
  int n = a->n_joints;
  double Jf[6 * NJ], J[6 * NJ], M[NJ * NJ], Minv[NJ * NJ];
  int idx[6], k = 0;
  const double zeros6[6] = {0, 0, 0, 0, 0, 0};
  const double* tv = target_velocity ? target_velocity : zeros6; /* osc.py:239-240 */
  for (int r = 0; r < 6; r++)
    if (P->ctrlr_dof[r]) idx[k++] = r;

  /* osc.py:242-247 */
  if (abrk_oracle_J(a, P->ref_frame, q, P->xyz_offset, Jf)) return -1;
  for (int r = 0; r < k; r++)
    for (int i = 0; i < n; i++) J[r * n + i] = Jf[idx[r] * n + i];
  abrk_oracle_M(a, q, M);

  /* _Mx, osc.py:120-147 */
  double Mx[36];
  abrk_oracle_osc_mx(n, k, M, J, 1e-3, Mx, Minv);

  /* desired task-space forces, osc.py:250-259 */
  double u_task[6] = {0, 0, 0, 0, 0, 0};
  if (P->ctrlr_dof[0] + P->ctrlr_dof[1] + P->ctrlr_dof[2] > 0) {
    double xyz[3];
    abrk_oracle_Tx(a, P->ref_frame, q, P->xyz_offset, xyz);
    for (int r = 0; r < 3; r++) u_task[r] = xyz[r] - target[r];
  }
  if (P->ctrlr_dof[3] + P->ctrlr_dof[4] + P->ctrlr_dof[5] > 0) {
    /* _calc_orientation_forces, osc.py:149-196 */
    double R_e[9];
    abrk_oracle_R(a, P->ref_frame, q, R_e);
    if (abrk_oracle_osc_orientation_forces(P->orientation_algorithm, R_e, target + 3, u_task + 3)) return -2;
  }

  /* integral term, osc.py:262-264 */
  if (P->ki != 0) {
    for (int r = 0; r < 6; r++) {
      integrated_error[r] += u_task[r];
      u_task[r] += P->ki * integrated_error[r];
    }
  }

  /* gains / velocity limiting, osc.py:266-272, 198-215, constants osc.py:89-115 */
  double gains[6] = {P->kp, P->kp, P->kp, P->ko, P->ko, P->ko};
  if (P->use_vmax) {
    abrk_oracle_osc_velocity_limiting(P, u_task);
  } else {
    for (int r = 0; r < 6; r++) u_task[r] *= gains[r];
  }

  /* velocity compensation, osc.py:274-282 */
  int tv_zero = 1;
  for (int r = 0; r < 6; r++)
    if (tv[r] != 0) tv_zero = 0;
  if (tv_zero) {
    double Mdq[NJ];
    matvec(M, n, dq, Mdq);
    for (int i = 0; i < n; i++) u[i] = -1 * P->kv * Mdq[i];
  } else {
    double dx[6] = {0, 0, 0, 0, 0, 0};
    for (int r = 0; r < k; r++) {
      double s = 0;
      for (int i = 0; i < n; i++) s += J[r * n + i] * dq[i];
      dx[idx[r]] = s;
    }
    for (int r = 0; r < 6; r++) u_task[r] += P->kv * (dx[r] - tv[r]);
    for (int i = 0; i < n; i++) u[i] = 0.0;
  }

  /* osc.py:285-288 : u -= J^T (Mx u_task[ctrlr_dof]) */
  double ut[6], f[6];
  for (int r = 0; r < k; r++) ut[r] = u_task[idx[r]];
  for (int r = 0; r < k; r++) {
    double s = 0;
    for (int c = 0; c < k; c++) s += Mx[r * k + c] * ut[c];
    f[r] = s;
  }
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int r = 0; r < k; r++) s += J[r * n + i] * f[r];
    u[i] -= s;
  }

  /* osc.py:291-292 */
  if (P->use_C) {
    double C[NJ * NJ], Cdq[NJ];
    abrk_oracle_C(a, q, dq, C);
    matvec(C, n, dq, Cdq);
    for (int i = 0; i < n; i++) u[i] -= Cdq[i];
  }
  if (training_signal)
    for (int i = 0; i < n; i++) training_signal[i] = u[i]; /* osc.py:297 */

  /* osc.py:300-301 */
  if (P->use_g) {
    double g[NJ];
    abrk_oracle_g(a, q, g);
    for (int i = 0; i < n; i++) u[i] -= g[i];
  }

  /* null space, osc.py:310-318 */
  int n_null = P->n_null + (u_null_ext ? 1 : 0);
  if (n_null > 0) {
    double JtMx[NJ * 6], Jbar[NJ * 6], filt[NJ * NJ];
    for (int i = 0; i < n; i++) /* J^T Mx  [n x k] */
      for (int c = 0; c < k; c++) {
        double s = 0;
        for (int r = 0; r < k; r++) s += J[r * n + i] * Mx[r * k + c];
        JtMx[i * k + c] = s;
      }
    for (int i = 0; i < n; i++) /* Jbar = Minv (J^T Mx) */
      for (int c = 0; c < k; c++) {
        double s = 0;
        for (int j = 0; j < n; j++) s += Minv[i * n + j] * JtMx[j * k + c];
        Jbar[i * k + c] = s;
      }
    for (int i = 0; i < n; i++) /* I - J^T Jbar^T */
      for (int j = 0; j < n; j++) {
        double s = 0;
        for (int r = 0; r < k; r++) s += J[r * n + i] * Jbar[j * k + r];
        filt[i * n + j] = (i == j ? 1.0 : 0.0) - s;
      }
    for (int c = 0; c < n_null; c++) {
      double un[NJ], pu[NJ];
      if (c < P->n_null)
        null_generate(a, &P->null_ctrl[c], q, dq, M, un);
      else
        for (int i = 0; i < n; i++) un[i] = u_null_ext[i];
      matvec(filt, n, un, pu);
      for (int i = 0; i < n; i++) u[i] += pu[i];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ Sliding.generate, sliding.py:34-99 */
int abrk_oracle_sliding_generate(const abrk_arm_desc* a, const abrk_sliding_params* P, const double* q,
                                 const double* dq, const double* target, const double* target_velocity,
                                 const double* target_acc, double* u, double* s_out) {
  int n = a->n_joints;
  double dq_ref[NJ], ddq_ref[NJ], s[NJ];
  if (P->cartesian) {
    double Jf[6 * NJ], dJf[6 * NJ], xyz[3], dxyz[3], Jinv[NJ * 3], v[3], w[3];
    if (abrk_oracle_J(a, P->ref_frame, q, P->offset, Jf)) return -1; /* [:3] = first 3 rows */
    abrk_oracle_Tx(a, P->ref_frame, q, P->offset, xyz);
    for (int r = 0; r < 3; r++) {
      double t = 0;
      for (int i = 0; i < n; i++) t += Jf[r * n + i] * dq[i];
      dxyz[r] = t;
    }
    la_pinv(Jf, 3, n, 1e-15, Jinv); /* np.linalg.pinv default rcond */
    abrk_oracle_dJ(a, P->ref_frame, q, dq, P->offset, dJf);
    for (int r = 0; r < 3; r++)
      v[r] = (target_velocity ? target_velocity[r] : 0.0) + P->lamb * (target[r] - xyz[r]);
    for (int i = 0; i < n; i++) {
      double t = 0;
      for (int r = 0; r < 3; r++) t += Jinv[i * 3 + r] * v[r];
      dq_ref[i] = t;
    }
    for (int r = 0; r < 3; r++) {
      double t = 0;
      for (int i = 0; i < n; i++) t += dJf[r * n + i] * dq_ref[i];
      w[r] = (target_acc ? target_acc[r] : 0.0) +
             P->lamb * ((target_velocity ? target_velocity[r] : 0.0) - dxyz[r]) - t;
    }
    for (int i = 0; i < n; i++) {
      double t = 0;
      for (int r = 0; r < 3; r++) t += Jinv[i * 3 + r] * w[r];
      ddq_ref[i] = t;
    }
  } else {
    for (int i = 0; i < n; i++) {
      double tvv = target_velocity ? target_velocity[i] : 0.0;
      double ta = target_acc ? target_acc[i] : 0.0;
      dq_ref[i] = tvv - P->lamb * (q[i] - target[i]);
      ddq_ref[i] = ta - P->lamb * (dq[i] - tvv);
    }
  }
  for (int i = 0; i < n; i++) s[i] = dq[i] - dq_ref[i];
  double M[NJ * NJ], C[NJ * NJ], g[NJ], a1[NJ], a2[NJ];
  abrk_oracle_M(a, q, M);
  abrk_oracle_C(a, q, dq, C);
  abrk_oracle_g(a, q, g);
  matvec(M, n, ddq_ref, a1);
  matvec(C, n, dq_ref, a2);
  for (int i = 0; i < n; i++) {
    u[i] = a1[i] + a2[i] + g[i] - P->kd * s[i];
    if (s_out) s_out[i] = s[i];
  }
  return 0;
}

/* ------------------------------------------------------------------ batch drivers (cpu_baseline) */
int abrk_oracle_osc_generate_batch(const abrk_arm_desc* a, const abrk_osc_params* P, int64_t B,
                                   const double* q, const double* dq, const double* target,
                                   const double* target_velocity, double* integrated_error,
                                   const double* u_null_ext, double* u, double* training_signal) {
  int n = a->n_joints;
  for (int64_t b = 0; b < B; b++) {
    int rc = abrk_oracle_osc_generate(
        a, P, q + b * n, dq + b * n, target + b * 6, target_velocity ? target_velocity + b * 6 : NULL,
        integrated_error ? integrated_error + b * 6 : NULL, u_null_ext ? u_null_ext + b * n : NULL,
        u + b * n, training_signal ? training_signal + b * n : NULL);
    if (rc) return rc;
  }
  return 0;
}

int abrk_oracle_sliding_generate_batch(const abrk_arm_desc* a, const abrk_sliding_params* P, int64_t B,
                                       const double* q, const double* dq, const double* target,
                                       const double* target_velocity, const double* target_acc,
                                       double* u, double* s) {
  int n = a->n_joints;
  int nt = P->cartesian ? 3 : n;
  for (int64_t b = 0; b < B; b++) {
    int rc = abrk_oracle_sliding_generate(
        a, P, q + b * n, dq + b * n, target + b * nt, target_velocity ? target_velocity + b * nt : NULL,
        target_acc ? target_acc + b * nt : NULL, u + b * n, s ? s + b * n : NULL);
    if (rc) return rc;
  }
  return 0;
}

/* ------------------------------------------------------------------ two-link plant + closed loop
 * ArmSim._step, arms/twojoint/arm_sim.py:101-137 (constants K1..K4 arm_sim.py:33-41), and the example
 * loop examples/PyGame/force_osc_xy.py:57-78: u = ctrlr.generate(q, dq, target); sim.send_forces(u). */
void abrk_oracle_twolink_step(const abrk_twolink_plant* K, double q[2], double dq[2], const double u[2]) {
  double C2 = cos(q[1]), S2 = sin(q[1]);
  double M11 = K->K1 + K->K2 * C2;
  double M12 = K->K3 + K->K4 * C2;
  double M21 = M12, M22 = K->K3;
  double H1 = -K->K2 * S2 * dq[0] * dq[1] - 1.0 / 2.0 * K->K2 * S2 * pow(dq[1], 2.0);
  double H2 = 1.0 / 2.0 * K->K2 * S2 * pow(dq[0], 2.0);
  double ddq1 = (H2 * M11 - H1 * M21 - M11 * u[1] + M21 * u[0]) / (pow(M12, 2.0) - M11 * M22);
  double ddq0 = (-H2 + u[1] - M22 * ddq1) / M21;
  dq[0] += ddq0 * K->dt;
  dq[1] += ddq1 * K->dt;
  q[0] += dq[0] * K->dt;
  q[1] += dq[1] * K->dt;
}

int abrk_oracle_rollout_twolink(const abrk_arm_desc* a, const abrk_osc_params* P, const abrk_twolink_plant* K,
                                int64_t B, int n_steps, int every, double* q, double* dq, const double* target,
                                double* integrated_error, double* q_traj, double* dq_traj, double* u_traj) {
  if (a->n_joints != 2) return -1;
  int n_chk = every > 0 ? n_steps / every : 0;
  for (int64_t b = 0; b < B; b++) {
    double u[2];
    for (int t = 0; t < n_steps; t++) {
      int rc = abrk_oracle_osc_generate(a, P, q + b * 2, dq + b * 2, target + b * 6, NULL,
                                        integrated_error ? integrated_error + b * 6 : NULL, NULL, u, NULL);
      if (rc) return rc;
      abrk_oracle_twolink_step(K, q + b * 2, dq + b * 2, u);
      if (every > 0 && (t + 1) % every == 0 && (t + 1) / every <= n_chk) {
        int64_t o = (b * n_chk + (t + 1) / every - 1) * 2;
        if (q_traj) { q_traj[o] = q[b * 2]; q_traj[o + 1] = q[b * 2 + 1]; }
        if (dq_traj) { dq_traj[o] = dq[b * 2]; dq_traj[o + 1] = dq[b * 2 + 1]; }
        if (u_traj) { u_traj[o] = u[0]; u_traj[o + 1] = u[1]; }
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ InverseKinematics.generate_path
 * controllers/path_planners/inverse_kinematics.py:28-135 */
void abrk_oracle_quat_from_euler_sxyz(double ai, double aj, double ak, double q[4]) {
  /* transformations.py:1096-1150 with axes 'sxyz' = (0,0,0,0): i=1, j=2, k=3, no swap, no negation */
  ai /= 2.0; aj /= 2.0; ak /= 2.0;
  double ci = cos(ai), si = sin(ai), cj = cos(aj), sj = sin(aj), ck = cos(ak), sk = sin(ak);
  double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
  q[0] = cj * cc + sj * ss;
  q[1] = cj * sc - sj * cs;
  q[2] = cj * ss + sj * cc;
  q[3] = cj * cs - sj * sc;
}

int abrk_oracle_ik_generate_path(const abrk_arm_desc* a, const abrk_ik_params* P, const double* position,
                                 const double* target, double* position_path, double* velocity_path) {
  int n = a->n_joints, EE = 2 * n + 1;
  double max_dq = P->max_dq * P->dt, max_dx = P->max_dx * P->dt, max_dr = P->max_dr * P->dt;
  double Qd[4], q[NJ];
  abrk_oracle_quat_from_euler_sxyz(target[3], target[4], target[5], Qd);
  tf_unit(Qd, 4);
  for (int i = 0; i < n; i++) q[i] = position[i];
  for (int ii = 0; ii < P->n_timesteps; ii++) {
    double J[6 * NJ], Tx[3], Qe[4], dx[3], dr[3], dq[NJ];
    abrk_oracle_J(a, EE, q, NULL, J);
    abrk_oracle_Tx(a, EE, q, NULL, Tx);
    for (int r = 0; r < 3; r++) dx[r] = target[r] - Tx[r];
    abrk_oracle_quaternion(a, EE, q, Qe);
    dr[0] = Qe[0] * Qd[1] - Qd[0] * Qe[1] - (Qd[2] * Qe[3] - Qd[3] * Qe[2]);
    dr[1] = Qe[0] * Qd[2] - Qd[0] * Qe[2] - (Qd[3] * Qe[1] - Qd[1] * Qe[3]);
    dr[2] = Qe[0] * Qd[3] - Qd[0] * Qe[3] - (Qd[1] * Qe[2] - Qd[2] * Qe[1]);
    double ndx = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
    double ndr = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
    if (ndx > max_dx) for (int r = 0; r < 3; r++) dx[r] = dx[r] / ndx * max_dx;
    if (ndr > max_dr) for (int r = 0; r < 3; r++) dr[r] = dr[r] / ndr * max_dr;
    if (P->method == 1) {
      double pJ[NJ * 6];
      la_pinv(J, 6, n, 1e-15, pJ);
      for (int i = 0; i < n; i++) {
        double s = 0;
        for (int r = 0; r < 3; r++) s += pJ[i * 6 + r] * dx[r] + pJ[i * 6 + 3 + r] * dr[r];
        dq[i] = s;
      }
    } else if (P->method == 2) {
      double A[36], Ai[36], b[6] = {dx[0], dx[1], dx[2], dr[0] * 0.3, dr[1] * 0.3, dr[2] * 0.3}, x[6];
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
          double s = (r == c) ? 0.001 : 0.0;
          for (int i = 0; i < n; i++) s += J[r * n + i] * J[c * n + i];
          A[r * 6 + c] = s;
        }
      la_inv(A, 6, Ai);
      for (int r = 0; r < 6; r++) {
        double s = 0;
        for (int c = 0; c < 6; c++) s += Ai[r * 6 + c] * b[c];
        x[r] = s;
      }
      for (int i = 0; i < n; i++) {
        double s = 0;
        for (int r = 0; r < 6; r++) s += J[r * n + i] * x[r];
        dq[i] = s;
      }
    } else {
      double pJx[NJ * 3], pJw[NJ * 3], b[NJ], jb[3];
      la_pinv(J, 3, n, 1e-15, pJx);
      la_pinv(J + 3 * n, 3, n, 1e-15, pJw);
      for (int i = 0; i < n; i++) b[i] = pJw[i * 3] * dr[0] + pJw[i * 3 + 1] * dr[1] + pJw[i * 3 + 2] * dr[2];
      for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += J[r * n + i] * b[i];
        jb[r] = s;
      }
      for (int i = 0; i < n; i++) {
        double a1 = pJx[i * 3] * dx[0] + pJx[i * 3 + 1] * dx[1] + pJx[i * 3 + 2] * dx[2];
        double a2 = pJx[i * 3] * jb[0] + pJx[i * 3 + 1] * jb[1] + pJx[i * 3 + 2] * jb[2];
        dq[i] = a1 + (b[i] - a2);
      }
    }
    double m = 0;
    for (int i = 0; i < n; i++) m = fmax(m, fabs(dq[i]));
    if (m > max_dq) for (int i = 0; i < n; i++) dq[i] = dq[i] / m * max_dq;
    for (int i = 0; i < n; i++) {
      position_path[ii * n + i] = q[i];
      velocity_path[ii * n + i] = dq[i];
      q[i] += dq[i];
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ AvoidJointLimits.generate
 * avoid_joint_limits.py:83-142, statement for statement (element i of every vector op).  The NaN
 * comparisons of "no limit" entries are all false in numpy, and those entries are zeroed at
 * :136,:139 - restated here with the flags so that no NaN arithmetic is needed.               */
int abrk_oracle_avoid_joint_limits_generate(int n, const abrk_limits_params* P, const double* q_in, double* u) {
  for (int i = 0; i < n; i++) {
    double q = q_in[i] - 1.0 * M_PI; /* :91 */
    double mn = P->min_joint_angles[i], mx = P->max_joint_angles[i];
    int nomin = P->no_limits_min[i], nomax = P->no_limits_max[i];
    int closer_to_min = (!nomin && !nomax) && (fabs(q - mn) >= fabs(q - mx)); /* :94-96 */
    int closer_to_max = (!nomin && !nomax) && (fabs(q - mn) <= fabs(q - mx)); /* :97-99 */
    double avoid_min = 0.0, avoid_max = 0.0;
    if (P->gradient[i]) { /* :108-115 */
      if (!nomin) avoid_min = fmin(exp(1.0 / (q - mn)), P->max_torque[i]);
      if (!nomax) avoid_max = -fmin(exp(-1.0 / (q - mx)), P->max_torque[i]);
    }
    int min_index = !nomin && (q - mn) < 0; /* :118 */
    int max_index = !nomax && (q - mx) > 0; /* :119 */
    if (P->cross_zero[i]) {                 /* :124-134 */
      int mi = min_index * (!nomax && (q - mx) > 0) * closer_to_max;
      int xi = max_index * (!nomin && (q - mn) < 0) * closer_to_min;
      min_index = mi;
      max_index = xi;
    }
    if (min_index) avoid_min = P->max_torque[i]; /* :136 */
    if (nomin) avoid_min = 0.0;                  /* :137 */
    if (max_index) avoid_max = -P->max_torque[i];
    if (nomax) avoid_max = 0.0;
    u[i] = avoid_min + avoid_max; /* :142 */
  }
  return 0;
}

/* ------------------------------------------------------------------ Floating.generate, floating.py:27-71
 * diag (may be NULL): [det(Mx_inv), s_min/s_max of Mx_inv] in task space - lets tests treat states at
 * the det / rcond thresholds (floating.py:50-56) separately.                                       */
int abrk_oracle_floating_generate(const abrk_arm_desc* a, int dynamic, int task_space, const double* q,
                                  const double* dq, double* u, double* diag) {
  int n = a->n_joints;
  double g[NJ], M[NJ * NJ], Jf[6 * NJ];
  int haveM = 0;
  abrk_oracle_g(a, q, g); /* :38 */
  if (task_space) {
    double Minv[NJ * NJ], T1[3 * NJ], Mxinv[9], Mx[9], Jbar[NJ * 3], tmp[NJ * 3], u_task[3];
    abrk_oracle_J(a, 2 * n + 1, q, NULL, Jf); /* J("EE")[:3], :42 */
    abrk_oracle_M(a, q, M);
    haveM = 1;
    la_inv(M, n, Minv); /* :47 */
    for (int r = 0; r < 3; r++)
      for (int j = 0; j < n; j++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += Jf[r * n + i] * Minv[i * n + j];
        T1[r * n + j] = s;
      }
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int j = 0; j < n; j++) s += T1[r * n + j] * Jf[c * n + j];
        Mxinv[r * 3 + c] = s; /* :49 */
      }
    double det = la_inv(Mxinv, 3, Mx);
    if (!(fabs(det) > 1e-3)) la_pinv(Mxinv, 3, 3, 1e-4, Mx); /* :50-56 */
    if (diag) {
      double w[3], V[9], lo = 1e300, hi = 0;
      la_eigh(Mxinv, 3, w, V);
      for (int r = 0; r < 3; r++) {
        lo = fmin(lo, fabs(w[r]));
        hi = fmax(hi, fabs(w[r]));
      }
      diag[0] = det;
      diag[1] = hi > 0 ? lo / hi : 0.0;
    }
    /* Jbar = Minv J^T Mx  [n x 3], :59 */
    for (int i = 0; i < n; i++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int r = 0; r < 3; r++) s += Jf[r * n + i] * Mx[r * 3 + c];
        tmp[i * 3 + c] = s;
      }
    for (int i = 0; i < n; i++)
      for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int j = 0; j < n; j++) s += Minv[i * n + j] * tmp[j * 3 + c];
        Jbar[i * 3 + c] = s;
      }
    for (int c = 0; c < 3; c++) { /* u_task = -Jbar^T g, :60 */
      double s = 0;
      for (int i = 0; i < n; i++) s += Jbar[i * 3 + c] * g[i];
      u_task[c] = -1 * s;
    }
    for (int i = 0; i < n; i++) { /* u = J^T u_task, :61 */
      double s = 0;
      for (int r = 0; r < 3; r++) s += Jf[r * n + i] * u_task[r];
      u[i] = s;
    }
  } else {
    for (int i = 0; i < n; i++) u[i] = -g[i]; /* :64 */
    if (diag) diag[0] = diag[1] = 1.0;
  }
  if (dynamic) { /* :67-69 */
    double Mdq[NJ];
    if (!haveM) abrk_oracle_M(a, q, M);
    matvec(M, n, dq, Mdq);
    for (int i = 0; i < n; i++) u[i] -= Mdq[i];
  }
  return 0;
}

/* ------------------------------------------------------------------ AvoidObstacles.generate
 * avoid_obstacles.py:38-120.  diag (may be NULL): the smallest |s_i/s_max - 0.01| over every pinv
 * taken (:112; 1.0 if none) - the distance of the state from the pinv truncation threshold; diag[1]: the
 * smallest s_max / (|segment|^2 trace(M^-1)) over them (1.0 if none) - at rounding-noise level the reference
 * inverts noise (the closest point sits on the axes it depends on) and its output is arbitrary.            */
int abrk_oracle_avoid_obstacles_generate(const abrk_arm_desc* a, const abrk_obstacles_params* P, const double* q,
                                         double* u, double* diag) {
  int n = a->n_joints;
  double u_psp[NJ], M[NJ * NJ], Minv[NJ * NJ];
  double margin = 1.0, mobility = 1.0, trMinv = 0.0;
  for (int i = 0; i < n; i++) u_psp[i] = 0.0;
  abrk_oracle_M(a, q, M); /* :54 */
  la_inv(M, n, Minv);
  for (int i = 0; i < n; i++) trMinv += Minv[i * n + i];
  for (int ob = 0; ob < P->n_obstacles; ob++) {
    const double* obstacle = P->obstacles[ob];
    const double* v = obstacle; /* :59 */
    for (int ii = 0; ii < n; ii++) {
      double p1[3], p2[3], vec_line[3], vec_ob_line[3], closest[3];
      abrk_oracle_Tx(a, 2 * ii + 1, q, NULL, p1); /* Tx("joint{ii}"), :64 */
      if (ii == n - 1) abrk_oracle_Tx(a, 2 * n + 1, q, NULL, p2);
      else abrk_oracle_Tx(a, 2 * (ii + 1) + 1, q, NULL, p2);
      double dot = 0, len2 = 0;
      for (int r = 0; r < 3; r++) {
        vec_line[r] = p2[r] - p1[r];    /* :72 */
        vec_ob_line[r] = v[r] - p1[r];  /* :74 */
        dot += vec_ob_line[r] * vec_line[r];
        len2 += vec_line[r] * vec_line[r];
      }
      /* :76; a zero-length segment gives 0/0 = NaN in numpy, every comparison below is then false
       * (Python's max(nan, x) keeps the nan) and the segment contributes nothing */
      if (len2 == 0.0) continue;
      double projection = dot / len2;
      for (int r = 0; r < 3; r++) {
        if (projection < 0) closest[r] = p1[r];      /* :77-79 */
        else if (projection > 1) closest[r] = p2[r]; /* :80-82 */
        else closest[r] = p1[r] + projection * vec_line[r];
      }
      double d2 = 0;
      for (int r = 0; r < 3; r++) d2 += (v[r] - closest[r]) * (v[r] - closest[r]);
      double dist = sqrt(d2);                                    /* :86 */
      double lo = P->threshold / 50, rho = dist - obstacle[3];  /* :90, Python max(a, b): b if b > a */
      if (lo > rho) rho = lo;
      if (rho < P->threshold) {
        double eta = 0.02, Fpsp[3];
        for (int r = 0; r < 3; r++) {
          double drhodx = (v[r] - closest[r]) / rho;
          Fpsp[r] = eta * (1.0 / rho - 1.0 / P->threshold) * 1.0 / pow(rho, 1.5) * drhodx; /* :96-102 */
        }
        /* :107-110: offset of the closest point in link ii+1's frame, Jacobian of that point */
        double Ti[16], m[3], Jf[6 * NJ];
        abrk_oracle_Tinv(a, 2 * (ii + 1), q, Ti);
        for (int r = 0; r < 3; r++)
          m[r] = Ti[r * 4] * closest[0] + Ti[r * 4 + 1] * closest[1] + Ti[r * 4 + 2] * closest[2] + Ti[r * 4 + 3];
        abrk_oracle_J(a, 2 * (ii + 1), q, m, Jf);
        /* :114-117 */
        double T1[3 * NJ], Mxinv[9], Mxpsp[9], f[3];
        for (int r = 0; r < 3; r++)
          for (int j = 0; j < n; j++) {
            double s = 0;
            for (int i = 0; i < n; i++) s += Jf[r * n + i] * Minv[i * n + j];
            T1[r * n + j] = s;
          }
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int j = 0; j < n; j++) s += T1[r * n + j] * Jf[c * n + j];
            Mxinv[r * 3 + c] = s;
          }
        la_pinv(Mxinv, 3, 3, 0.01, Mxpsp);
        if (diag) {
          double w[3], V[9], hi = 0;
          la_eigh(Mxinv, 3, w, V);
          for (int r = 0; r < 3; r++) hi = fmax(hi, fabs(w[r]));
          for (int r = 0; r < 3; r++)
            if (hi > 0) margin = fmin(margin, fabs(fabs(w[r]) / hi - 0.01));
          if (hi > 0) mobility = fmin(mobility, hi / (len2 * trMinv)); /* exactly 0: pinv(0) = 0, well defined */
        }
        for (int r = 0; r < 3; r++) f[r] = Mxpsp[r * 3] * Fpsp[0] + Mxpsp[r * 3 + 1] * Fpsp[1] + Mxpsp[r * 3 + 2] * Fpsp[2];
        for (int i = 0; i < n; i++) {
          double s = 0;
          for (int r = 0; r < 3; r++) s += Jf[r * n + i] * f[r];
          u_psp[i] += -1 * s; /* :119 */
        }
      }
    }
  }
  for (int i = 0; i < n; i++) { /* np.clip, :121 */
    double x = u_psp[i] * P->gain;
    u[i] = x < -P->maximum ? -P->maximum : (x > P->maximum ? P->maximum : x);
  }
  if (diag) {
    diag[0] = margin;
    diag[1] = mobility;
  }
  return 0;
}
