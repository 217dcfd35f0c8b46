// abrk_params.h - host-side conversion of the C ABI parameter structs (include/abrk.h) into the
// typed blocks the row programs consume.  Shared by abrk_host.cpp and tests/hostsim.
#pragma once
#include <cstring>

#include "../../include/abrk.h"
#include "abrk_ctrl.h"

namespace abrk {

inline int frame_m(int frame, int n) {
  if (frame == 2 * n + 1) return n;
  int e = frame / 2;
  return e < n ? e : n;
}

template <class T>
inline void fill_null(NullP<T>& d, const abrk_null_ctrl& s) {
  d.kind = s.kind;
  d.kp = T(s.kp);
  d.kv = T(s.kv);
  for (int i = 0; i < 7; i++) {
    d.mask[i] = s.rest_mask[i];
    d.rest[i] = T(s.rest_angles[i]);
  }
}

template <class T>
inline OscP<T> make_oscp(const abrk_osc_params& s, int n) {
  OscP<T> p;
  memset(&p, 0, sizeof p);
  p.kp = T(s.kp);
  p.ko = T(s.ko);
  p.kv = T(s.kv);
  p.ki = T(s.ki);
  p.vmax0 = T(s.vmax[0]);
  p.vmax1 = T(s.vmax[1]);
  if (s.use_vmax) {  // the kernels' own expressions, in their arithmetic type
    p.sat_xyz = p.vmax0 / p.kp * p.kv;
    p.sat_abg = p.vmax1 / p.ko * p.kv;
    p.lamb_xyz = p.kp / p.kv;
    p.lamb_abg = p.ko / p.kv;
  }
  p.use_vmax = s.use_vmax;
  p.use_g = s.use_g;
  p.alg = s.orientation_algorithm;
  for (int r = 0; r < 6; r++) p.dof[r] = s.ctrlr_dof[r] ? 1 : 0;
  p.pos_on = (p.dof[0] + p.dof[1] + p.dof[2]) > 0;
  p.ori_on = (p.dof[3] + p.dof[4] + p.dof[5]) > 0;
  p.ref_frame = s.ref_frame;
  p.m_joints = frame_m(s.ref_frame, n);
  p.has_off = (s.xyz_offset[0] != 0 || s.xyz_offset[1] != 0 || s.xyz_offset[2] != 0);
  for (int r = 0; r < 3; r++) p.off[r] = T(s.xyz_offset[r]);
  p.n_null = s.n_null;
  for (int c = 0; c < s.n_null; c++) fill_null(p.nul[c], s.null_ctrl[c]);
  return p;
}

template <class T>
inline SlidingP<T> make_slidingp(const abrk_sliding_params& s, int n) {
  SlidingP<T> p;
  memset(&p, 0, sizeof p);
  p.kd = T(s.kd);
  p.lamb = T(s.lamb);
  for (int r = 0; r < 3; r++) p.off[r] = T(s.offset[r]);
  p.cartesian = s.cartesian ? 1 : 0;
  p.ref_frame = s.ref_frame;
  p.m_joints = frame_m(s.ref_frame, n);
  p.has_off = (s.offset[0] != 0 || s.offset[1] != 0 || s.offset[2] != 0);
  return p;
}

template <class T>
inline JointP<T> make_jointp(const abrk_null_ctrl& c, int account_for_gravity) {
  JointP<T> p;
  memset(&p, 0, sizeof p);
  fill_null(p.c, c);
  p.account_for_gravity = account_for_gravity;
  return p;
}

template <class T>
inline LimitsP<T> make_limitsp(const abrk_limits_params& s) {
  LimitsP<T> p;
  memset(&p, 0, sizeof p);
  for (int i = 0; i < 7; i++) {
    p.mn[i] = T(s.min_joint_angles[i]);
    p.mx[i] = T(s.max_joint_angles[i]);
    p.mt[i] = T(s.max_torque[i]);
    p.cz[i] = s.cross_zero[i] ? 1 : 0;
    p.gr[i] = s.gradient[i] ? 1 : 0;
    p.nomin[i] = s.no_limits_min[i] ? 1 : 0;
    p.nomax[i] = s.no_limits_max[i] ? 1 : 0;
  }
  return p;
}

template <class T>
inline ObsP<T> make_obsp(const abrk_obstacles_params& s) {
  ObsP<T> p;
  memset(&p, 0, sizeof p);
  p.n = s.n_obstacles;
  p.threshold = T(s.threshold);
  p.gain = T(s.gain);
  p.maximum = T(s.maximum);
  for (int i = 0; i < s.n_obstacles && i < 16; i++)
    for (int r = 0; r < 4; r++) p.obs[i][r] = T(s.obstacles[i][r]);
  return p;
}

// The FAST OSC kernels apply when the task rows are exactly the first k position rows of the end-effector:
// k = 3 (x,y,z) for every arm, k = 2 (x,y - the planar examples, examples/PyGame/force_osc_xy.py:36-41) for
// arms of up to three joints.  Returns k, or 0 for the general (masked six-row) kernel.
inline int osc_fast_rows(const abrk_osc_params& P, int n, bool has_ext) {
  int k = 0;
  for (int r = 0; r < 6; r++) k += P.ctrlr_dof[r] ? 1 : 0;
  if (P.ref_frame != 2 * n + 1 || has_ext) return 0;
  if (k == 3 && P.ctrlr_dof[0] && P.ctrlr_dof[1] && P.ctrlr_dof[2]) return 3;
  if (k == 2 && P.ctrlr_dof[0] && P.ctrlr_dof[1] && n <= 3) return 2;
  return 0;
}

}  // namespace abrk
