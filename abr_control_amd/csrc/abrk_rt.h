// abrk_rt.h - user (runtime-table) arms: table construction on the host and the per-N
// kernel instantiations.  Each abrk_arm_rt<N>.hip defines ABRK_RT_N and includes this.
#pragma once
#include "../../include/abrk_types.h"
#include "abrk_kernels.h"

namespace abrk {

// same derivations as StaticArm, done at abrk_arm_create time
template <int N, class T>
void rt_fill(const abrk_arm_desc* d, RtArm<N, T>* t) {
  t->NL = d->n_links_dyn;
  const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int e = 0; e < 12; e++) {
    t->J0v[e] = T(aff_mul(d->A0, d->AJ[0], e));
    t->A0v[e] = T(d->A0[e]);
    t->BEv[e] = T(d->has_ee ? aff_mul(d->B[N - 1], d->E, e) : d->B[N - 1][e]);
  }
  for (int i = 0; i < N; i++)
    for (int e = 0; e < 12; e++) {
      t->Sv[i][e] = T(i + 1 < N ? aff_mul(d->B[i], d->AJ[i + 1], e) : ident[e]);
      t->Bv[i][e] = T(d->B[i][e]);
    }
  for (int l = 0; l <= N; l++)
    for (int r = 0; r < 6; r++) t->MDv[l][r] = T(d->mdiag[l][r]);
  for (int m = 0; m <= N; m++)
    for (int r = 0; r < 3; r++) {
      double s = 0.0;
      for (int l = m + 1; l < d->n_links_dyn && l <= N; l++) s += d->mdiag[l][3 + r];
      t->Isufv[m][r] = T(s);
    }
}

template <int N>
struct RtOps {
  static const ArmOps* ops() { return OpsFor<RtArm<N, double>, RtArm<N, float>>::ops(); }
  static size_t size(int dtype) { return dtype == 0 ? sizeof(RtArm<N, double>) : sizeof(RtArm<N, float>); }
  static void fill(int dtype, const abrk_arm_desc* d, void* dst) {
    if (dtype == 0) rt_fill<N, double>(d, static_cast<RtArm<N, double>*>(dst));
    else rt_fill<N, float>(d, static_cast<RtArm<N, float>*>(dst));
  }
};

#define ABRK_RT_DECL(NN)                                            \
  const ArmOps* ops_rt##NN();                                       \
  size_t rt_size##NN(int dtype);                                    \
  void rt_fill##NN(int dtype, const abrk_arm_desc* d, void* dst);
ABRK_RT_DECL(1) ABRK_RT_DECL(2) ABRK_RT_DECL(3) ABRK_RT_DECL(4) ABRK_RT_DECL(5) ABRK_RT_DECL(6) ABRK_RT_DECL(7)

#define ABRK_RT_DEFINE(NN)                                                                      \
  const ArmOps* ops_rt##NN() { return RtOps<NN>::ops(); }                                       \
  size_t rt_size##NN(int dtype) { return RtOps<NN>::size(dtype); }                              \
  void rt_fill##NN(int dtype, const abrk_arm_desc* d, void* dst) { RtOps<NN>::fill(dtype, d, dst); }

}  // namespace abrk
