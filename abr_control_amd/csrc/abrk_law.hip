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
template <int N, class T>
static hipError_t finish_launch(const LaunchArgs& la, const FinishArgs& a) {
  const unsigned nchunk = (unsigned)((la.B + kBlock - 1) / kBlock);
  hipLaunchKernelGGL((osc6_finish_kernel<N, T>), dim3(nchunk, (unsigned)a.slots), dim3(kBlock), 0, la.stream,
                     (const unsigned long long*)a.masks, (const T*)a.rec, a.nulls, a.coop_rounds, (T*)a.u, (T*)a.ts);
  return hipGetLastError();
}
hipError_t launch_osc6_finish(int n, int dtype, const LaunchArgs& la, const FinishArgs& a) {
  if (a.slots < 1 || a.slots > kBlock || la.B < 1 || la.B > kHandoverMaxRows) return hipErrorInvalidValue;
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
