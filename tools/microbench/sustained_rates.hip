// Sustained (power-limited) VALU issue rates on gfx950: each instruction mix runs back to back for ~1.2 s and the rate of
// the last 0.6 s is reported - valu_rates.hip's 5 ms launches run at boost clocks, the HBM-sized kernels of this
// library do not.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/sustained_rates.hip -o /tmp/sustained && /tmp/sustained
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(64, 2) chains(float* out, int iters, float seed) {
  double d[16];
  float a[16];
  double e[16];
  for (int i = 0; i < 16; i++) {
    d[i] = seed + i;
    a[i] = seed - i;
    e[i] = 1.0 + 1e-9 * (seed + i);  // stays in registers: not foldable (seed is a kernel argument)
  }
  const double m = 1.0000001, c = 1e-7;
  const float mf = 1.0000001f, cf = 1e-7f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (KIND == 0) d[i] = __builtin_fma(d[i], m, c);                 // v_fma_f64
      if (KIND == 1) d[i] = d[i] * m;                                   // v_mul_f64
      if (KIND == 2) d[i] = d[i] + c;                                   // v_add_f64
      if (KIND == 3) a[i] = __builtin_fmaf(a[i], mf, cf);               // v_fma_f32
      if (KIND == 5) d[i] = __builtin_fma(d[i], e[i], e[(i + 5) & 15]);  // v_fma_f64, three VGPR operands
      if (KIND == 4) {                                                  // 3 fp64 : 1 fp32 interleaved
        if (i % 4 == 3) a[i] = __builtin_fmaf(a[i], mf, cf);
        else d[i] = __builtin_fma(d[i], m, c);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 16; i++) s += d[i] + a[i] + e[i];
  out[blockIdx.x * 64 + threadIdx.x] = (float)s;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd * 4, iters = 20000;
  float* out;
  (void)hipMalloc(&out, blocks * 64 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto t0 = std::chrono::steady_clock::now();
  auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  float first = 0;
  int n = 0;
  while (secs() < 0.6) {  // heat up
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(chains<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (n++ == 1) first = ms;
  }
  double sum = 0;
  int k = 0;
  while (secs() < 1.2) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(chains<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    sum += ms;
    k++;
  }
  const double instr = (double)blocks * 64 * iters * 16;
  printf("%-28s %d waves/SIMD  first %7.3f ms (%6.2f T/s)   sustained %7.3f ms (%6.2f T lane-instr/s = %4.2f GHz-equivalent at full issue)\n",
         name, waves_per_simd, first, instr / (first * 1e-3) / 1e12, sum / k, instr / (sum / k * 1e-3) / 1e12,
         instr / (sum / k * 1e-3) / (1024.0 * (KIND == 3 ? 32 : 16)) / 1e9);
  (void)hipFree(out);
}

int main() {
  run<0>("v_fma_f64", 2);
  run<1>("v_mul_f64", 2);
  run<2>("v_add_f64", 2);
  run<3>("v_fma_f32", 2);
  run<4>("3 v_fma_f64 : 1 v_fma_f32", 2);
  run<5>("v_fma_f64, 3 VGPR operands", 2);
  run<0>("v_fma_f64", 1);
  return 0;
}
