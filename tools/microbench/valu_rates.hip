// VALU issue-rate microbenchmark for gfx950: independent FMA chains in registers, no memory traffic.
// Answers the question the packed-FP32 design depends on: does v_pk_fma_f32 retire two fp32 FMAs per lane in the issue
// time of one v_fma_f32?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(64, 2) chains(float* out, int iters, float seed) {
  // 16 independent accumulator chains per lane
  float a[16];
  float2v p[16];
  double d[16];
  for (int i = 0; i < 16; i++) {
    a[i] = seed + i;
    p[i] = float2v{seed + i, seed - i};
    d[i] = seed + i;
  }
  const float m = 1.0000001f, c = 1e-7f;
  const float2v pm = {m, m}, pc = {c, c};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);
      if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], pm, pc);
      if (KIND == 2) d[i] = __builtin_fma(d[i], (double)m, (double)c);
      if (KIND == 3) p[i] = p[i] * pm;
      if (KIND == 4) p[i] = p[i] + pc;
      // round 3: what the small factorisations are made of besides FMAs
      if (KIND == 5) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
      if (KIND == 6) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
      if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
      if (KIND == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)m));
      if (KIND == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)c));
      if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 11) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 12) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[i]), "v"((double)c) : "vcc");
      if (KIND == 13) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
      if (KIND == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[2:3]" : "+v"(a[i]) : "v"(c));
      if (KIND == 15) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(m), "v"(c));
      if (KIND == 16) d[i] = d[i] > (double)seed ? d[i] : (double)c;  // compiler's own double select
      if (KIND == 17) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 18) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"((double)c));
      if (KIND == 19) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
      if (KIND == 20) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
      if (KIND == 21) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"((double)m), "s"((double)c));
      if (KIND == 22) asm volatile("v_accvgpr_write_b32 a0, %0" : : "v"(a[i]) : "a0");
      if (KIND == 23) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 24) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 25) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
double run(const char* name, int flops_per_instr_lane) {
  const int blocks = 256 * 4 * 2 * 4, iters = 20000;  // 2 waves per SIMD, 4 rounds
  float* out;
  (void)hipMalloc(&out, blocks * 64 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(chains<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double instr = (double)blocks * 64 * iters * 16;  // lane-instructions
  const double rate = instr / (best * 1e-3);
  printf("%-20s %8.3f ms  %7.2f T lane-instr/s  = %5.1f lanes/clk/SIMD at 2.4 GHz  (%6.1f TFLOP/s)\n", name, best, rate / 1e12,
         rate / (1024 * 2.4e9), rate * flops_per_instr_lane / 1e12);
  (void)hipFree(out);
  return rate;
}

int main() {
  run<0>("v_fma_f32", 2);
  run<1>("v_pk_fma_f32", 4);
  run<3>("v_pk_mul_f32", 2);
  run<4>("v_pk_add_f32", 2);
  run<2>("v_fma_f64", 2);
  run<8>("v_mul_f64", 1);
  run<9>("v_add_f64", 1);
  run<5>("v_rsq_f64", 1);
  run<6>("v_rcp_f64", 1);
  run<13>("v_sqrt_f64", 1);
  run<11>("v_rsq_f32", 1);
  run<12>("v_cmp_gt_f64", 1);
  run<7>("v_cndmask_b32", 1);
  run<10>("v_mov_b32", 1);
  run<14>("v_cndmask_e64 sgpr", 1);
  run<15>("v_cndmask nodep", 1);
  run<16>("double select (cc)", 1);
  run<17>("v_xor_b32", 1);
  run<18>("v_max_f64", 1);
  run<19>("v_rndne_f64", 1);
  run<20>("v_cvt_i32_f64", 1);
  run<21>("v_fma_f64 sgpr c", 2);
  run<22>("v_accvgpr_write", 1);
  run<23>("v_add_f32", 1);
  run<24>("v_add_u32", 1);
  run<25>("v_lshlrev_b32", 1);
  return 0;
}
