// Chip-level issue rate of FP64 VALU instructions on gfx950: W wavefronts per SIMD, each running 8 independent
// v_fma_f64 / v_mul_f64 / v_add_f64 chains (no memory traffic).  Prints wave-instructions per SIMD-cycle-quad and TFLOP/s.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/fp64_rate.hip -o tools/ubench/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(64) chains(double* out, int iters, double a, double b) {
  double x[8];
  for (int c = 0; c < 8; c++) x[c] = a + c + (threadIdx.x & 63);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (OP == 0) x[c] = fma(x[c], b, a);
      if (OP == 1) x[c] = x[c] * b;
      if (OP == 2) x[c] = x[c] + a;
    }
  }
  double s = 0;
  for (int c = 0; c < 8; c++) s += x[c];
  if (s == 12345.678) out[0] = s;
}
template <int OP>
__global__ void __launch_bounds__(64) chains32(float* out, int iters, float a, float b) {
  float x[8];
  for (int c = 0; c < 8; c++) x[c] = a + c + (threadIdx.x & 63);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++) x[c] = fmaf(x[c], b, a);
  }
  float s = 0;
  for (int c = 0; c < 8; c++) s += x[c];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  double* out;
  hipMalloc(&out, 64);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 200000;
  const double ghz = p.clockRate * 1e-6;
  printf("%d CUs, %.2f GHz (reported max)\n", cus, ghz);
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32"};
  for (int op = 0; op < 4; op++)
    for (int w : {1, 2, 3, 4}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      dim3 grid(cus * 4 * w);
      auto launch = [&]() {
        if (op == 0) hipLaunchKernelGGL(chains<0>, grid, dim3(64), 0, 0, out, iters, 1.000001, 0.999999);
        if (op == 1) hipLaunchKernelGGL(chains<1>, grid, dim3(64), 0, 0, out, iters, 1.000001, 0.999999);
        if (op == 2) hipLaunchKernelGGL(chains<2>, grid, dim3(64), 0, 0, out, iters, 1.000001, 0.999999);
        if (op == 3) hipLaunchKernelGGL(chains32<0>, grid, dim3(64), 0, 0, (float*)out, iters, 1.000001f, 0.999999f);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)grid.x * iters * 8;  // wave-instructions
      const double per_simd_cycle = insts / (cus * 4.0) / (ms * 1e-3 * ghz * 1e9);
      printf("%-10s %d wave(s)/SIMD: %7.3f ms, %.3f wave-instructions per SIMD cycle (= %.2f cycles each), %.1f T%s/s\n", names[op], w, ms,
             per_simd_cycle, 1.0 / per_simd_cycle, insts * 64 * (op == 0 || op == 3 ? 2 : 1) / (ms * 1e-3) / 1e12, "FLOP");
    }
  return 0;
}
