// What does the shader clock do under this path's kind of load?  Every wavefront runs FP64 FMA / mul / add chains (the mix of the
// spline kernels) for tens of milliseconds and reads both timers around them: s_memtime (clock64(): shader-clock cycles) and
// s_memrealtime (wall_clock64(): constant 100 MHz).  cycles / realtime = the clock the SIMDs actually ran at — what
// "cycles per instruction" figures and the VALU peak of DESIGN.md should be priced with, instead of the 2.4 GHz the device reports.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/clock_under_load.hip -o tools/ubench/clock_under_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) load(double* out, long long* stamps, int iters, double a, double b) {
  double x[8];
  for (int c = 0; c < 8; c++) x[c] = a + c + (threadIdx.x & 63);
  const long long c0 = clock64(), r0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      x[c] = fma(x[c], b, a);
      x[c] = x[c] * b;
      x[c] = x[c] + a;
    }
  }
  const long long c1 = clock64(), r1 = wall_clock64();
  double s = 0;
  for (int c = 0; c < 8; c++) s += x[c];
  if (s == 12345.678) out[0] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  double* out;
  long long* st;
  (void)hipMalloc(&out, 64);
  for (int waves = 1; waves <= 4; waves++) {
    const int blocks = p.multiProcessorCount * 4 * waves;
    (void)hipMalloc(&st, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ev_ms = 0.f;
    for (int rep = 0; rep < 2; rep++) {  // (the second repetition: clocks settled)
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(load, dim3(blocks), dim3(64), 0, 0, out, st, 60000, 1.0000001, 0.9999999);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ev_ms, e0, e1);
    }
    std::vector<long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), st, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; i++) cyc += (double)h[2 * i], rt += (double)h[2 * i + 1];
    const double ghz = cyc / rt * 0.1;  // realtime ticks at 100 MHz
    const double insts = 60000.0 * 24;   // per wavefront
    // (a wavefront lives shorter than the kernel: the wavefronts of a SIMD do not all start together — the average number
    //  resident is waves x lifetime / kernel duration)
    const double life_ms = rt / blocks / 1e5;
    printf("%d wavefront(s) per SIMD on all %d CUs: kernel %.2f ms (HIP events), a wavefront lives %.2f ms (s_memrealtime, 100 MHz) = %.2f resident per SIMD on average; "
           "shader clock s_memtime / s_memrealtime = %.3f GHz (device reports %.2f); %.2f ns per FP64 wave-instruction and SIMD over the kernel = %.2f cycles at that clock\n",
           waves, p.multiProcessorCount, ev_ms, life_ms, waves * life_ms / ev_ms, ghz, p.clockRate / 1e6, ev_ms * 1e6 / (insts * waves),
           ev_ms * 1e6 / (insts * waves) * ghz);
    (void)hipFree(st);
  }
  return 0;
}
