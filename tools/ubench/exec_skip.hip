// Micro-benchmark: does a wave64 FP64 VALU instruction cost less when only lanes 0..15 (one 16-lane pass) are active?
// Also: dependent-chain latency vs independent issue rate of v_fma_f64, v_rcp_f64, v_sqrt/div sequences for ONE wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ACTIVE, int CHAINS>
__global__ void fma_chain(double* out, long long* cyc, int iters, double a, double b) {
  const int lane = threadIdx.x & 63;
  double x[CHAINS];
  for (int c = 0; c < CHAINS; c++) x[c] = a + c + lane;
  long long t0 = 0, t1 = 0;
  if (lane < ACTIVE) {
    t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) x[c] = fma(x[c], b, a);
    }
    t1 = clock64();
  }
  double s = 0;
  for (int c = 0; c < CHAINS; c++) s += x[c];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ACTIVE>
__global__ void div_chain(double* out, long long* cyc, int iters, double a, double b) {
  const int lane = threadIdx.x & 63;
  double x = a + lane;
  long long t0 = 0, t1 = 0;
  if (lane < ACTIVE) {
    t0 = clock64();
    for (int i = 0; i < iters; i++) x = b / x + a;
    t1 = clock64();
  }
  out[blockIdx.x * 64 + lane] = x;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ACTIVE>
__global__ void sqrt_chain(double* out, long long* cyc, int iters, double a, double b) {
  const int lane = threadIdx.x & 63;
  double x = a + lane;
  long long t0 = 0, t1 = 0;
  if (lane < ACTIVE) {
    t0 = clock64();
    for (int i = 0; i < iters; i++) x = sqrt(x) + a;
    t1 = clock64();
  }
  out[blockIdx.x * 64 + lane] = x;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char* name, K kern, int iters, int ops_per_iter) {
  double* out;
  long long* cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, sizeof(long long));
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, cyc, iters, 1.000001, 0.999999);
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, out, cyc, iters, 1.000001, 0.999999);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%-40s %8.2f cycles per op\n", name, (double)c / ((double)iters * ops_per_iter));
  hipFree(out);
  hipFree(cyc);
}

int main() {
  const int it = 20000;
  run("fma dependent, 64 lanes", fma_chain<64, 1>, it, 1);
  run("fma dependent, 16 lanes", fma_chain<16, 1>, it, 1);
  run("fma dependent, 4 lanes", fma_chain<4, 1>, it, 1);
  run("fma 8 independent chains, 64 lanes", fma_chain<64, 8>, it, 8);
  run("fma 8 independent chains, 16 lanes", fma_chain<16, 8>, it, 8);
  run("fma 8 independent chains, 4 lanes", fma_chain<4, 8>, it, 8);
  run("div+add dependent, 64 lanes", div_chain<64>, it, 1);
  run("div+add dependent, 16 lanes", div_chain<16>, it, 1);
  run("sqrt+add dependent, 64 lanes", sqrt_chain<64>, it, 1);
  run("sqrt+add dependent, 16 lanes", sqrt_chain<16>, it, 1);
  return 0;
}
