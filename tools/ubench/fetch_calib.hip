// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE counters on gfx950 with kernels whose HBM traffic is known:
// each reads (or writes) a 1 GiB buffer exactly once — far beyond the 32 MB of L2 and the 256 MB infinity cache — with
// 4, 8 or 16 bytes per lane per load, lane-coalesced.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench/fetch_calib      (and again with --pmc WRITE_SIZE)
// and compare the counter (KB) of read_b32 / read_b64 / read_b128 / write_b64 with the 1 048 576 KB they move
// (tools/profile_gpu.sh does that and stores the factors in profiles/pmc_traffic.json).
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x)                                                         \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
      return 1;                                                          \
    }                                                                    \
  } while (0)

__global__ void read_b32(const unsigned* __restrict__ src, size_t n, double* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += stride) acc ^= src[i];
  if (acc == 0x12345678u) sink[0] = 1.0;  // (practically never: keeps the loads)
}
__global__ void read_b64(const double* __restrict__ src, size_t n, double* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i < n; i += stride) acc += src[i];
  if (acc == 12345.678) sink[0] = acc;
}
__global__ void read_b128(const double2* __restrict__ src, size_t n, double* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i < n; i += stride) {
    double2 v = src[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) sink[0] = acc;
}
// the access shape of the path stage's basis records: a lane moves one 64-byte record with four 16-byte loads
__global__ void read_rec64(const double2* __restrict__ src, size_t n_rec, double* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i < n_rec; i += stride) {
    const double2* p = src + 4 * i;
    double2 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.x + d.y;
  }
  if (acc == 12345.678) sink[0] = acc;
}
__global__ void write_b64(double* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (double)i;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void* buf = nullptr;
  double* sink = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(read_b32, grid, block, 0, 0, (const unsigned*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_b64, grid, block, 0, 0, (const double*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(read_b128, grid, block, 0, 0, (const double2*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(read_rec64, grid, block, 0, 0, (const double2*)buf, bytes / 64, sink);
    hipLaunchKernelGGL(write_b64, grid, block, 0, 0, (double*)buf, bytes / 8);
    CHECK(hipDeviceSynchronize());
  }
  printf("fetch_calib: 3 x (read_b32, read_b64, read_b128, read_rec64, write_b64) over %zu bytes each\n", bytes);
  return 0;
}
