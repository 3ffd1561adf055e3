// Is v_mfma_f64_16x16x4_f64 bit-equal to the inner product the reference's NumPy gets from OpenBLAS?
//
// north_star reserves MFMA for a "normal-equation GEMM" the reference does not have (SciPy's splprep is a Givens QR,
// DESIGN.md "Why there is no MFMA").  The one GEMM-shaped piece of the path is the N x 6 x N squared-distance matrix of the
// sorter (utils/math_utils.py:120-150: np.dot of [1, 1, a0, a1, a0^2, a1^2]-style rows), whose elements feed the kNN
// order — bit-exactness matters.  OpenBLAS evaluates one element as   acc = x0*y0; acc = fma(x_k, y_k, acc), k = 1..5.
// This micro-benchmark asks whether two chained MFMA issues (K = 4 + 2 zero-padded, the first one's result as the second
// one's C) return exactly that, on random rows incl. cancellation cases, by comparing every output element with
//     (a) the ascending fma chain from C        (b) the descending chain        (c) the exactly rounded sum of products.
// It also times the MFMA against the VALU fma chain for the same 16 x 16 x 6 tile.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/mfma_f64_order.hip -o tools/ubench/mfma_f64_order
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));

// One wavefront, one 16 x 16 tile per trip.  The operands arrive per lane and issue (the host applies the instruction's
// operand layout, which main() first establishes by experiment); the four result registers leave per lane.
__global__ void __launch_bounds__(64) mfma_tiles(const double* __restrict__ a_lane, const double* __restrict__ b_lane, double* __restrict__ d_lane,
                                                 int tiles, int issues) {
  const int l = threadIdx.x;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < issues; q++)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_lane[((size_t)t * issues + q) * 64 + l], b_lane[((size_t)t * issues + q) * 64 + l], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) d_lane[((size_t)t * 64 + l) * 4 + v] = acc[v];
  }
}

// the same tile on the VALU: element (i, j) = OpenBLAS order chain; four elements per lane like the MFMA's output
__global__ void __launch_bounds__(64) valu_tiles(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ D, int tiles,
                                                 int kk) {
  const int l = threadIdx.x;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const double* a = A + (size_t)t * 16 * kk;
    const double* b = B + (size_t)t * kk * 16;
    for (int v = 0; v < 4; v++) {
      const int i = 4 * (l / 16) + v, j = l % 16;
      double acc = a[i * kk] * b[j];
      for (int k = 1; k < kk; k++) acc = fma(a[i * kk + k], b[k * 16 + j], acc);
      D[(size_t)t * 256 + i * 16 + j] = acc;
    }
  }
}

// exactly rounded sum of products (Shewchuk expansions: two_prod / grow_expansion, then the sum's leading component)
static void two_sum(double a, double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}
static double exact_dot(const double* x, const double* y, int n) {
  std::vector<double> ex;  // non-overlapping expansion, increasing magnitude
  auto grow = [&](double v) {
    std::vector<double> out;
    double q = v;
    for (double e : ex) {
      double s, r;
      two_sum(q, e, s, r);
      if (r != 0.0) out.push_back(r);
      q = s;
    }
    if (q != 0.0) out.push_back(q);
    ex.swap(out);
  };
  for (int k = 0; k < n; k++) {
    const double p = x[k] * y[k], e = fma(x[k], y[k], -p);
    grow(e);
    grow(p);
  }
  // the components do not overlap and grow in magnitude: summed smallest first in 64-bit-mantissa arithmetic the result is the
  // exact sum up to a double rounding in rare near-tie cases (a statistic is all this is used for)
  long double acc = 0.0L;
  for (double e : ex) acc += (long double)e;
  return (double)acc;
}

int main() {
  const int kk = 6, tiles = 4096;  // 4096 x 256 = 1 048 576 output elements
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-60.0, 60.0);
  std::vector<double> A((size_t)tiles * 16 * kk), B((size_t)tiles * kk * 16), D((size_t)tiles * 256), Dv(D.size());
  for (int t = 0; t < tiles; t++) {
    // rows in the shape the sorter feeds: [1, 1, -2 a0, -2 a1, a0^2, a1^2] . [b0^2, b1^2, b0, b1, 1, 1] = |a - b|^2 — heavy cancellation
    // for nearby cones; every fourth tile plain random operands over 12 orders of magnitude
    for (int i = 0; i < 16; i++) {
      const double a0 = U(rng), a1 = U(rng);
      double* r = &A[((size_t)t * 16 + i) * kk];
      if (t % 4 == 3) {
        for (int k = 0; k < kk; k++) r[k] = U(rng) * std::pow(10.0, (double)(rng() % 13) - 6.0);
      } else {
        r[0] = 1.0, r[1] = 1.0, r[2] = a0, r[3] = a1, r[4] = a0 * a0, r[5] = a1 * a1;
      }
    }
    for (int j = 0; j < 16; j++) {
      const double b0 = (t % 2) ? A[((size_t)t * 16 + j) * kk + 2] + 1e-3 * U(rng) : U(rng), b1 = (t % 2) ? A[((size_t)t * 16 + j) * kk + 3] + 1e-3 * U(rng) : U(rng);
      double c[6] = {b0 * b0, b1 * b1, -2 * b0, -2 * b1, 1.0, 1.0};
      if (t % 4 == 3)
        for (int k = 0; k < kk; k++) c[k] = U(rng) * std::pow(10.0, (double)(rng() % 13) - 6.0);
      for (int k = 0; k < kk; k++) B[((size_t)t * kk + k) * 16 + j] = c[k];
    }
  }
  double *dA, *dB, *dD;
  (void)hipMalloc(&dA, A.size() * 8);
  (void)hipMalloc(&dB, B.size() * 8);
  (void)hipMalloc(&dD, D.size() * 8);
  (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  // ---- operand layout by experiment: A = [1, i + 1, 0, 0], B = [100 (j + 1); 1; 0; 0] -> D(i, j) = 100 (j + 1) + i + 1, exact and distinct;
  //      then k-sensitivity: which lane supplies which k.  Candidates: lane l supplies (row l % 16, k = l / 16) or (row l / 4, k = l % 4).
  const int issues = 2;
  double *dal, *dbl, *ddl;
  (void)hipMalloc(&dal, (size_t)tiles * issues * 64 * 8);
  (void)hipMalloc(&dbl, (size_t)tiles * issues * 64 * 8);
  (void)hipMalloc(&ddl, (size_t)tiles * 256 * 8);
  int a_mode = -1, b_mode = -1;
  int out_i[64][4], out_j[64][4];
  for (int am = 0; am < 2 && a_mode < 0; am++)
    for (int bm = 0; bm < 2 && a_mode < 0; bm++) {
      std::vector<double> al(128, 0.0), bl(128, 0.0), dl(256);
      double Ac[16][4], Bc[4][16];
      for (int i = 0; i < 16; i++)
        for (int k = 0; k < 4; k++) Ac[i][k] = k == 0 ? 1.0 : (k == 1 ? i + 1.0 : (k == 2 ? 1e-3 * (i + 1) : 0.0));
      for (int k = 0; k < 4; k++)
        for (int j = 0; j < 16; j++) Bc[k][j] = k == 0 ? 100.0 * (j + 1) : (k == 1 ? 1.0 : (k == 2 ? 0.0 : 7.0));
      for (int l = 0; l < 64; l++) {
        const int ai = am == 0 ? l % 16 : l / 4, ak = am == 0 ? l / 16 : l % 4;
        const int bj = bm == 0 ? l % 16 : l / 4, bk = bm == 0 ? l / 16 : l % 4;
        al[l] = Ac[ai][ak];
        bl[l] = Bc[bk][bj];
      }
      (void)hipMemcpy(dal, al.data(), 128 * 8, hipMemcpyHostToDevice);
      (void)hipMemcpy(dbl, bl.data(), 128 * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(mfma_tiles, dim3(1), dim3(64), 0, 0, dal, dbl, ddl, 1, 1);
      (void)hipMemcpy(dl.data(), ddl, 256 * 8, hipMemcpyDeviceToHost);
      bool ok = true;
      bool seen[16][16] = {};
      for (int l = 0; l < 64 && ok; l++)
        for (int v = 0; v < 4 && ok; v++) {
          const double d = dl[l * 4 + v];
          const int j = (int)(d / 100.0) - 1, i = (int)std::lround(d - 100.0 * (j + 1)) - 1;
          if (j < 0 || j > 15 || i < 0 || i > 15 || d != 100.0 * (j + 1) + (i + 1) || seen[i][j]) ok = false;
          else seen[i][j] = true, out_i[l][v] = i, out_j[l][v] = j;
        }
      if (ok) a_mode = am, b_mode = bm;
    }
  if (a_mode < 0) {
    printf("could not establish the operand layout\n");
    return 1;
  }
  printf("operand layout (by experiment): lane l supplies A[%s], B[%s]; D register v of lane l = element [%d*(l/16)+%d*v][l%%16] (lane 17: v0 -> (%d, %d), v1 -> (%d, %d))\n",
         a_mode == 0 ? "l % 16][l / 16" : "l / 4][l % 4", b_mode == 0 ? "l / 16][l % 16" : "l % 4][l / 4", out_i[16][0] - out_i[0][0], out_i[0][1] - out_i[0][0],
         out_i[17][0], out_j[17][0], out_i[17][1], out_j[17][1]);
  {
    std::vector<double> al((size_t)tiles * issues * 64), bl(al.size());
    for (int t = 0; t < tiles; t++)
      for (int q = 0; q < issues; q++)
        for (int l = 0; l < 64; l++) {
          const int ai = a_mode == 0 ? l % 16 : l / 4, ak = 4 * q + (a_mode == 0 ? l / 16 : l % 4);
          const int bj = b_mode == 0 ? l % 16 : l / 4, bk = 4 * q + (b_mode == 0 ? l / 16 : l % 4);
          al[((size_t)t * issues + q) * 64 + l] = ak < kk ? A[((size_t)t * 16 + ai) * kk + ak] : 0.0;
          bl[((size_t)t * issues + q) * 64 + l] = bk < kk ? B[((size_t)t * kk + bk) * 16 + bj] : 0.0;
        }
    (void)hipMemcpy(dal, al.data(), al.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dbl, bl.data(), bl.size() * 8, hipMemcpyHostToDevice);
  }
  hipLaunchKernelGGL(mfma_tiles, dim3(1024), dim3(64), 0, 0, dal, dbl, ddl, tiles, issues);
  {
    std::vector<double> dl((size_t)tiles * 256);
    (void)hipMemcpy(dl.data(), ddl, dl.size() * 8, hipMemcpyDeviceToHost);
    for (int t = 0; t < tiles; t++)
      for (int l = 0; l < 64; l++)
        for (int v = 0; v < 4; v++) D[(size_t)t * 256 + out_i[l][v] * 16 + out_j[l][v]] = dl[((size_t)t * 64 + l) * 4 + v];
  }
  hipLaunchKernelGGL(valu_tiles, dim3(1024), dim3(64), 0, 0, dA, dB, dD, tiles, kk);
  (void)hipMemcpy(Dv.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);

  long long n = 0, eq_asc = 0, eq_desc = 0, eq_exact = 0, eq_valu = 0, eq_asc2 = 0;
  double worst = 0.0;
  int shown = 0;
  for (int t = 0; t < tiles; t++)
    for (int i = 0; i < 16; i++)
      for (int j = 0; j < 16; j++) {
        double x[6], y[6];
        for (int k = 0; k < kk; k++) x[k] = A[((size_t)t * 16 + i) * kk + k], y[k] = B[((size_t)t * kk + k) * 16 + j];
        double asc = x[0] * y[0];  // OpenBLAS: acc = x0*y0; fma(x_k, y_k, acc)
        for (int k = 1; k < kk; k++) asc = fma(x[k], y[k], asc);
        double desc = 0.0;  // each issue descending, the issues in order
        {
          double a1 = 0.0;
          for (int k = 3; k >= 0; k--) a1 = fma(x[k], y[k], a1);
          desc = a1;
          for (int k = 5; k >= 4; k--) desc = fma(x[k], y[k], desc);
        }
        double asc2;  // per issue: the exactly rounded sum of its four products + C (a fused dot product per issue)
        {
          double xx[5] = {x[0], x[1], x[2], x[3], 1.0}, yy[5] = {y[0], y[1], y[2], y[3], 0.0};
          const double c1 = exact_dot(xx, yy, 4);
          double x2[3] = {x[4], x[5], 1.0}, y2[3] = {y[4], y[5], c1};
          asc2 = exact_dot(x2, y2, 3);
        }
        const double ex = exact_dot(x, y, kk);
        const double d = D[(size_t)t * 256 + i * 16 + j];
        n++;
        eq_asc += d == asc;
        eq_desc += d == desc;
        eq_asc2 += d == asc2;
        eq_exact += d == ex;
        eq_valu += Dv[(size_t)t * 256 + i * 16 + j] == asc;
        if (d != asc) {
          const double rel = std::fabs(d - asc) / std::fmax(std::fabs(asc), 1e-300);
          worst = std::fmax(worst, rel);
          if (shown < 3) {
            printf("  counter-example (tile %d, i %d, j %d): mfma %a  openblas-order chain %a  exact %a\n", t, i, j, d, asc, ex);
            shown++;
          }
        }
      }
  printf("v_mfma_f64_16x16x4_f64, K = 6 as two chained issues (4 + 2 zero-padded), %lld output elements:\n", n);
  printf("  == OpenBLAS-order fma chain (ascending k from 0):            %lld (%.4f %%)\n", eq_asc, 100.0 * eq_asc / n);
  printf("  == descending-k chain per issue:                              %lld (%.4f %%)\n", eq_desc, 100.0 * eq_desc / n);
  printf("  == one exactly rounded dot product per issue (4 products + C): %lld (%.4f %%)\n", eq_asc2, 100.0 * eq_asc2 / n);
  printf("  == exactly rounded 6-term dot product:                        %lld (%.4f %%)\n", eq_exact, 100.0 * eq_exact / n);
  printf("  worst relative difference from the OpenBLAS-order chain: %.3e\n", worst);
  printf("  (control) the VALU chain kernel == host chain: %lld of %lld\n", eq_valu, n);

  // time: 64 x the tile set
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms_m = 0, ms_v = 0;
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    for (int r = 0; r < 64; r++) hipLaunchKernelGGL(mfma_tiles, dim3(1024), dim3(64), 0, 0, dal, dbl, ddl, tiles, issues);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms_m, e0, e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 64; r++) hipLaunchKernelGGL(valu_tiles, dim3(1024), dim3(64), 0, 0, dA, dB, dD, tiles, kk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms_v, e0, e1);
  }
  printf("  time for 64 x %d tiles (loads from L2 / HBM included): mfma %.3f ms, valu chain %.3f ms\n", tiles, ms_m, ms_v);
  return 0;
}
