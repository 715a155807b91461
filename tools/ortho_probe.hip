// measurement: which operation of the orthographic three-point solver rounds differently on the device? Every intermediate from the
// device (intrinsics as in mvgx_geofilter.hip) beside the host's plain IEEE evaluation (-ffp-contract=off).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
__device__ double dmul(double a, double b) { return __dmul_rn(a, b); }
__device__ double dadd(double a, double b) { return __dadd_rn(a, b); }
__global__ void k(const double* in, double* out) {
  const double p0x = in[0], p0y = in[1], p1x = in[2], p1y = in[3], p2x = in[4], p2y = in[5], q0x = in[6], q0y = in[7], q1x = in[8], q1y = in[9], q2x = in[10], q2y = in[11];
  int o = 0;
  const double xd1x = dadd(p1x, -p0x), xd1y = dadd(p1y, -p0y), yd1x = dadd(p2x, -p0x), yd1y = dadd(p2y, -p0y);
  const double xd2x = dadd(q1x, -q0x), xd2y = dadd(q1y, -q0y), yd2x = dadd(q2x, -q0x), yd2y = dadd(q2y, -q0y);
  const double denom = dadd(dmul(xd1x, yd1y), -dmul(xd1y, yd1x)); out[o++] = denom;
  const double n_aac = dadd(dmul(xd1y, yd2x), -dmul(xd2x, yd1y)); out[o++] = n_aac;
  const double aac = n_aac / denom; out[o++] = aac;
  const double aad = dadd(dmul(xd1y, yd2y), -dmul(xd2y, yd1y)) / denom; out[o++] = aad;
  const double bbc = dadd(dmul(xd2x, yd1x), -dmul(xd1x, yd2x)) / denom; out[o++] = bbc;
  const double bbd = dadd(dmul(xd2y, yd1x), -dmul(xd1x, yd2y)) / denom; out[o++] = bbd;
  const double aac_sq = dmul(aac, aac), bbc_sq = dmul(bbc, bbc);
  const double dd_2 = dadd(dadd(dadd(-aac_sq, dmul(aad, aad)), -bbc_sq), dmul(bbd, bbd)); out[o++] = dd_2;
  const double dd_1c = dadd(dmul(dmul(2.0, aac), aad), dmul(dmul(2.0, bbc), bbd)); out[o++] = dd_1c;
  const double dd_0 = dadd(dadd(aac_sq, bbc_sq), -1.0); out[o++] = dd_0;
  const double d4_4 = dadd(dmul(dd_1c, dd_1c), dmul(dd_2, dd_2)); out[o++] = d4_4;
  const double d4_2 = dadd(dmul(-dd_1c, dd_1c), dmul(dmul(2.0, dd_0), dd_2)); out[o++] = d4_2;
  const double d4_0 = dmul(dd_0, dd_0); out[o++] = d4_0;
  const double rad = dadd(dmul(d4_2, d4_2), -dmul(dmul(4.0, d4_4), d4_0)); out[o++] = rad;
  const double tmp = sqrt(rad); out[o++] = tmp;
  const double root = dadd(d4_2, tmp); out[o++] = root;
  const double t1 = -root / d4_4; out[o++] = t1;
  const double dsol = sqrt(dmul(t1, 0.5)); out[o++] = dsol;
  const double num = dadd(dadd(dadd(dmul(dmul(dd_2, dsol), dsol), aac_sq), bbc_sq), -1.0); out[o++] = num;
  const double den = dadd(dmul(dmul(dmul(2.0, aac), aad), dsol), dmul(dmul(dmul(2.0, bbc), bbd), dsol)); out[o++] = den;
  const double csol = -num / den; out[o++] = csol;
  out[o++] = dadd(dmul(aac, csol), dmul(aad, dsol));
  out[o++] = dadd(dmul(bbc, csol), dmul(bbd, dsol));
}
int main() {
  const char* names[] = {"denom", "n_aac", "aac", "aad", "bbc", "bbd", "dd_2", "dd_1c", "dd_0", "d4_4", "d4_2", "d4_0", "rad", "tmp", "root", "t1", "dsol", "num", "den", "csol", "asol", "bsol"};
  std::mt19937 g(5);
  std::uniform_real_distribution<double> u(-0.5, 0.5);
  double *din, *dout;
  hipMalloc(&din, 12 * 8); hipMalloc(&dout, 32 * 8);
  int bad[22] = {0};
  for (int t = 0; t < 2000; ++t) {
    double in[12], out[32];
    // a planar-motion-like configuration: second view = small rotation + shift of the first + noise
    for (int i = 0; i < 3; ++i) { in[2 * i] = u(g); in[2 * i + 1] = u(g); in[6 + 2 * i] = 0.98 * in[2 * i] + 0.05 * in[2 * i + 1] + 0.1 + 1e-3 * u(g); in[7 + 2 * i] = in[2 * i + 1] + 1e-3 * u(g); }
    hipMemcpy(din, in, sizeof(in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, din, dout);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    volatile double p0x = in[0], p0y = in[1], p1x = in[2], p1y = in[3], p2x = in[4], p2y = in[5], q0x = in[6], q0y = in[7], q1x = in[8], q1y = in[9], q2x = in[10], q2y = in[11];
    double h[22]; int o = 0;
    const double xd1x = p1x - p0x, xd1y = p1y - p0y, yd1x = p2x - p0x, yd1y = p2y - p0y, xd2x = q1x - q0x, xd2y = q1y - q0y, yd2x = q2x - q0x, yd2y = q2y - q0y;
    const double denom = xd1x * yd1y - xd1y * yd1x; h[o++] = denom;
    const double n_aac = xd1y * yd2x - xd2x * yd1y; h[o++] = n_aac;
    const double aac = n_aac / denom; h[o++] = aac;
    const double aad = (xd1y * yd2y - xd2y * yd1y) / denom; h[o++] = aad;
    const double bbc = (xd2x * yd1x - xd1x * yd2x) / denom; h[o++] = bbc;
    const double bbd = (xd2y * yd1x - xd1x * yd2y) / denom; h[o++] = bbd;
    const double aac_sq = aac * aac;
    const double dd_2 = -aac_sq + aad * aad - bbc * bbc + bbd * bbd; h[o++] = dd_2;
    const double dd_1c = 2.0 * aac * aad + 2.0 * bbc * bbd; h[o++] = dd_1c;
    const double dd_0 = aac_sq + bbc * bbc - 1.0; h[o++] = dd_0;
    const double d4_4 = dd_1c * dd_1c + dd_2 * dd_2; h[o++] = d4_4;
    const double d4_2 = -dd_1c * dd_1c + 2.0 * dd_0 * dd_2; h[o++] = d4_2;
    const double d4_0 = dd_0 * dd_0; h[o++] = d4_0;
    const double rad = d4_2 * d4_2 - 4.0 * d4_4 * d4_0; h[o++] = rad;
    const double tmp = std::sqrt(rad); h[o++] = tmp;
    const double root = d4_2 + tmp; h[o++] = root;
    const double t1 = -root / d4_4; h[o++] = t1;
    const double dsol = std::sqrt(t1 / 2.0); h[o++] = dsol;
    const double num = dd_2 * dsol * dsol + aac_sq + bbc * bbc - 1.0; h[o++] = num;
    const double den = 2.0 * aac * aad * dsol + 2.0 * bbc * bbd * dsol; h[o++] = den;
    const double csol = -num / den; h[o++] = csol;
    h[o++] = aac * csol + aad * dsol;
    h[o++] = bbc * csol + bbd * dsol;
    for (int i = 0; i < 22; ++i) {
      const bool same = (out[i] == h[i]) || (out[i] != out[i] && h[i] != h[i]);
      if (!same && bad[i]++ < 2) printf("trial %d %s device %.17g host %.17g\n", t, names[i], out[i], h[i]);
    }
  }
  printf("mismatches per intermediate over 2000 trials:");
  for (int i = 0; i < 22; ++i) printf(" %s=%d", names[i], bad[i]);
  printf("\n");
  return 0;
}
