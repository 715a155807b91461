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
  const double u1x = dadd(p1x, -p0x), u1y = dadd(p1y, -p0y), v1x = dadd(p2x, -p0x), v1y = dadd(p2y, -p0y);
  const double u2x = dadd(q1x, -q0x), u2y = dadd(q1y, -q0y), v2x = dadd(q2x, -q0x), v2y = dadd(q2y, -q0y);
  const double denom = dadd(dmul(u1x, v1y), -dmul(u1y, v1x)); out[o++] = denom;
  const double n_aac = dadd(dmul(u1y, v2x), -dmul(u2x, v1y)); out[o++] = n_aac;
  const double ac = n_aac / denom; out[o++] = ac;
  const double ad = dadd(dmul(u1y, v2y), -dmul(u2y, v1y)) / denom; out[o++] = ad;
  const double bc = dadd(dmul(u2x, v1x), -dmul(u1x, v2x)) / denom; out[o++] = bc;
  const double bd = dadd(dmul(u2y, v1x), -dmul(u1x, v2y)) / denom; out[o++] = bd;
  const double ac2 = dmul(ac, ac), bc2 = dmul(bc, bc);
  const double g2 = dadd(dadd(dadd(-ac2, dmul(ad, ad)), -bc2), dmul(bd, bd)); out[o++] = g2;
  const double g1 = dadd(dmul(dmul(2.0, ac), ad), dmul(dmul(2.0, bc), bd)); out[o++] = g1;
  const double g0 = dadd(dadd(ac2, bc2), -1.0); out[o++] = g0;
  const double h4 = dadd(dmul(g1, g1), dmul(g2, g2)); out[o++] = h4;
  const double h2 = dadd(dmul(-g1, g1), dmul(dmul(2.0, g0), g2)); out[o++] = h2;
  const double h0 = dmul(g0, g0); out[o++] = h0;
  const double rad = dadd(dmul(h2, h2), -dmul(dmul(4.0, h4), h0)); out[o++] = rad;
  const double rdisc = sqrt(rad); out[o++] = rdisc;
  const double root = dadd(h2, rdisc); out[o++] = root;
  const double t1 = -root / h4; out[o++] = t1;
  const double sd = sqrt(dmul(t1, 0.5)); out[o++] = sd;
  const double num = dadd(dadd(dadd(dmul(dmul(g2, sd), sd), ac2), bc2), -1.0); out[o++] = num;
  const double den = dadd(dmul(dmul(dmul(2.0, ac), ad), sd), dmul(dmul(dmul(2.0, bc), bd), sd)); out[o++] = den;
  const double sc = -num / den; out[o++] = sc;
  out[o++] = dadd(dmul(ac, sc), dmul(ad, sd));
  out[o++] = dadd(dmul(bc, sc), dmul(bd, sd));
}
int main() {
  const char* names[] = {"denom", "n_aac", "ac", "ad", "bc", "bd", "g2", "g1", "g0", "h4", "h2", "h0", "rad", "rdisc", "root", "t1", "sd", "num", "den", "sc", "sa", "sb"};
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
    const double u1x = p1x - p0x, u1y = p1y - p0y, v1x = p2x - p0x, v1y = p2y - p0y, u2x = q1x - q0x, u2y = q1y - q0y, v2x = q2x - q0x, v2y = q2y - q0y;
    const double denom = u1x * v1y - u1y * v1x; h[o++] = denom;
    const double n_aac = u1y * v2x - u2x * v1y; h[o++] = n_aac;
    const double ac = n_aac / denom; h[o++] = ac;
    const double ad = (u1y * v2y - u2y * v1y) / denom; h[o++] = ad;
    const double bc = (u2x * v1x - u1x * v2x) / denom; h[o++] = bc;
    const double bd = (u2y * v1x - u1x * v2y) / denom; h[o++] = bd;
    const double ac2 = ac * ac;
    const double g2 = -ac2 + ad * ad - bc * bc + bd * bd; h[o++] = g2;
    const double g1 = 2.0 * ac * ad + 2.0 * bc * bd; h[o++] = g1;
    const double g0 = ac2 + bc * bc - 1.0; h[o++] = g0;
    const double h4 = g1 * g1 + g2 * g2; h[o++] = h4;
    const double h2 = -g1 * g1 + 2.0 * g0 * g2; h[o++] = h2;
    const double h0 = g0 * g0; h[o++] = h0;
    const double rad = h2 * h2 - 4.0 * h4 * h0; h[o++] = rad;
    const double rdisc = std::sqrt(rad); h[o++] = rdisc;
    const double root = h2 + rdisc; h[o++] = root;
    const double t1 = -root / h4; h[o++] = t1;
    const double sd = std::sqrt(t1 / 2.0); h[o++] = sd;
    const double num = g2 * sd * sd + ac2 + bc * bc - 1.0; h[o++] = num;
    const double den = 2.0 * ac * ad * sd + 2.0 * bc * bd * sd; h[o++] = den;
    const double sc = -num / den; h[o++] = sc;
    h[o++] = ac * sc + ad * sd;
    h[o++] = bc * sc + bd * sd;
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
