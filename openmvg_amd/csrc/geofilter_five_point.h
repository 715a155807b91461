// geofilter_five_point.h - the five-point relative-pose solver of the essential-matrix model, one wave per sample (included by
// mvgx_geofilter.hip after its helpers: wave_sync, lane_value_f64, shfl_f64, wave_max_u32).
//
// Reference (paths under /root/reference/src/openMVG): multiview/solver_essential_five_point.cpp:170-230 FivePointsRelativePose =
//   1. FivePointsNullspaceBasis :35-42      four-dimensional null space of the 5 x 9 epipolar system (there: the eigenvectors of A^T A)
//   2. FivePointsPolynomialConstraints :115-168   det(E) = 0 and 2 E E^T E - trace(E E^T) E = 0 on E = x E1 + y E2 + z E3 + E4: ten cubic
//                                            polynomials, monomial order [xxx xxy xyy yyy xxz xyz yyz xzz yzz zzz | xx xy yy xz yz zz x y z 1]
//   3. Gauss-Jordan elimination of the 10 x 20 system on its cubic columns (there: FullPivLU of the left block, solve) :180-182
//   4. action matrix of the multiplication by x on [xx xy yy xz yz zz x y z 1] :187-193
//   5. real eigenvalues / eigenvectors of the 10 x 10 action matrix (there: Eigen::EigenSolver = Householder Hessenberg + Francis
//      double-shift QR), E = E_basis (x, y, z, 1) from the last four eigenvector components :195-212
// What runs where: 1 - elimination with complete pivoting over 45 lanes, then a Gram-Schmidt pass (the reference's basis is
// orthonormal: the conditioning of 2 - 5 depends on it); 2 - lane r builds row r of the constraint matrix; 3 - lane r owns row r,
// pivot rows travel through v_readlane; 5 - Householder Hessenberg reduction and the EISPACK hqr iteration on the wave's 10 x 10 matrix
// in LDS: the shift logic is wave-uniform scalar code, the row / column updates of every reflector are spread over the lanes; the ten
// candidate eigenvectors are then solved for at once, lane s taking eigenvalue s (the action matrix's rows 6 - 9 give four components in
// closed form, the other five are the least-squares solution of a 6 x 5 system by Householder QR - no lane crossing).
// Not bit-exact with Eigen (another null-space basis, another elimination order, no Schur vectors): an essential matrix agrees with
// the reference's to rounding; the parity policy of the F and H models applies (tests/_geofilter_cases.py).
#pragma once

// Contraction of a product and a sum into one fused operation by the SOURCE expression only (a * b + c written as one expression), not
// wherever the optimiser finds a product next to a sum: solve() and solve4() (geofilter_five_point_x4.h) must produce the same bits, and
// with the toolchain's default (-ffp-contract=fast) which products get fused depended on the code around them - an unrelated edit moved
// one sample in 64 by 7e-15 (call r5_29). Restored at the end of geofilter_five_point_x4.h.
#pragma clang fp contract(on)

namespace five_point {

#ifdef MVGX_FIVE_POINT_STAMPS   // measurement build: shader clocks of lane 0 per stage of solve(), summed over all calls
__device__ unsigned long long g_stamps[12];
#define FP_STAMP(i) do { const long long t_now = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&g_stamps[i], (unsigned long long)(t_now - t_prev)); t_prev = t_now; } while (0)
#else
#define FP_STAMP(i) do { } while (0)
#endif

constexpr int kN = 10;          // order of the action matrix
constexpr int kPolyScratch = 10 * 11 /* the Hyman polynomials x_i */ + 12 /* monic coefficients */ + 20 /* roots (re | im) */ + 10 /* start radii */;
constexpr int kScratch = 100 /* H */ + 10 /* v */ + 10 /* wr */ + 10 /* wi */ + 36 /* null-space basis */ + kPolyScratch;   // doubles of wave-private LDS
#ifndef MVGX_FIVE_POINT_ABERTH
#define MVGX_FIVE_POINT_ABERTH 1   // eigenvalues of the action matrix: characteristic polynomial + Ehrlich-Aberth (hqr stays as the fallback and, with 0, as the only route)
#endif

// 1 / x and 1 / sqrt(x) from v_rcp_f64 / v_rsq_f64 + two Newton steps (full precision for finite normal arguments): the iteration below
// spent 70 % of its clocks in IEEE division and square-root sequences (~150 clocks each, all on the critical path of a single wave)
__device__ __forceinline__ double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double frsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  r = fma(fma(-hx * r, r, 0.5), r, r);
  r = fma(fma(-hx * r, r, 0.5), r, r);
  return r;
}

// ---- polynomials in (x, y, z): degree 1 as {x, y, z, 1}, degree 2 as {xx, xy, yy, xz, yz, zz, x, y, z, 1} (the reference's columns
// 10..19), degree 3 in the reference's full order (solver_essential_five_point.hpp:103-125) ----
__device__ __forceinline__ void o1(const double (&a)[4], const double (&b)[4], double (&r)[10]) {   // :44-66
  r[0] = a[0] * b[0];
  r[1] = a[0] * b[1] + a[1] * b[0];
  r[2] = a[1] * b[1];
  r[3] = a[0] * b[2] + a[2] * b[0];
  r[4] = a[1] * b[2] + a[2] * b[1];
  r[5] = a[2] * b[2];
  r[6] = a[0] * b[3] + a[3] * b[0];
  r[7] = a[1] * b[3] + a[3] * b[1];
  r[8] = a[2] * b[3] + a[3] * b[2];
  r[9] = a[3] * b[3];
}
// r += a (degree 2) x b (degree 1)   (:68-113)
__device__ __forceinline__ void o2_add(const double (&a)[10], const double (&b)[4], double (&r)[20]) {
  const double axx = a[0], axy = a[1], ayy = a[2], axz = a[3], ayz = a[4], azz = a[5], ax = a[6], ay = a[7], az = a[8], a1 = a[9];
  const double bx = b[0], by = b[1], bz = b[2], b1 = b[3];
  r[0] += axx * bx;
  r[1] += axx * by + axy * bx;
  r[2] += axy * by + ayy * bx;
  r[3] += ayy * by;
  r[4] += axx * bz + axz * bx;
  r[5] += axy * bz + ayz * bx + axz * by;
  r[6] += ayy * bz + ayz * by;
  r[7] += axz * bz + azz * bx;
  r[8] += ayz * bz + azz * by;
  r[9] += azz * bz;
  r[10] += axx * b1 + ax * bx;
  r[11] += axy * b1 + ax * by + ay * bx;
  r[12] += ayy * b1 + ay * by;
  r[13] += axz * b1 + ax * bz + az * bx;
  r[14] += ayz * b1 + ay * bz + az * by;
  r[15] += azz * b1 + az * bz;
  r[16] += ax * b1 + a1 * bx;
  r[17] += ay * b1 + a1 * by;
  r[18] += az * b1 + a1 * bz;
  r[19] += a1 * b1;
}

// ---- 1. null space: basis[u][k] = component u (row-major 3 x 3 index) of basis vector k, the same in every lane ----
__device__ __forceinline__ void nullspace(const double* __restrict__ b1, const double* __restrict__ b2, const uint32_t (&s)[7], int lane, double (&basis)[9][4]) {
  // lane 9 r + c holds A[r][c] = x2[c / 3] x1[c % 3] of sample point r < 5 (EncodeEpipolarEquation, solver_fundamental_kernel.hpp:83-93)
  const int r = lane < 45 ? lane / 9 : 4, c = lane < 45 ? lane - 9 * (lane / 9) : 8;
  uint32_t si = s[0];
#pragma unroll
  for (int k = 1; k < 5; ++k) si = (r == k) ? s[k] : si;
  const int ci = c / 3, cj = c - 3 * ci;
  double a = b2[3 * (size_t)si + ci] * b1[3 * (size_t)si + cj];
  uint32_t row_used = 0, col_used = 0;
  int prow[5], pcol[5], n_piv = 0;
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    const bool cand = lane < 45 && !((row_used >> r) & 1u) && !((col_used >> c) & 1u);
    const float mag = (float)fabs(a);
    const uint32_t key = (cand && mag > 0.f && mag == mag) ? ((__float_as_uint(mag) & ~63u) | (uint32_t)(63 - lane)) : 0u;
    const uint32_t best = wave_max_u32(key);
    if (best == 0u) break;   // rank deficient sample (wave-uniform)
    const int who = 63 - (int)(best & 63u);
    const int pr = who / 9, pc = who - 9 * (who / 9);
    const double ipiv = 1.0 / lane_value_f64(a, who);
    const double rowv = shfl_f64(a, 9 * pr + c);   // pivot row, my column
    const double colv = shfl_f64(a, 9 * r + pc);   // my row, pivot column
    if (r != pr) a -= (colv * ipiv) * rowv;
    row_used |= 1u << pr; col_used |= 1u << pc;
    prow[step] = pr; pcol[step] = pc;
    n_piv = step + 1;
  }
  int fcol[4] = {8, 8, 8, 8}, nf = 0;
#pragma unroll
  for (int u = 0; u < 9; ++u)
    if (!((col_used >> u) & 1u) && nf < 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k == nf) fcol[k] = u;
      ++nf;
    }
  // lane u < 9: component u of the four vectors - 1 at the vector's free column, -A[prow][free] / A[prow][pcol] at a pivot column
  double bu[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bu[k] = lane == fcol[k] ? 1.0 : 0.0;
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    if (step < n_piv) {   // wave-uniform
      const double iden = 1.0 / lane_value_f64(a, 9 * prow[step] + pcol[step]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double nk = lane_value_f64(a, 9 * prow[step] + fcol[k]);
        if (lane == pcol[step]) bu[k] = -nk * iden;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 9; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) basis[u][k] = lane_value_f64(bu[k], u);
  // modified Gram-Schmidt (every lane the same arithmetic)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < k; ++j) {
      double d = 0.0;
#pragma unroll
      for (int u = 0; u < 9; ++u) d += basis[u][j] * basis[u][k];
#pragma unroll
      for (int u = 0; u < 9; ++u) basis[u][k] -= d * basis[u][j];
    }
    double n2 = 0.0;
#pragma unroll
    for (int u = 0; u < 9; ++u) n2 += basis[u][k] * basis[u][k];
    const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
    for (int u = 0; u < 9; ++u) basis[u][k] *= inv;
  }
}

// ---- 2. row `row` (< 10, this lane's) of the constraint matrix: 0 = det(E), 1 + 3 i + j = (2 E E^T E - trace(E E^T) E)(i, j) / 2.
// B = the null-space basis in LDS, B[4 u + k] = coefficient k of entry u of E (every polynomial is fetched where it is used: held in
// registers next to this function's intermediate polynomials, the basis pushed the kernel into scratch memory) ----
__device__ __forceinline__ void ld4(const double* __restrict__ B, int u, double (&e)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) e[t] = B[4 * u + t];
}
// `park` (solve4: 20 doubles of this lane's LDS, or nullptr): where the row's lane keeps the determinant row while the other form is
// built - both forms live in registers at once were 80 of the 256 a wave of the essential kernel has
__device__ __forceinline__ void constraint_row(const double* __restrict__ B, int row, double (&m)[20], double* __restrict__ park = nullptr) {
  const bool row0 = row == 0;
  // Both forms are evaluated by every lane and selected per element (a divergent branch around the determinant row kept the row in
  // scratch memory).
  double m0[20];   // the determinant row (:127-129)
#pragma unroll
  for (int t = 0; t < 20; ++t) { m[t] = 0.0; m0[t] = 0.0; }
  {
    const int trip[3][5] = {{1, 5, 2, 4, 6}, {2, 3, 0, 5, 7}, {0, 4, 1, 3, 8}};   // (E[a] E[b] - E[c] E[d]) E[e]
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      double ea[4], eb[4], p[10], q[10];
      ld4(B, trip[w][0], ea); ld4(B, trip[w][1], eb);
      o1(ea, eb, p);
      ld4(B, trip[w][2], ea); ld4(B, trip[w][3], eb);
      o1(ea, eb, q);
#pragma unroll
      for (int t = 0; t < 10; ++t) p[t] -= q[t];
      ld4(B, trip[w][4], ea);
      o2_add(p, ea, m0);
    }
  }
  if (park && row0) {
#pragma unroll
    for (int t = 0; t < 20; ++t) park[t] = m0[t];
  }
  if (row < 1) row = 1;   // (row 0 takes m0 below; its lane evaluates the (0, 0) form like lane 1)
  const int i = (row - 1) / 3, j = (row - 1) - 3 * i;
  // half the trace of E E^T (:147)
  double tr[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) tr[t] = 0.0;
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    double e[4], p[10];
    ld4(B, u, e);
    o1(e, e, p);
#pragma unroll
    for (int t = 0; t < 10; ++t) tr[t] += p[t];
  }
  // row i of E (this lane's i: a per-lane address)
  double ei[3][4];
#pragma unroll
  for (int k = 0; k < 3; ++k) ld4(B, 3 * i + k, ei[k]);
  // L[i][k] = (E E^T)(i, k) - (i == k) trace / 2, then sum_k L[i][k] E[k][j]   (:136-160)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double l[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) l[t] = 0.0;
#pragma unroll
    for (int mm = 0; mm < 3; ++mm) {
      double e[4], p[10];
      ld4(B, 3 * k + mm, e);
      o1(ei[mm], e, p);
#pragma unroll
      for (int t = 0; t < 10; ++t) l[t] += p[t];
    }
    if (i == k) {
#pragma unroll
      for (int t = 0; t < 10; ++t) l[t] -= 0.5 * tr[t];
    }
    double ej[4];
    ld4(B, 3 * k + j, ej);
    o2_add(l, ej, m);
  }
  if (row0) {
#pragma unroll
    for (int t = 0; t < 20; ++t) m[t] = park ? park[t] : m0[t];
  }
}

// ---- 3 + 4. lane r < 10 owns row r of [left | right]; Gauss-Jordan with complete pivoting on the left block; the action matrix goes to
// H (LDS, row-major 10 x 10). Returns false (wave-uniform) if the left block is singular to working precision. ----
__device__ __forceinline__ bool action_matrix(const double (&m_in)[20], int lane, double* __restrict__ H) {
  double m[20];   // (a private copy: worked on through the caller's array, the row stayed in scratch memory - hipcc, ROCm 7.2)
#pragma unroll
  for (int c = 0; c < 20; ++c) m[c] = m_in[c];
  uint32_t row_used = 0, col_used = 0;
  int my_pcol = -1;   // the pivot column of this lane's row
  const int r = lane < kN ? lane : kN - 1;
#pragma unroll
  for (int step = 0; step < kN; ++step) {   // (unrolled: rolled, the compiler kept the row in scratch memory - 5 000 clocks per step)
    uint32_t key = 0u;
    if (lane < kN && !((row_used >> r) & 1u)) {
#pragma unroll
      for (int c = 0; c < kN; ++c) {
        const float mag = (float)fabs(m[c]);
        const uint32_t k = (!((col_used >> c) & 1u) && mag > 0.f && mag == mag) ? ((__float_as_uint(mag) & ~127u) | (uint32_t)(127 - (kN * r + c))) : 0u;
        key = k > key ? k : key;
      }
    }
    const uint32_t best = wave_max_u32(key);
    if (best == 0u) return false;   // (wave-uniform)
    const int who = 127 - (int)(best & 127u);
    const int pr = who / kN, pc = who - kN * pr;
    double rowv[20];
#pragma unroll
    for (int c = 0; c < 20; ++c) rowv[c] = lane_value_f64(m[c], pr);
    double piv = rowv[0], colv = m[0];
#pragma unroll
    for (int c = 1; c < kN; ++c) { piv = (c == pc) ? rowv[c] : piv; colv = (c == pc) ? m[c] : colv; }
    const double f = colv * (1.0 / piv);
    if (lane < kN && r != pr) {
#pragma unroll
      for (int c = 0; c < 20; ++c) m[c] -= f * rowv[c];
    }
    if (lane == pr) my_pcol = pc;
    row_used |= 1u << pr; col_used |= 1u << pc;
  }
  // B[pcol] = right part of the row / its pivot; rows 0 1 2 4 5 7 of B are the rows 0..5 of the action matrix (:187-192)
  if (lane < kN) {
    double piv = m[0];
#pragma unroll
    for (int c = 1; c < kN; ++c) piv = (c == my_pcol) ? m[c] : piv;
    const double ip = 1.0 / piv;
    const int dst = my_pcol == 0 ? 0 : my_pcol == 1 ? 1 : my_pcol == 2 ? 2 : my_pcol == 4 ? 3 : my_pcol == 5 ? 4 : my_pcol == 7 ? 5 : -1;
    if (dst >= 0) {
#pragma unroll
      for (int c = 0; c < kN; ++c) H[dst * kN + c] = m[kN + c] * ip;
    }
  } else if (lane >= 16 && lane < 16 + 4 * kN) {   // rows 6..9: -1 at (6,0) (7,1) (8,3) (9,6)
    const int e = lane - 16, rr = 6 + e / kN, cc = e - kN * (e / kN);
    const int one = rr == 6 ? 0 : rr == 7 ? 1 : rr == 8 ? 3 : 6;
    H[rr * kN + cc] = cc == one ? -1.0 : 0.0;
  }
  wave_sync();
  return true;
}

// ---- 5a. Householder reduction of H (LDS) to upper Hessenberg form; v = 10 doubles of LDS ----
__device__ __forceinline__ void hessenberg(double* __restrict__ H, double* __restrict__ v, int lane) {
#pragma unroll
  for (int k = 0; k < kN - 2; ++k) {
    double s = 0.0;
    for (int i = k + 2; i < kN; ++i) { const double t = H[i * kN + k]; s += t * t; }   // (wave-uniform reads)
    if (s == 0.0) continue;
    const double x0 = H[(k + 1) * kN + k];
    const double norm = sqrt(x0 * x0 + s);
    const double v0 = x0 + (x0 >= 0.0 ? norm : -norm);
    const double tau = 1.0 / (norm * fabs(v0));   // 2 / (v^T v)
    wave_sync();
    if (lane > k && lane < kN) v[lane] = lane == k + 1 ? v0 : H[lane * kN + k];
    wave_sync();
    if (lane >= k && lane < kN) {   // (I - tau v v^T) H: column `lane`
      double d = 0.0;
      for (int i = k + 1; i < kN; ++i) d += v[i] * H[i * kN + lane];
      d *= tau;
      for (int i = k + 1; i < kN; ++i) H[i * kN + lane] -= d * v[i];
    }
    wave_sync();
    if (lane < kN) {   // H (I - tau v v^T): row `lane`
      double d = 0.0;
      for (int j = k + 1; j < kN; ++j) d += H[lane * kN + j] * v[j];
      d *= tau;
      for (int j = k + 1; j < kN; ++j) H[lane * kN + j] -= d * v[j];
    }
    wave_sync();
    if (lane > k + 1 && lane < kN) H[lane * kN + k] = 0.0;
    wave_sync();
  }
}

// ---- 5b. eigenvalues of the upper Hessenberg H (destroyed): the EISPACK hqr iteration (Francis double shift, the exceptional shifts at
// iterations 10 and 20, 30 iterations per eigenvalue at most). The shift logic is the same scalar code in every lane (reads of LDS at
// wave-uniform addresses); a reflector's row update runs on lanes k..nn (one column each), its column update on lanes l..min(nn, k + 3).
// Returns false if an eigenvalue did not converge. ----
__device__ __forceinline__ bool hqr(double* __restrict__ a, double* __restrict__ wr, double* __restrict__ wi, int lane) {
  auto A = [&](int i, int j) -> double& { return a[i * kN + j]; };
  double anorm = 0.0;
  for (int i = 0; i < kN; ++i)
    for (int j = (i > 0 ? i - 1 : 0); j < kN; ++j) anorm += fabs(A(i, j));
  int nn = kN - 1;
  double t = 0.0;
  while (nn >= 0) {
    int its = 0, l;
    do {
      for (l = nn; l >= 1; --l) {
        double s = fabs(A(l - 1, l - 1)) + fabs(A(l, l));
        if (s == 0.0) s = anorm;
        if (fabs(A(l, l - 1)) + s == s) {
          wave_sync();
          if (lane == 0) A(l, l - 1) = 0.0;
          wave_sync();
          break;
        }
      }
      double x = A(nn, nn);
      if (l == nn) {   // one root
        wave_sync();
        if (lane == 0) { wr[nn] = x + t; wi[nn] = 0.0; }
        --nn;
      } else {
        double y = A(nn - 1, nn - 1), w = A(nn, nn - 1) * A(nn - 1, nn);
        if (l == nn - 1) {   // two roots
          const double p = 0.5 * (y - x), q = p * p + w;
          double z = sqrt(fabs(q));
          x += t;
          wave_sync();
          if (q >= 0.0) {
            z = p + (p >= 0.0 ? fabs(z) : -fabs(z));
            if (lane == 0) { wr[nn - 1] = wr[nn] = x + z; if (z != 0.0) wr[nn] = x - w / z; wi[nn - 1] = wi[nn] = 0.0; }
          } else if (lane == 0) {
            wr[nn - 1] = wr[nn] = x + p; wi[nn - 1] = z; wi[nn] = -z;
          }
          nn -= 2;
        } else {
          if (its == 30) return false;
          if (its == 10 || its == 20) {   // exceptional shift
            t += x;
            wave_sync();
            if (lane <= nn) A(lane, lane) -= x;
            wave_sync();
            const double s = fabs(A(nn, nn - 1)) + fabs(A(nn - 1, nn - 2));
            y = x = 0.75 * s;
            w = -0.4375 * s * s;
          }
          ++its;
          int m;
          double p = 0.0, q = 0.0, r = 0.0, z;
          for (m = nn - 2; m >= l; --m) {
            z = A(m, m);
            r = x - z;
            double s = y - z;
            p = (r * s - w) * frcp(A(m + 1, m)) + A(m, m + 1);
            q = A(m + 1, m + 1) - z - r - s;
            r = A(m + 2, m + 1);
            s = fabs(p) + fabs(q) + fabs(r);
            { const double is = s != 0.0 ? frcp(s) : 0.0; p *= is; q *= is; r *= is; }
            if (m == l) break;
            const double u = fabs(A(m, m - 1)) * (fabs(q) + fabs(r));
            const double vv = fabs(p) * (fabs(A(m - 1, m - 1)) + fabs(z) + fabs(A(m + 1, m + 1)));
            if (u + vv == vv) break;
          }
          wave_sync();
          if (lane >= m + 2 && lane <= nn) {
            A(lane, lane - 2) = 0.0;
            if (lane != m + 2) A(lane, lane - 3) = 0.0;
          }
          wave_sync();
          for (int k = m; k <= nn - 1; ++k) {
            if (k != m) {
              p = A(k, k - 1); q = A(k + 1, k - 1); r = 0.0;
              if (k != nn - 1) r = A(k + 2, k - 1);
              if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) { const double ix = frcp(x); p *= ix; q *= ix; r *= ix; }
            }
            const double n2 = p * p + q * q + r * r;
            const double irs = n2 > 0.0 ? frsqrt(n2) : 0.0;   // 1 / |s|
            const double sq = n2 * irs;
            const double s = p >= 0.0 ? sq : -sq;
            if (s != 0.0) {
              wave_sync();
              if (lane == 0) {
                if (k == m) { if (l != m) A(k, k - 1) = -A(k, k - 1); }
                else A(k, k - 1) = -s * x;
              }
              { const double is = p >= 0.0 ? irs : -irs;   // 1 / s
                p += s; x = p * is; y = q * is; z = r * is;
                const double ip = frcp(p); q *= ip; r *= ip; }
              wave_sync();
              if (lane >= k && lane <= nn) {   // row modification, column `lane`
                double pp = A(k, lane) + q * A(k + 1, lane);
                if (k != nn - 1) { pp += r * A(k + 2, lane); A(k + 2, lane) -= pp * z; }
                A(k + 1, lane) -= pp * y;
                A(k, lane) -= pp * x;
              }
              wave_sync();
              const int mmin = nn < k + 3 ? nn : k + 3;
              if (lane >= l && lane <= mmin) {   // column modification, row `lane`
                double pp = x * A(lane, k) + y * A(lane, k + 1);
                if (k != nn - 1) { pp += z * A(lane, k + 2); A(lane, k + 2) -= pp * r; }
                A(lane, k + 1) -= pp * q;
                A(lane, k) -= pp;
              }
              wave_sync();
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  wave_sync();
  return true;
}

// ---- 5c. M(lam) = [C(lam) | d(lam)], the 6 x 6 polynomial matrix of the action matrix's rows 0..5 after its rows 6..9 have been used
// in closed form (v0 = lam^2 v9, v1 = -lam v7, v3 = -lam v8, v6 = -lam v9): columns = the unknowns (v2, v4, v5, v7, v8) and the right-hand
// side for v9 = 1. det M(lam) is the characteristic polynomial of the action matrix (degree 10, same roots); at a root the
// least-squares solution of C u = d gives the eigenvector. One routine serves both: five Householder reflections in registers (fixed
// control flow, no pivoting, backward stable), leaving R and Q^T d in C. `At`: LDS, rows 0..5 of the action matrix (row stride kN). ----
// The matrix is used in its HOMOGENEOUS form M_h(alpha, beta), lam = beta / alpha with alpha^2 + beta^2 = 1: column k of M scaled by
// alpha^(degree of the column) - entries bounded by |A| for every lam, det M_h = alpha^10 det M (same sign, same roots). In lam itself
// the columns carry lam^2 and lam^3 and the determinant's rounding error moved roots of |lam| ~ 10 by 1e-12 (1e-7 in the essential matrix).
__device__ __forceinline__ void qr_rows(const double* __restrict__ At, double al, double be, double (&C)[6][6]) {
  const double a2 = al * al, ab = al * be, b2 = be * be;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const double* __restrict__ a = At + r * kN;
    C[r][0] = a[2] * al - (r == 2 ? be : 0.0);
    C[r][1] = a[4] * al - (r == 4 ? be : 0.0);
    C[r][2] = a[5] * al - (r == 5 ? be : 0.0);
    C[r][3] = a[7] * a2 - a[1] * ab + (r == 1 ? b2 : 0.0);
    C[r][4] = a[8] * a2 - a[3] * ab + (r == 3 ? b2 : 0.0);
    C[r][5] = -(a[0] * (al * b2) - a[6] * (a2 * be) + a[9] * (a2 * al)) + (r == 0 ? b2 * be : 0.0);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {   // Householder reflection of column k, applied to the columns behind it and to the right-hand side
    double s = 0.0;
#pragma unroll
    for (int i = k + 1; i < 6; ++i) s += C[i][k] * C[i][k];
    const double x0 = C[k][k];
    const double n2 = x0 * x0 + s;
    if (n2 > 0.0) {
      const double norm = n2 * frsqrt(n2);
      const double v0 = x0 + (x0 >= 0.0 ? norm : -norm);
      const double tau = frcp(norm * fabs(v0));
#pragma unroll
      for (int j = k + 1; j < 6; ++j) {
        double d = v0 * C[k][j];
#pragma unroll
        for (int i = k + 1; i < 6; ++i) d += C[i][k] * C[i][j];
        d *= tau;
        C[k][j] -= d * v0;
#pragma unroll
        for (int i = k + 1; i < 6; ++i) C[i][j] -= d * C[i][k];
      }
      C[k][k] = x0 >= 0.0 ? -norm : norm;
    }
  }
}
// tail = (v6, v7, v8, v9) ~ (x, y, z, 1) of the eigenvector for the real eigenvalue lam = beta / alpha (any common scale: an essential
// matrix is defined up to scale): the solution w of C_h w = d_h gives (-beta, w3, w4, alpha)
__device__ __forceinline__ void eigenvector_tail(const double* __restrict__ At, double lam, double (&tail)[4]) {
  const double ih = frsqrt(1.0 + lam * lam);
  const double al = ih, be = lam * ih;
  double C[6][6];
  qr_rows(At, al, be, C);
  double u[5];
#pragma unroll
  for (int k = 4; k >= 0; --k) {   // back-substitution R u = Q^T d
    double t = C[k][5];
#pragma unroll
    for (int j = k + 1; j < 5; ++j) t -= C[k][j] * u[j];
    u[k] = t / C[k][k];
  }
  tail[0] = -be; tail[1] = u[3]; tail[2] = u[4]; tail[3] = al;
}

// ---- 5e. the eigenvalues without a QR iteration (round 4) ----
// hqr on ONE 10 x 10 matrix keeps a wave busy for ~209 k clocks of mostly scalar, dependent work (70 % of a five-point solve).
// Here instead, from the Hessenberg form H (same eigenvalues as the action matrix):
//   1. the characteristic polynomial by Hyman's recurrence in polynomial arithmetic: x_9 = 1,
//      x_{i-1}(z) = ((z - h_ii) x_i(z) - sum_{j > i} h_ij x_j(z)) / h_{i,i-1},  p(z) ~ (z - h_00) x_0(z) - sum_{j >= 1} h_0j x_j(z);
//      lane k carries coefficient k, ten steps of at most ten multiply-adds;
//   2. all ten roots at once by the Ehrlich-Aberth iteration on the monic coefficients (lane i = root i, complex arithmetic, start
//      radii from the Newton polygon of the coefficient moduli as in Bini's algorithm): cubic convergence, ~10 rounds of ~250
//      instructions;
//   3. the roots with a negligible imaginary part are polished ON THE MATRIX - Newton steps with p and p' from Hyman's recurrence
//      evaluated on H itself (backward stable), so what the coefficients lost does not reach the eigenvalue - and must converge as
//      real roots to count as real (Eigen's criterion is imag() == 0 of a RealSchur form: a 2 x 2 block with non-negative
//      discriminant; the two agree unless the discriminant is within rounding of zero).
// Anything unusual - a vanishing subdiagonal, a non-finite value, no convergence in kAberthIter rounds, two polished roots that
// coincide - returns false and the caller runs hqr on the same H.
constexpr int kAberthIter = 48;
__device__ unsigned long long g_hqr_fallbacks;   // solves whose eigenvalues came from hqr after all (diagnostic: mvgx_debug_five_point_fallbacks)
#ifdef MVGX_FIVE_POINT_COUNT_ROUNDS
__device__ unsigned long long g_aberth_rounds, g_aberth_solves;
#endif
__device__ __forceinline__ bool finite_d(double x) { return fabs(x) < 1.0e300; }   // (false for NaN as well)

// p(x) and p'(x) up to the same constant factor, from the Hessenberg matrix in LDS (every lane may pass its own x): Hyman's
// recurrence and its derivative, backward stable (a difference quotient of p instead of p' lost one sample in 20 000 to cancellation)
__device__ __forceinline__ void hyman(const double* __restrict__ H, const double* __restrict__ rsub /* 1 / h_{i,i-1}, i = 1..9 */, double x,
                                      double& q, double& dq) {
  double y[kN], dy[kN];
  y[kN - 1] = 1.0; dy[kN - 1] = 0.0;
#pragma unroll
  for (int i = kN - 1; i >= 1; --i) {
    double t = (x - H[i * kN + i]) * y[i], dt = y[i] + (x - H[i * kN + i]) * dy[i];
#pragma unroll
    for (int j = i + 1; j < kN; ++j) { t -= H[i * kN + j] * y[j]; dt -= H[i * kN + j] * dy[j]; }
    y[i - 1] = t * rsub[i]; dy[i - 1] = dt * rsub[i];
  }
  double t = (x - H[0]) * y[0], dt = y[0] + (x - H[0]) * dy[0];
#pragma unroll
  for (int j = 1; j < kN; ++j) { t -= H[j] * y[j]; dt -= H[j] * dy[j]; }
  q = t; dq = dt;
}

__device__ __forceinline__ bool eigenvalues_aberth(const double* __restrict__ H, double* __restrict__ wr, double* __restrict__ wi,
                                                   double* __restrict__ ps /* kPolyScratch */, double* __restrict__ rsub /* 10: v */, int lane) {
  double* const X = ps;              // [i][k]: coefficient k of x_i
  double* const coef = ps + 110;     // monic: coef[k], k = 0..10
  double* const zre = coef + 12;
  double* const zim = zre + 10;
  double* const rad = zim + 10;
  // ---- 1. characteristic polynomial ----
  bool ok = true;
  if (lane >= 1 && lane < kN) {
    const double h = H[lane * kN + lane - 1];
    ok = h != 0.0 && finite_d(h);
    rsub[lane] = ok ? 1.0 / h : 0.0;
  }
  if (__ballot(!ok)) return false;
  if (lane <= kN) X[(kN - 1) * 11 + lane] = lane == 0 ? 1.0 : 0.0;
  wave_sync();
#pragma unroll
  for (int i = kN - 1; i >= 1; --i) {   // (uniform; unrolled: the reads of a step are then in flight together)
    if (lane <= kN) {
      double t = (lane > 0 ? X[i * 11 + lane - 1] : 0.0) - H[i * kN + i] * X[i * 11 + lane];
#pragma unroll
      for (int j = i + 1; j < kN; ++j) t -= H[i * kN + j] * X[j * 11 + lane];
      X[(i - 1) * 11 + lane] = t * rsub[i];
    }
    wave_sync();
  }
  double mine = 0.0;
  if (lane <= kN) {
    double t = (lane > 0 ? X[lane - 1] : 0.0) - H[0] * X[lane];
#pragma unroll
    for (int j = 1; j < kN; ++j) t -= H[j] * X[j * 11 + lane];
    mine = t;
  }
  const double lead = __shfl(mine, kN);
  if (!(lead != 0.0) || !finite_d(lead)) return false;
  mine = mine / lead;
  if (__ballot(lane <= kN && !finite_d(mine))) return false;
  if (lane <= kN) coef[lane] = mine;
  wave_sync();
  // ---- 2a. start radii: Newton polygon of (k, log |a_k|) - its upper hull; an edge from k1 to k2 carries k2 - k1 roots of modulus
  // (|a_k1| / |a_k2|)^(1 / (k2 - k1)). One lane walks the eleven points (gift wrapping from k = 0). ----
  {
    // lane k2 holds log |a_k2| and computes its slope from the current hull vertex k1; the steepest slope, farthest point on ties,
    // is the next vertex (wave maximum of a key), its k2 - k1 roots get the modulus exp(-slope): one step per hull edge
    const double la = (lane <= kN && fabs(mine) > 0.0) ? log(fabs(mine)) : -1.0e300;
    int k1 = 0;
#pragma unroll 1
    while (k1 < kN) {   // (uniform)
      const double lk1 = shfl_f64(la, k1);
      const bool cand = lane > k1 && lane <= kN;
      const double sl = cand ? (la - lk1) * frcp((double)(lane - k1)) : -1.0e308;
      // order-preserving key of the slope (doubles of one sign compare like their bit patterns; negative ones reversed), low 4 bits: k2
      unsigned long long bits = (unsigned long long)__double_as_longlong(sl);
      bits = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
      const uint32_t hi = (uint32_t)(bits >> 32);
      const uint32_t best_hi = wave_max_u32(cand ? hi : 0u);
      const uint32_t lo = (cand && hi == best_hi) ? (((uint32_t)bits & ~15u) | (uint32_t)lane) : 0u;
      const uint32_t best_lo = wave_max_u32(lo);
      const int best = (int)(best_lo & 15u);
      if (best <= k1) break;   // (cannot happen: lane kN is always a candidate)
      const double slope = shfl_f64(sl, best);
      const double r = exp(-slope);
      if (lane >= k1 && lane < best) rad[lane] = r;
      k1 = best;
    }
  }
  wave_sync();
  // ---- 2b. Ehrlich-Aberth ----
  double zr = 0.0, zi = 0.0;
  if (lane < kN) {
    double r = rad[lane];
    if (!(r > 1.0e-150)) r = 1.0e-150;
    if (!(r < 1.0e150)) r = 1.0e150;
    const double ang = 0.62831853071795865 * (double)lane + 0.7;   // 2 pi / 10 apart, off the axes
    zr = r * cos(ang); zi = r * sin(ang);
    zre[lane] = zr; zim[lane] = zi;
  }
  wave_sync();
  bool done = lane >= kN;
  int it = 0;
  for (; it < kAberthIter; ++it) {   // (uniform)
    double wr_ = 0.0, wi_ = 0.0;
    if (lane < kN && !done) {
      // p and p' by Horner on the monic coefficients (real) at the complex point
      double pr = 1.0, pi = 0.0, dr = 0.0, di = 0.0;
      double a[kN];   // (the coefficients first: a load per step in front of its multiply-adds was a chain of LDS latencies)
#pragma unroll
      for (int k = 0; k < kN; ++k) a[k] = coef[k];
#pragma unroll
      for (int k = kN - 1; k >= 0; --k) {
        const double ndr = dr * zr - di * zi + pr, ndi = dr * zi + di * zr + pi;
        const double npr = pr * zr - pi * zi + a[k], npi = pr * zi + pi * zr;
        dr = ndr; di = ndi; pr = npr; pi = npi;
      }
      // Newton correction N = p / p'
      const double dn = dr * dr + di * di;
      double nr = 0.0, ni = 0.0;
      if (dn > 0.0 && finite_d(dn)) { const double idn = frcp(dn); nr = (pr * dr + pi * di) * idn; ni = (pi * dr - pr * di) * idn; }
      // S = sum_{j != i} 1 / (z_i - z_j)
      double sr = 0.0, si = 0.0;
      {
        double ar[kN], ai[kN], ia[kN];   // (all ten reciprocals in flight: one at a time the loop was a chain of LDS and v_rcp latencies)
#pragma unroll
        for (int j = 0; j < kN; ++j) { ar[j] = zr - zre[j]; ai[j] = zi - zim[j]; }
#pragma unroll
        for (int j = 0; j < kN; ++j) { const double an = ar[j] * ar[j] + ai[j] * ai[j]; ia[j] = (j != lane && an > 0.0) ? frcp(an) : 0.0; }
#pragma unroll
        for (int j = 0; j < kN; ++j) { sr += ar[j] * ia[j]; si -= ai[j] * ia[j]; }
      }
      // w = N / (1 - N S)
      const double er = 1.0 - (nr * sr - ni * si), ei = -(nr * si + ni * sr);
      const double en = er * er + ei * ei;
      if (en > 0.0 && finite_d(en)) { const double ie = frcp(en); wr_ = (nr * er + ni * ei) * ie; wi_ = (ni * er - nr * ei) * ie; }
      else { wr_ = nr; wi_ = ni; }
    }
    wave_sync();   // every lane has read the roots of this round
    if (lane < kN && !done) {
      zr -= wr_; zi -= wi_;
      zre[lane] = zr; zim[lane] = zi;
      const double zz = zr * zr + zi * zi, ww = wr_ * wr_ + wi_ * wi_;
      done = ww <= 1.0e-24 * zz || (zz == 0.0 && ww == 0.0);   // |w| <= 1e-12 |z|: the real ones are polished on the matrix below
      if (!finite_d(zz)) done = false;
    }
    wave_sync();
    if (!__ballot(!done)) break;
  }
#ifdef MVGX_FIVE_POINT_COUNT_ROUNDS
  if (lane == 0) { atomicAdd(&g_aberth_rounds, (unsigned long long)(it + 1)); atomicAdd(&g_aberth_solves, 1ull); }
#endif
  if (__ballot(!done)) return false;   // no convergence (or a non-finite iterate): hqr decides
  // ---- 3. real roots: polished on the matrix ----
  bool real = false;
  double x = zr;
  if (lane < kN) {
    const double az = sqrt(zr * zr + zi * zi);
    real = fabs(zi) <= 1.0e-6 * az || az == 0.0;
  }
  bool bad = false;
  if (real) {
    double step = 0.0;
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {   // (from the 1e-12 the iteration above stops at - or whatever the coefficients cost - to rounding)
      double q, dq;
      hyman(H, rsub, x, q, dq);
      step = dq != 0.0 ? q * frcp(dq) : 0.0;
      x -= step;
    }
    // a root that is real converges quadratically from here; one that is not (a conjugate pair close to the axis) does not
    const bool converged = fabs(step) <= 1.0e-9 * fabs(x) + 1.0e-300 && finite_d(x);
    if (!converged) { if (fabs(zi) <= 1.0e-12 * fabs(zr)) bad = true; real = false; }   // (an essentially real root that does not polish: let hqr decide)
  }
  if (__ballot(bad)) return false;
  // two polished roots on the same value: a double root or a pair drawn together - hqr decides
  if (lane < kN) { zre[lane] = real ? x : 0.0; zim[lane] = real ? 0.0 : 1.0; }
  wave_sync();
  bool dup = false;
  if (real) {
#pragma unroll
    for (int j = 0; j < kN; ++j)
      if (j != lane && zim[j] == 0.0 && fabs(zre[j] - x) <= 1.0e-10 * fabs(x)) dup = true;
  }
  if (__ballot(dup)) return false;
  if (lane < kN) { wr[lane] = real ? x : zr; wi[lane] = real ? 0.0 : (zi != 0.0 ? zi : 1.0); }
  wave_sync();
  return true;
}

// FivePointSolver::Solve on the sample s[0..4] (wave-uniform). scr: kScratch doubles of wave-private LDS. The essential matrices of
// the real solutions (row-major 3 x 3) go to Es[model][9] (LDS, 10 x 9 doubles); returns their number (wave-uniform).
__device__ __forceinline__ int solve(const double* __restrict__ b1, const double* __restrict__ b2, const uint32_t (&s)[7], int lane,
                                     double* __restrict__ scr, double* __restrict__ Es) {
  double* const H = scr;
  double* const v = scr + 100;
  double* const wr = v + 10;
  double* const wi = wr + 10;
  double* const basis_lds = wi + 10;
  double basis[9][4];
#ifdef MVGX_FIVE_POINT_STAMPS
  long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  nullspace(b1, b2, s, lane, basis);
  FP_STAMP(0);
  double At_row[kN];   // this lane's row of the action matrix (lanes < 10): kept for the eigenvectors, H is destroyed by the iteration
  {
    // the basis lives in LDS from here on (it is wave-uniform): its 72 registers are what the expansion, the elimination and the
    // iteration were spilling
    if (lane < 36) {
      double mine = 0.0;
#pragma unroll
      for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) mine = lane == 4 * u + k ? basis[u][k] : mine;
      basis_lds[lane] = mine;
    }
    wave_sync();
    double m[20];
    constraint_row(basis_lds, lane < kN ? lane : kN - 1, m);
    wave_sync();
    FP_STAMP(1);
    if (!action_matrix(m, lane, H)) return 0;
    FP_STAMP(2);
  }
#pragma unroll
  for (int c = 0; c < kN; ++c) At_row[c] = H[(lane < kN ? lane : 0) * kN + c];
  hessenberg(H, v, lane);
  FP_STAMP(3);
#if MVGX_FIVE_POINT_ABERTH
  if (!eigenvalues_aberth(H, wr, wi, basis_lds + 36, v, lane)) {   // (reads H only: the fallback starts from the same matrix)
    if (lane == 0) atomicAdd(&g_hqr_fallbacks, 1ull);
    if (!hqr(H, wr, wi, lane)) return 0;
  }
#else
  if (!hqr(H, wr, wi, lane)) return 0;
#endif
  FP_STAMP(4);
  // the action matrix again (the iteration worked in place), then one eigenvector per lane
#pragma unroll
  for (int c = 0; c < kN; ++c)
    if (lane < kN) H[lane * kN + c] = At_row[c];
  wave_sync();
  const double lam = wr[lane < kN ? lane : 0];
  const bool real = lane < kN && wi[lane < kN ? lane : 0] == 0.0 && lam == lam && fabs(lam) < 1.0e150;
  double tail[4] = {0.0, 0.0, 0.0, 0.0};
  if (real) eigenvector_tail(H, lam, tail);
  const unsigned long long real_mask = __ballot(real);
  const int n = (int)__popcll(real_mask);
  const int slot = (int)__popcll(real_mask & ((1ull << lane) - 1ull));
  if (real) {
#pragma unroll
    for (int u = 0; u < 9; ++u)
      Es[slot * 9 + u] = basis_lds[4 * u] * tail[0] + basis_lds[4 * u + 1] * tail[1] + basis_lds[4 * u + 2] * tail[2] + basis_lds[4 * u + 3] * tail[3];
  }
  wave_sync();
  FP_STAMP(5);
  return n;
}

}  // namespace five_point
