// geofilter_five_point_x4.h - the five-point solver of geofilter_five_point.h on FOUR samples at once, one per 16-lane row of the wave
// (included by mvgx_geofilter.hip right after geofilter_five_point.h).
//
// Why: solve() keeps a whole wave busy with the dependent instruction stream of ONE sample - ten or eleven lanes carry the rows of the
// constraint matrix, the coefficients of the characteristic polynomial, the ten roots; the other fifty run along. 85 % of an
// a-contrario iteration of the essential model is that stream (profiles/round4_geofilter_stage_clocks_call_r4_41.txt). The a-contrario
// loop draws its samples from a generator whose state does not depend on the models (only a better model or the change of the sampling
// mode re-seats the pool), so the kernel draws four samples ahead, solves them here side by side and evaluates them in order
// (mvgx_geofilter.hip: "samples ahead").
//
// Every ELEMENT sees the arithmetic of solve() - same operations, same order, same pivot choices (keys and tie-breaks), same
// convergence tests per sample - so the essential matrices are the same bits (tests/test_geofilter_e.py: solve4 == 4 x solve on the
// device and under the emulation). What changes is where an element lives and how it travels:
//   * a row of 16 lanes per sample: wave-uniform decisions become row-uniform (everything is predicated - no wave-collective
//     operation sits in divergent code), v_readlane becomes an exchange through the LDS crossbar, the wave maximum a DPP row maximum;
//   * the 5 x 9 elimination of the null space: lane c < 9 holds COLUMN c (five registers) instead of one element per lane of 45;
//   * a sample that fails a stage (rank-deficient elimination, Ehrlich-Aberth without convergence ...) carries a flag instead of
//     leaving; the rare hqr fall-back then runs wave-wide on that row's matrix, one row after the other - the same code path as solve().
#pragma once

namespace five_point {

__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {   // maximum over the 16 lanes of a DPP row, in every lane of the row
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // lane ^ 1
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // lane ^ 2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // the other quad of the half row
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // the other half of the row
  return v;
}
// value of lane `src` (0..15, the same in the 16 lanes of a row) of this lane's row
__device__ __forceinline__ double row_value_f64(double v, int src, int lane) { return __shfl(v, (lane & 48) | (src & 15)); }
__device__ __forceinline__ uint32_t row_ballot(bool p, int lane) { return (uint32_t)(__ballot(p) >> (lane & 48)) & 0xffffu; }

// ---- 1. null space (nullspace() of geofilter_five_point.h): the orthonormal basis goes to B (this row's 36 doubles of LDS, B[4 u + k]) ----
__device__ __forceinline__ void nullspace4(const double* __restrict__ b1, const double* __restrict__ b2, const uint32_t (&s)[5], int lane, double* __restrict__ B) {
  const int gl = lane & 15;
  const bool live = gl < 9;
  const int c = live ? gl : 8;
  const int ci = c / 3, cj = c - 3 * ci;
  double a[5];   // column c of the 5 x 9 system: a[r] = x2[c / 3] x1[c % 3] of sample point r
#pragma unroll
  for (int r = 0; r < 5; ++r) a[r] = b2[3 * (size_t)s[r] + ci] * b1[3 * (size_t)s[r] + cj];
  uint32_t row_used = 0, col_used = 0;
  int prow[5] = {0, 0, 0, 0, 0}, pcol[5] = {0, 0, 0, 0, 0}, n_piv = 0;
  bool going = true;   // (row-uniform) false from the step that finds no pivot: a rank-deficient sample
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    uint32_t key = 0u;
    if (live && !((col_used >> c) & 1u)) {
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const float mag = (float)fabs(a[r]);
        const uint32_t k = (!((row_used >> r) & 1u) && mag > 0.f && mag == mag) ? ((__float_as_uint(mag) & ~63u) | (uint32_t)(63 - (9 * r + c))) : 0u;
        key = k > key ? k : key;
      }
    }
    const uint32_t best = row_max_u32(key);
    going = going && best != 0u;
    const int who = going ? 63 - (int)(best & 63u) : 0;
    const int pr = who / 9, pc = who - 9 * (who / 9);
    double sel = a[0];   // this column's element of the pivot row
#pragma unroll
    for (int r = 1; r < 5; ++r) sel = (pr == r) ? a[r] : sel;
    const double ipiv = 1.0 / row_value_f64(sel, pc, lane);
    const double rowv = sel;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const double colv = row_value_f64(a[r], pc, lane);   // row r, pivot column
      if (going && r != pr) a[r] -= (colv * ipiv) * rowv;
    }
    if (going) {
      row_used |= 1u << pr; col_used |= 1u << pc;
      prow[step] = pr; pcol[step] = pc;
      n_piv = step + 1;
    }
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
  for (int k = 0; k < 4; ++k) bu[k] = gl == fcol[k] ? 1.0 : 0.0;
#pragma unroll
  for (int step = 0; step < 5; ++step) {
    double sel = a[0];
#pragma unroll
    for (int r = 1; r < 5; ++r) sel = (prow[step] == r) ? a[r] : sel;
    const double iden = 1.0 / row_value_f64(sel, pcol[step], lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double nk = row_value_f64(sel, fcol[k], lane);
      if (step < n_piv && gl == pcol[step]) bu[k] = -nk * iden;
    }
  }
  wave_sync();
  if (gl < 9) {
#pragma unroll
    for (int k = 0; k < 4; ++k) B[4 * gl + k] = bu[k];
  }
  wave_sync();
  double basis[9][4];
#pragma unroll
  for (int u = 0; u < 9; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) basis[u][k] = B[4 * u + k];
  wave_sync();
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
  if (gl == 0) {
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) B[4 * u + k] = basis[u][k];
  }
  wave_sync();
}

// ---- 3 + 4. action_matrix(): lane gl < 10 of the row owns row gl of [left | right]. Returns false (row-uniform) for a singular left block. ----
__device__ __forceinline__ bool action_matrix4(const double (&m_in)[20], int lane, double* __restrict__ H) {
  const int gl = lane & 15;
  double m[20];
#pragma unroll
  for (int c = 0; c < 20; ++c) m[c] = m_in[c];
  uint32_t row_used = 0, col_used = 0;
  int my_pcol = -1;
  const int r = gl < kN ? gl : kN - 1;
  bool going = true;
#pragma unroll
  for (int step = 0; step < kN; ++step) {
    uint32_t key = 0u;
    if (gl < kN && !((row_used >> r) & 1u)) {
#pragma unroll
      for (int c = 0; c < kN; ++c) {
        const float mag = (float)fabs(m[c]);
        const uint32_t k = (!((col_used >> c) & 1u) && mag > 0.f && mag == mag) ? ((__float_as_uint(mag) & ~127u) | (uint32_t)(127 - (kN * r + c))) : 0u;
        key = k > key ? k : key;
      }
    }
    const uint32_t best = row_max_u32(key);
    going = going && best != 0u;
    const int who = going ? 127 - (int)(best & 127u) : 0;
    const int pr = who / kN, pc = who - kN * pr;
    double rowv[20];
#pragma unroll
    for (int c = 0; c < 20; ++c) rowv[c] = row_value_f64(m[c], pr, lane);
    double piv = rowv[0], colv = m[0];
#pragma unroll
    for (int c = 1; c < kN; ++c) { piv = (c == pc) ? rowv[c] : piv; colv = (c == pc) ? m[c] : colv; }
    const double f = colv * (1.0 / piv);
    if (going && gl < kN && r != pr) {
#pragma unroll
      for (int c = 0; c < 20; ++c) m[c] -= f * rowv[c];
    }
    if (going) {
      if (gl == pr) my_pcol = pc;
      row_used |= 1u << pr; col_used |= 1u << pc;
    }
  }
  if (going && gl < kN) {
    double piv = m[0];
#pragma unroll
    for (int c = 1; c < kN; ++c) piv = (c == my_pcol) ? m[c] : piv;
    const double ip = 1.0 / piv;
    const int dst = my_pcol == 0 ? 0 : my_pcol == 1 ? 1 : my_pcol == 2 ? 2 : my_pcol == 4 ? 3 : my_pcol == 5 ? 4 : my_pcol == 7 ? 5 : -1;
    if (dst >= 0) {
#pragma unroll
      for (int c = 0; c < kN; ++c) H[dst * kN + c] = m[kN + c] * ip;
    }
  }
  for (int e = gl; e < 4 * kN; e += 16) {   // rows 6..9: -1 at (6,0) (7,1) (8,3) (9,6)
    const int rr = 6 + e / kN, cc = e - kN * (e / kN);
    const int one = rr == 6 ? 0 : rr == 7 ? 1 : rr == 8 ? 3 : 6;
    H[rr * kN + cc] = cc == one ? -1.0 : 0.0;
  }
  wave_sync();
  return going;
}

// ---- 5a. hessenberg() on this row's H and v ----
__device__ __forceinline__ void hessenberg4(double* __restrict__ H, double* __restrict__ v, int lane) {
  const int gl = lane & 15;
#pragma unroll
  for (int k = 0; k < kN - 2; ++k) {
    double s = 0.0;
    for (int i = k + 2; i < kN; ++i) { const double t = H[i * kN + k]; s += t * t; }   // (row-uniform reads)
    const bool act = s != 0.0;   // (solve(): `continue`)
    const double x0 = H[(k + 1) * kN + k];
    const double norm = sqrt(x0 * x0 + s);
    const double v0 = x0 + (x0 >= 0.0 ? norm : -norm);
    const double tau = 1.0 / (norm * fabs(v0));   // 2 / (v^T v)
    wave_sync();
    if (act && gl > k && gl < kN) v[gl] = gl == k + 1 ? v0 : H[gl * kN + k];
    wave_sync();
    if (act && gl >= k && gl < kN) {   // (I - tau v v^T) H: column gl
      double d = 0.0;
      for (int i = k + 1; i < kN; ++i) d += v[i] * H[i * kN + gl];
      d *= tau;
      for (int i = k + 1; i < kN; ++i) H[i * kN + gl] -= d * v[i];
    }
    wave_sync();
    if (act && gl < kN) {   // H (I - tau v v^T): row gl
      double d = 0.0;
      for (int j = k + 1; j < kN; ++j) d += H[gl * kN + j] * v[j];
      d *= tau;
      for (int j = k + 1; j < kN; ++j) H[gl * kN + j] -= d * v[j];
    }
    wave_sync();
    if (act && gl > k + 1 && gl < kN) H[gl * kN + k] = 0.0;
    wave_sync();
  }
}

// ---- 5e. eigenvalues_aberth() on this row's matrix; `alive`: the row still carries a sample. Returns (row-uniform) whether wr / wi hold
// the eigenvalues; false where solve() returns false: the caller runs hqr on the same H ----
#ifdef MVGX_FIVE_POINT_COUNT_ROUNDS
__device__ unsigned long long g_fallback_cause[8];   // why a row's eigenvalues were left to hqr (measurement build)
#define FP_CAUSE(i, was_ok) do { if ((lane & 15) == 0 && (was_ok) && !ok) atomicAdd(&g_fallback_cause[i], 1ull); } while (0)
#else
#define FP_CAUSE(i, was_ok) do { } while (0)
#endif
__device__ __forceinline__ bool eigenvalues_aberth4(const double* __restrict__ H, double* __restrict__ wr, double* __restrict__ wi,
                                                    double* __restrict__ ps /* kPolyScratch */, double* __restrict__ rsub /* 10: v */, int lane, bool alive) {
  const int gl = lane & 15;
  double* const X = ps;              // [i][k]: coefficient k of x_i
  double* const coef = ps + 110;     // monic: coef[k], k = 0..10
  double* const zre = coef + 12;
  double* const zim = zre + 10;
  double* const rad = zim + 10;
  bool ok = alive;
#ifdef MVGX_FIVE_POINT_STAMPS   // (slots 6, 7: parts of the eigenvalue stage - polynomial + start radii, the iteration)
  long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  // ---- 1. characteristic polynomial ----
  {
    bool okl = true;
    if (gl >= 1 && gl < kN) {
      const double h = H[gl * kN + gl - 1];
      okl = h != 0.0 && finite_d(h);
      rsub[gl] = okl ? 1.0 / h : 0.0;
    }
    { const bool was_ok_ = ok; const uint32_t votes = row_ballot(!okl, lane); ok = ok && !votes; FP_CAUSE(1, was_ok_); }
  }
  if (gl <= kN) X[(kN - 1) * 11 + gl] = gl == 0 ? 1.0 : 0.0;
  wave_sync();
#pragma unroll
  for (int i = kN - 1; i >= 1; --i) {
    if (gl <= kN) {
      double t = (gl > 0 ? X[i * 11 + gl - 1] : 0.0) - H[i * kN + i] * X[i * 11 + gl];
#pragma unroll
      for (int j = i + 1; j < kN; ++j) t -= H[i * kN + j] * X[j * 11 + gl];
      X[(i - 1) * 11 + gl] = t * rsub[i];
    }
    wave_sync();
  }
  double mine = 0.0;
  if (gl <= kN) {
    double t = (gl > 0 ? X[gl - 1] : 0.0) - H[0] * X[gl];
#pragma unroll
    for (int j = 1; j < kN; ++j) t -= H[j] * X[j * 11 + gl];
    mine = t;
  }
  const double lead = row_value_f64(mine, kN, lane);
  { const bool was_ok_ = ok; ok = ok && lead != 0.0 && finite_d(lead); FP_CAUSE(0, was_ok_); }
  mine = mine / lead;
  { const bool was_ok_ = ok; const uint32_t votes = row_ballot(gl <= kN && !finite_d(mine), lane); ok = ok && !votes; FP_CAUSE(2, was_ok_); }
  if (gl <= kN) coef[gl] = mine;
  wave_sync();
  // ---- 2a. start radii: Newton polygon (one step per hull edge; rows that are through - or out - wait for the others) ----
  {
    const double la = (gl <= kN && fabs(mine) > 0.0) ? log(fabs(mine)) : -1.0e300;
    int k1 = ok ? 0 : kN;
#pragma unroll 1
    while (__ballot(k1 < kN)) {
      const bool act = k1 < kN;
      const int k1c = act ? k1 : 0;
      const double lk1 = row_value_f64(la, k1c, lane);
      const bool cand = act && gl > k1c && gl <= kN;
      const double sl = cand ? (la - lk1) * frcp((double)(gl - k1c)) : -1.0e308;
      unsigned long long bits = (unsigned long long)__double_as_longlong(sl);
      bits = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
      const uint32_t hi = (uint32_t)(bits >> 32);
      const uint32_t best_hi = row_max_u32(cand ? hi : 0u);
      const uint32_t lo = (cand && hi == best_hi) ? (((uint32_t)bits & ~15u) | (uint32_t)gl) : 0u;
      const uint32_t best_lo = row_max_u32(lo);
      const int best = (int)(best_lo & 15u);
      const double slope = row_value_f64(sl, best, lane);
      const double r = exp(-slope);
      if (act) {
        if (best <= k1) k1 = kN;   // (cannot happen: lane kN is always a candidate)
        else {
          if (gl >= k1 && gl < best) rad[gl] = r;
          k1 = best;
        }
      }
    }
  }
  wave_sync();
  FP_STAMP(6);
  // ---- 2b. Ehrlich-Aberth ----
  double zr = 0.0, zi = 0.0;
  if (gl < kN) {
    double r = rad[gl];
    if (!(r > 1.0e-150)) r = 1.0e-150;
    if (!(r < 1.0e150)) r = 1.0e150;
    const double ang = 0.62831853071795865 * (double)gl + 0.7;   // 2 pi / 10 apart, off the axes
    zr = r * cos(ang); zi = r * sin(ang);
    zre[gl] = zr; zim[gl] = zi;
  }
  wave_sync();
  bool done = gl >= kN || !ok;
  const bool lane_in = gl < kN && ok;
  int it = 0;
  for (; it < kAberthIter; ++it) {   // (wave-uniform: until every row is through)
    double wr_ = 0.0, wi_ = 0.0;
    if (lane_in && !done) {
      double pr = 1.0, pi = 0.0, dr = 0.0, di = 0.0;
      double a[kN];
#pragma unroll
      for (int k = 0; k < kN; ++k) a[k] = coef[k];
#pragma unroll
      for (int k = kN - 1; k >= 0; --k) {
        const double ndr = dr * zr - di * zi + pr, ndi = dr * zi + di * zr + pi;
        const double npr = pr * zr - pi * zi + a[k], npi = pr * zi + pi * zr;
        dr = ndr; di = ndi; pr = npr; pi = npi;
      }
      const double dn = dr * dr + di * di;
      double nr = 0.0, ni = 0.0;
      if (dn > 0.0 && finite_d(dn)) { const double idn = frcp(dn); nr = (pr * dr + pi * di) * idn; ni = (pi * dr - pr * di) * idn; }
      double sr = 0.0, si = 0.0;
      {
        double ar[kN], ai[kN], ia[kN];
#pragma unroll
        for (int j = 0; j < kN; ++j) { ar[j] = zr - zre[j]; ai[j] = zi - zim[j]; }
#pragma unroll
        for (int j = 0; j < kN; ++j) { const double an = ar[j] * ar[j] + ai[j] * ai[j]; ia[j] = (j != gl && an > 0.0) ? frcp(an) : 0.0; }
#pragma unroll
        for (int j = 0; j < kN; ++j) { sr += ar[j] * ia[j]; si -= ai[j] * ia[j]; }
      }
      const double er = 1.0 - (nr * sr - ni * si), ei = -(nr * si + ni * sr);
      const double en = er * er + ei * ei;
      if (en > 0.0 && finite_d(en)) { const double ie = frcp(en); wr_ = (nr * er + ni * ei) * ie; wi_ = (ni * er - nr * ei) * ie; }
      else { wr_ = nr; wi_ = ni; }
    }
    wave_sync();   // every lane has read the roots of this round
    if (lane_in && !done) {
      zr -= wr_; zi -= wi_;
      zre[gl] = zr; zim[gl] = zi;
      const double zz = zr * zr + zi * zi, ww = wr_ * wr_ + wi_ * wi_;
      done = ww <= 1.0e-24 * zz || (zz == 0.0 && ww == 0.0);
      if (!finite_d(zz)) done = false;
    }
    wave_sync();
    if (!__ballot(!done)) break;
  }
#ifdef MVGX_FIVE_POINT_COUNT_ROUNDS
  if (lane == 0) { atomicAdd(&g_aberth_rounds, (unsigned long long)(it + 1)); atomicAdd(&g_aberth_solves, 1ull); }
#endif
  FP_STAMP(7);
  { const bool was_ok_ = ok; const uint32_t votes = row_ballot(!done, lane); ok = ok && !votes; FP_CAUSE(3, was_ok_); }   // no convergence (or a non-finite iterate): hqr decides
  // ---- 3. real roots: polished on the matrix ----
  bool real = false;
  double x = zr;
  if (ok && gl < kN) {
    const double az = sqrt(zr * zr + zi * zi);
    real = fabs(zi) <= 1.0e-6 * az || az == 0.0;
  }
  bool bad = false;
  if (real) {
    double step = 0.0;
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
      double q, dq;
      hyman(H, rsub, x, q, dq);
      step = dq != 0.0 ? q * frcp(dq) : 0.0;
      x -= step;
    }
    const bool converged = fabs(step) <= 1.0e-9 * fabs(x) + 1.0e-300 && finite_d(x);
    if (!converged) { if (fabs(zi) <= 1.0e-12 * fabs(zr)) bad = true; real = false; }
  }
  FP_STAMP(8);
  { const bool was_ok_ = ok; const uint32_t votes = row_ballot(bad, lane); ok = ok && !votes; FP_CAUSE(4, was_ok_); }
  if (gl < kN) { zre[gl] = real ? x : 0.0; zim[gl] = real ? 0.0 : 1.0; }
  wave_sync();
  bool dup = false;
  if (real) {
#pragma unroll
    for (int j = 0; j < kN; ++j)
      if (j != gl && zim[j] == 0.0 && fabs(zre[j] - x) <= 1.0e-10 * fabs(x)) dup = true;
  }
  { const bool was_ok_ = ok; const uint32_t votes = row_ballot(dup, lane); ok = ok && !votes; FP_CAUSE(5, was_ok_); }
  if (ok && gl < kN) { wr[gl] = real ? x : zr; wi[gl] = real ? 0.0 : (zi != 0.0 ? zi : 1.0); }
  wave_sync();
  FP_STAMP(9);
  return ok;
}

// FivePointSolver::Solve on four samples: s = the sample of this lane's row (rows whose sample is a copy of another row's are the
// caller's business). scr4: 4 x kScratch doubles of wave-private LDS, Es4: 4 x 90 doubles; row g's essential matrices go to
// Es4 + 90 g. Returns the number of solutions of this lane's row.
__device__ __forceinline__ int solve4(const double* __restrict__ b1, const double* __restrict__ b2, const uint32_t (&s)[5], int lane,
                                      double* __restrict__ scr4, double* __restrict__ Es4) {
  const int gl = lane & 15, g = lane >> 4;
  double* const scr = scr4 + g * kScratch;
  double* const H = scr;
  double* const v = scr + 100;
  double* const wr = v + 10;
  double* const wi = wr + 10;
  double* const basis_lds = wi + 10;
  double* const Es = Es4 + 90 * g;
#ifdef MVGX_FIVE_POINT_STAMPS
  long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  nullspace4(b1, b2, s, lane, basis_lds);
  FP_STAMP(0);
  bool alive;
  {
    double m[20];
    constraint_row(basis_lds, gl < kN ? gl : kN - 1, m, basis_lds + 36);   // (the polynomial scratch is free until the eigenvalue stage)
    wave_sync();
    FP_STAMP(1);
    alive = action_matrix4(m, lane, H);
    FP_STAMP(2);
  }
  double At_row[kN];   // this lane's row of the action matrix (lanes gl < 10): kept for the eigenvectors
#pragma unroll
  for (int c = 0; c < kN; ++c) At_row[c] = H[(gl < kN ? gl : 0) * kN + c];
  hessenberg4(H, v, lane);
  FP_STAMP(3);
#if MVGX_FIVE_POINT_ABERTH
  const bool have = eigenvalues_aberth4(H, wr, wi, basis_lds + 36, v, lane, alive);
#else
  const bool have = false;
#endif
  {
    // rows whose eigenvalues did not come out: solve()'s own fall-back, wave-wide on that row's matrix, one row after the other
    const unsigned long long need = __ballot(alive && !have);
    for (int q = 0; q < 4; ++q) {   // (wave-uniform)
      if (!((need >> (16 * q)) & 1ull)) continue;
      double* const sq = scr4 + q * kScratch;
#if MVGX_FIVE_POINT_ABERTH
      if (lane == 0) atomicAdd(&g_hqr_fallbacks, 1ull);
#endif
      const bool okq = hqr(sq, sq + 110, sq + 120, lane);
      if (g == q && !okq) alive = false;
    }
  }
  FP_STAMP(4);
  // the action matrix again (the iteration worked in place), then one eigenvector per lane
  wave_sync();
#pragma unroll
  for (int c = 0; c < kN; ++c)
    if (gl < kN) H[gl * kN + c] = At_row[c];
  wave_sync();
  const double lam = wr[gl < kN ? gl : 0];
  const bool real = alive && gl < kN && wi[gl < kN ? gl : 0] == 0.0 && lam == lam && fabs(lam) < 1.0e150;
  double tail[4] = {0.0, 0.0, 0.0, 0.0};
  if (real) eigenvector_tail(H, lam, tail);
  const uint32_t real_mask = row_ballot(real, lane);
  const int n = (int)__popc(real_mask);
  const int slot = (int)__popc(real_mask & ((1u << gl) - 1u));
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

#pragma clang fp contract(fast)   // (the toolchain's default for the rest of the unit)
