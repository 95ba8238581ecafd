// kh_logadd.h — LogAdd(double) (base/kaldi-math.h:178-195) for the lattice sweeps.
//
// The sweeps' critical path is a chain of dependent LogAdds on ONE wave (a state folds its incoming arcs in ascending
// order; the states of the next frame wait for it), so what the function costs a lone wave in dependent fp64
// instructions is what a level of the lattice costs.  log1p(exp(diff)) is only ever asked for diff in
// [log(DBL_EPSILON), 0]: exp by one reduction step (n = rint(diff / ln 2)) and a degree-13 polynomial, log1p(t) for t in
// (0, 1] as log(u) + (t - (u - 1)) / u with u = 1 + t, log(u) = k ln 2 + 2 atanh(s), s = g / (2 + g), g = m - 1,
// m = u or u / 2 in [sqrt(1/2), sqrt(2)) - no special cases, no table.  tools/logadd_lab.hip: error against long double
// and cycles per dependent LogAdd, next to ocml's exp + log1p.
#pragma once
#include <hip/hip_runtime.h>

namespace kh {

__device__ __forceinline__ double Log1pExpNeg(double d) {
  // t = exp(d), d in [-36.05, 0]
  const double n = rint(d * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, d);   // ln 2, high part (32 bits: n * hi is exact)
  r = fma(-n, 1.90821492927058770002e-10, r);          // low part
  double p = 1.6059043836821613e-10;                   // 1 / 13!
  p = fma(p, r, 2.08767569878681e-09);                 // 1 / 12!
  p = fma(p, r, 2.505210838544172e-08);
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.7557319223985893e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 1.984126984126984e-04);
  p = fma(p, r, 1.388888888888889e-03);
  p = fma(p, r, 8.333333333333333e-03);
  p = fma(p, r, 4.1666666666666664e-02);
  p = fma(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double t = ldexp(p, static_cast<int>(n));
  // log1p(t), t in (0, 1]
  const double u = 1.0 + t;
  const double c = t - (u - 1.0);                      // what the rounding of 1 + t lost (exact)
  const bool hi = u > 1.4142135623730951;
  const double m = hi ? 0.5 * u : u;
  const double g = m - 1.0;                            // exact
  const double s = g / (2.0 + g);
  const double z = s * s;
  double q = 4.7619047619047616e-02;                   // 1 / 21
  q = fma(q, z, 5.2631578947368418e-02);               // 1 / 19
  q = fma(q, z, 5.8823529411764705e-02);
  q = fma(q, z, 6.6666666666666666e-02);
  q = fma(q, z, 7.6923076923076927e-02);
  q = fma(q, z, 9.0909090909090912e-02);
  q = fma(q, z, 1.1111111111111111e-01);
  q = fma(q, z, 1.4285714285714285e-01);
  q = fma(q, z, 0.2);
  q = fma(q, z, 3.3333333333333331e-01);
  const double s2 = s + s;
  double lg = fma(s2 * z, q, s2);                      // 2 atanh(s)
  lg += c * __builtin_amdgcn_rcp(u);                   // (a correction of at most 2^-53: the reciprocal's 2^-20 is plenty)
  return hi ? lg + 6.9314718055994529e-01 : lg;
}

__device__ __forceinline__ double LogAddD(double x, double y, double min_log_diff) {
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= min_log_diff) return x + Log1pExpNeg(diff);
  return x;
}

}  // namespace kh
