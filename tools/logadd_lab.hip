// logadd_lab.hip — the cost and the accuracy of LogAdd(double) on one lone wave (the lattice sweeps' dependent chain):
// ocml's exp + log1p against a version written for the only arguments LogAdd has, diff in [log(DBL_EPSILON), 0].
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math tools/logadd_lab.hip -o tools/logadd_lab && tools/logadd_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../old-kaldi-git_amd/csrc/kh_logadd.h"

__device__ __forceinline__ double LogAddOcml(double x, double y, double min_log_diff) {
  double diff;
  if (x < y) { diff = x - y; x = y; } else { diff = y - x; }
  if (diff >= min_log_diff) return x + log1p(exp(diff));
  return x;
}

template <int V>
__global__ void Chain(const double *in, double *out, long long *cyc, int n) {
  double a = in[threadIdx.x];
  const double step = in[64 + threadIdx.x];
  const double mld = -36.04365338911715;
  const long long t0 = clock64();
  for (int i = 0; i < n; i++) {
    const double y = a + step;   // (the next operand depends on the running value: a dependent chain, as in the sweeps)
    a = V == 0 ? LogAddOcml(a, y, mld) : kh::LogAddD(a, y, mld);
  }
  const long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V>
__global__ void Eval(const double *d, double *f, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] = V == 0 ? log1p(exp(d[i])) : kh::Log1pExpNeg(d[i]);
}

int main() {
  const int n = 1 << 20;
  std::vector<double> hd(n), f0(n), f1(n);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); };
  for (int i = 0; i < n; i++) {
    const double u = rnd();
    hd[i] = i % 4 == 0 ? -36.04365338911715 * u : i % 4 == 1 ? -u : i % 4 == 2 ? -1e-3 * u : -8.0 * u;
  }
  hd[0] = 0.0; hd[1] = -36.04365338911715; hd[2] = -0.6931471805599453; hd[3] = -1e-300;
  double *dd, *df;
  hipMalloc(&dd, n * 8); hipMalloc(&df, n * 8);
  hipMemcpy(dd, hd.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(Eval<0>, dim3(n / 256), dim3(256), 0, 0, dd, df, n);
  hipMemcpy(f0.data(), df, n * 8, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(Eval<1>, dim3(n / 256), dim3(256), 0, 0, dd, df, n);
  hipMemcpy(f1.data(), df, n * 8, hipMemcpyDeviceToHost);
  double worst[3] = {0, 0, 0};
  int at[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) {
    const long double ref = log1pl(expl(static_cast<long double>(hd[i])));
    const double r = static_cast<double>(ref);
    const double ulp = std::nextafter(std::fabs(r), INFINITY) - std::fabs(r);
    const double host = std::log1p(std::exp(hd[i]));
    const double e[3] = {std::fabs(static_cast<double>(f0[i] - ref)) / ulp, std::fabs(static_cast<double>(f1[i] - ref)) / ulp,
                         std::fabs(static_cast<double>(host - ref)) / ulp};
    for (int k = 0; k < 3; k++) if (e[k] > worst[k]) { worst[k] = e[k]; at[k] = i; }
  }
  printf("log1p(exp(d)), %d arguments in [log(DBL_EPSILON), 0], error against long double in ulp of the result: ocml %.2f (d = %.17g), "
         "own %.2f (d = %.17g), host libm %.2f\n", n, worst[0], hd[at[0]], worst[1], hd[at[1]], worst[2]);
  // cost on a lone wave
  std::vector<double> hin(128);
  for (int i = 0; i < 64; i++) { hin[i] = -100.0 - i; hin[64 + i] = -0.01 * (i + 1); }
  double *din, *dout;
  long long *dc, hc;
  hipMalloc(&din, 128 * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&dc, 8);
  hipMemcpy(din, hin.data(), 128 * 8, hipMemcpyHostToDevice);
  const int steps = 20000;
  for (int v = 0; v < 2; v++) {
    double res[2][64];
    for (int rep = 0; rep < 2; rep++) {
      if (v == 0) hipLaunchKernelGGL(Chain<0>, dim3(1), dim3(64), 0, 0, din, dout, dc, steps);
      else hipLaunchKernelGGL(Chain<1>, dim3(1), dim3(64), 0, 0, din, dout, dc, steps);
      hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
      hipMemcpy(res[rep], dout, 64 * 8, hipMemcpyDeviceToHost);
    }
    printf("%s: %.0f shader cycles per dependent LogAdd on a lone wave (chain of %d; lane 0 ends at %.17g)\n", v == 0 ? "ocml exp + log1p" : "own log1p(exp)",
           static_cast<double>(hc) / steps, steps, res[1][0]);
  }
  return 0;
}
