// Developer probe (GPU box): relative error of v_rcp_f64 and of one / two Newton steps on it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/rcp_accuracy tools/probe/rcp_accuracy.hip && /tmp/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const double *x, double *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double e = __builtin_fma(-v, r0, 1.0);
  double r1 = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-v, r1, 1.0);
  double r2 = __builtin_fma(r1, e, r1);
  out[3 * i] = r0;
  out[3 * i + 1] = r1;
  out[3 * i + 2] = r2;
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), out(3 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = double(s >> 11) / 9007199254740992.0;
    x[i] = std::ldexp(1.0 + u, int(s % 80) - 40) * ((s >> 3) & 1 ? -1.0 : 1.0);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8);
  hipMalloc(&dout, 3 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  probe<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(out.data(), dout, 3 * n * 8, hipMemcpyDeviceToHost);
  double err[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3; ++j) {
      const long double exact = 1.0L / (long double)x[i];
      const double rel = double(fabsl(((long double)out[3 * i + j] - exact) / exact));
      if (rel > err[j]) err[j] = rel;
    }
  std::printf("max relative error: v_rcp_f64 %.3e, one Newton step %.3e, two steps %.3e\n", err[0], err[1], err[2]);
  return 0;
}
