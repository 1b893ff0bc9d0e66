#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* o) {
  int i = threadIdx.x; double x = 0.37 + 1.731 * i;
  double r = __builtin_amdgcn_rcp(x), s = __builtin_amdgcn_rsq(x);
  o[4*i] = r * x - 1.0; o[4*i+1] = s * s * x - 1.0;
  double y = s, h = 0.5 * x; y = y * fma(-h * y, y, 1.5); o[4*i+2] = y * y * x - 1.0;
  double z = r; z = fma(fma(-x, z, 1.0), z, z); o[4*i+3] = z * x - 1.0;
}
int main() { double* d; hipMalloc(&d, 64*4*8); k<<<1,64>>>(d); double h[256]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
  double m[4] = {0,0,0,0}; for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) m[j] = fmax(m[j], fabs(h[4*i+j]));
  printf("max rel err: rcp %.2e  rsq(y^2 x - 1) %.2e  rsq+1NR %.2e  rcp+1NR %.2e\n", m[0], m[1], m[2], m[3]); return 0; }
