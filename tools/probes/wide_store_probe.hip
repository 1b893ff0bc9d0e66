// wide_store_probe.hip -- the write-out of one 16x16 tile from an LDS matrix with odd row stride (LD = 113), four 16-byte
// stores per lane, in the variants tried for tgp_prep.hpp's write_L; checks the result on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __amdgpu_buffer_rsrc_t ws_rsrc_t;
__device__ __forceinline__ void st2(ws_rsrc_t rs, size_t off, double a, double b) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const u4v v = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(off * 8), 0, AUX);
}
#define MP 112
#define LD 113
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(double* ws, int ntile) {
  __shared__ double A[MP * LD];
  for (int i = threadIdx.x; i < MP * LD; i += blockDim.x) A[i] = (i / LD) * 1000.0 + (i % LD) + 0.123456789012345;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const ws_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7fffffff, 0x00020000);
  for (int rep = 0; rep < 50; ++rep) for (int ti = 1; ti < 7; ++ti) for (int tj = 0; tj < ti; ++tj) {
    if (((ti * 7 + tj) & 3) != wv) continue;
    const int rw = lane >> 3, cp = 2 * (lane & 7);
    double a[2][2], b[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * h + rw;
      a[h][0] = A[(16 * ti + row) * LD + 16 * tj + cp];
      a[h][1] = A[(16 * ti + row) * LD + 16 * tj + cp + 1];
      b[h][0] = A[(16 * ti + cp) * LD + 16 * tj + row];
      b[h][1] = A[(16 * ti + cp + 1) * LD + 16 * tj + row];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * h + rw;
      st2(rs, (size_t)(16 * ti + row) * MP + 16 * tj + cp, a[h][0], a[h][1]);
      st2(rs, (size_t)MP * MP + (size_t)(16 * tj + row) * MP + 16 * ti + cp, b[h][0], b[h][1]);
    }
  }
}
int main() {
  double* ws; hipMalloc(&ws, 2 * MP * MP * 8); hipMemset(ws, 0, 2 * MP * MP * 8);
  k<<<1, 256>>>(ws, 40); hipDeviceSynchronize();
  double* h = (double*)malloc(2 * MP * MP * 8); hipMemcpy(h, ws, 2 * MP * MP * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 16; r < MP; ++r) for (int c = 0; c < (r / 16) * 16; ++c) {
    const double want = r * 1000.0 + c + 0.123456789012345;
    if (h[r * MP + c] != want) { if (bad < 8) printf("L[%d][%d] = %.17g want %.17g\n", r, c, h[r * MP + c], want); ++bad; }
    if (h[MP * MP + c * MP + r] != want) { if (bad < 8) printf("LT[%d][%d] = %.17g want %.17g\n", c, r, h[MP * MP + c * MP + r], want); ++bad; }
  }
  printf("aux=%d: %d bad entries\n", AUX, bad);
  return 0;
}
