// tgp_gemm128.hip -- the 128 x 128-tile GEMM kernels of tgp_gemm.hpp (k_gemm, k_gemm_pair) and their launchers, as a
// translation unit of their own so that the Makefile can give them their own instruction scheduler (GEMM128_EXTRA):
// the arguments are prepared (gemm_normalise) and the tiling is chosen by launch_gemm in tgp_big.hip.
#include <hip/hip_runtime.h>

#include "tgp_dev.hpp"
#include "tgp_gemm.hpp"
#include "tgp_launch.hpp"

namespace tgp {

#define LAUNCH_CHECK()                                              \
  do {                                                              \
    hipError_t e_ = hipGetLastError();                              \
    if (e_ != hipSuccess) return set_error(e_, __FILE__, __LINE__); \
  } while (0)

template <bool TA, bool TB, bool MOD, bool EPI>
static int launch_gemm_t(const GemmArgs& g, hipStream_t st) {
  static bool attr_done = false;
  const void* f = reinterpret_cast<const void*>(k_gemm<TA, TB, MOD, EPI>);
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_error(e, __FILE__, __LINE__); }
    attr_done = true;
  }
  const int nj = g.n / GT, nb = g.m / GT;
  dim3 grid(g.pair ? (nj + 1) / 2 : nj, nb, g.ksplit), block(256);
  if (g.xcd == 3) grid = dim3(nb * (nb + 1) / 2 * g.ksplit, 1, 1);
  if (g.xcd == 4) grid.y = (nb + 7) & ~7;   // whole groups of 8 tile rows (one per XCD); the padding rows return at once
  hipLaunchKernelGGL((k_gemm<TA, TB, MOD, EPI>), grid, block, GEMM_LDS_BYTES, st, g);
  LAUNCH_CHECK();
  return 0;
}

template <bool MOD, bool EPI>
static int launch_gemm_l(bool ta, bool tb, const GemmArgs& g, hipStream_t st) {
  if (ta && tb) return launch_gemm_t<true, true, MOD, EPI>(g, st);
  if (ta) return launch_gemm_t<true, false, MOD, EPI>(g, st);
  if (tb) return launch_gemm_t<false, true, MOD, EPI>(g, st);
  return launch_gemm_t<false, false, MOD, EPI>(g, st);
}

int launch_gemm128(bool ta, bool tb, bool mod, bool epi, const GemmArgs& g, hipStream_t st) {
  if (mod) return epi ? launch_gemm_l<true, true>(ta, tb, g, st) : launch_gemm_l<true, false>(ta, tb, g, st);
  return epi ? launch_gemm_l<false, true>(ta, tb, g, st) : launch_gemm_l<false, false>(ta, tb, g, st);
}

// a (op(B) transposed) and b (no transposition) in one launch; both normalised, both plain or both with the C epilogue
int launch_gemm128_pair_ft_ff(bool epi, const GemmArgs& a, const GemmArgs& b, int na, int gxa, int gya, int gxb, int gyb,
                              hipStream_t st) {
  const int nbk = gxb * gyb;
  static bool attr_done[2] = {false, false};
  const void* f = epi ? reinterpret_cast<const void*>(k_gemm_pair<false, true, false, true, false, false, false, true>)
                      : reinterpret_cast<const void*>(k_gemm_pair<false, true, false, false, false, false, false, false>);
  if (!attr_done[epi]) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_error(e, __FILE__, __LINE__); }
    attr_done[epi] = true;
  }
  if (epi)
    hipLaunchKernelGGL((k_gemm_pair<false, true, false, true, false, false, false, true>), dim3(na + nbk), dim3(256), GEMM_LDS_BYTES, st,
                       a, b, na, gxa, gya, gxb, gyb);
  else
    hipLaunchKernelGGL((k_gemm_pair<false, true, false, false, false, false, false, false>), dim3(na + nbk), dim3(256), GEMM_LDS_BYTES, st,
                       a, b, na, gxa, gya, gxb, gyb);
  LAUNCH_CHECK();
  return 0;
}

}  // namespace tgp
