// tgp_rows2_inst.hip -- the team-split row kernel (tgp_rows2.hpp), one translation unit per MT (compiled 8x with
// -DTGP_MT=1..8); each defines launch_rows2_mt<N>().
#include "tgp_rows2.hpp"
#include "tgp_launch.hpp"

#ifndef TGP_MT
#error "compile with -DTGP_MT=<1..8>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace tgp {

template <int DP>
static int launch2_one(const RowArgs& a, int T, size_t lds, hipStream_t st) {
  auto kern = k_rows2<TGP_MT, DP>;
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_cur)) return rc;
  hipLaunchKernelGGL(kern, dim3(a.p.nblocks + a.p.MT * a.p.MT + 1), dim3(256 * T), lds, st, a, T);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(e, __FILE__, __LINE__);
  return 0;
}

int CAT(launch_rows2_mt, TGP_MT)(const RowArgs& a, int T, size_t lds, hipStream_t st) {
  switch (a.p.DP) {
    case 4: return launch2_one<4>(a, T, lds, st);
    case 8: return launch2_one<8>(a, T, lds, st);
    default: return launch2_one<16>(a, T, lds, st);
  }
}

}  // namespace tgp
