// tgp_rows_inst.hip -- one translation unit per MT (compiled 8x with -DTGP_MT=1..8 so the big unrolled
// kernels build in parallel); each defines launch_rows_mt<N>().
#include "tgp_rows.hpp"
#include "tgp_launch.hpp"

#ifndef TGP_MT
#error "compile with -DTGP_MT=<1..8>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace tgp {

template <int DP, bool TRAIN>
static int launch_one(const RowArgs& a, size_t lds, hipStream_t st) {
  auto kern = k_rows<TGP_MT, DP, TRAIN>;
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_cur)) return rc;
  hipLaunchKernelGGL(kern, dim3(a.p.nblocks), dim3(256), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(e, __FILE__, __LINE__);
  return 0;
}

int CAT(launch_rows_mt, TGP_MT)(const RowArgs& a, bool train, size_t lds, hipStream_t st) {
  switch (a.p.DP) {
    case 4: return train ? launch_one<4, true>(a, lds, st) : launch_one<4, false>(a, lds, st);
    case 8: return train ? launch_one<8, true>(a, lds, st) : launch_one<8, false>(a, lds, st);
    default: return train ? launch_one<16, true>(a, lds, st) : launch_one<16, false>(a, lds, st);
  }
}

}  // namespace tgp
