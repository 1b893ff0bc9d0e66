// tgp_rows_inst.hip -- one translation unit per MT (compiled 8x with -DTGP_MT=1..8 so the big unrolled
// kernels build in parallel); each defines launch_rows_mt<N>().
#include "tgp_rows.hpp"
#include "tgp_rows4.hpp"
#include "tgp_launch.hpp"

#ifndef TGP_MT
#error "compile with -DTGP_MT=<1..8>"
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

namespace tgp {

template <int DP, int MODE, int RW = 16>
static int launch_one(const RowArgs& a, size_t lds, hipStream_t st) {
  constexpr bool TRAIN = MODE != 0;
  auto kern = k_rows<TGP_MT, DP, MODE, RW>;
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_cur)) return rc;
  // the row blocks, then (training) the MT passenger blocks
  const int grid = a.p.nblocks + (TRAIN ? a.p.MT : 0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(e, __FILE__, __LINE__);
  return 0;
}

// the 4-rows-per-wave kernel (tgp_rows4.hpp): NW waves per workgroup
template <int DP, bool TRAIN, int NW>
static int launch_one4(const RowArgs& a, hipStream_t st) {
  auto kern = k_rows4<TGP_MT, DP, TRAIN, NW>;
  const size_t lds = row4_lds(a.p, TRAIN, a.prog.nslots, NW).total * sizeof(double);
  static size_t lds_cur = 48 * 1024;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_cur)) return rc;
  const int grid = a.p.nblocks + (TRAIN ? a.p.MT : 0);   // Plan.nblocks = ceil(N / 4 NW); then the MT passenger blocks
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(e, __FILE__, __LINE__);
  return 0;
}

template <int DP>
static int launch_dp(const RowArgs& a, int mode, size_t lds, hipStream_t st) {
  switch (mode) {
    case 150 + 4: return launch_one4<DP, true, 4>(a, st);
    case 150 + 8: return launch_one4<DP, true, 8>(a, st);
    case 200: return launch_one<DP, 1, TGP_RW_SMALL>(a, lds, st);
    case 0: return launch_one<DP, 0>(a, lds, st);
    case 1: return launch_one<DP, 1>(a, lds, st);
    default: return launch_one<DP, 2>(a, lds, st);
  }
}

int CAT(launch_rows_mt, TGP_MT)(const RowArgs& a, int mode, size_t lds, hipStream_t st) {
  switch (a.p.DP) {
    case 4: return launch_dp<4>(a, mode, lds, st);
    case 8: return launch_dp<8>(a, mode, lds, st);
    default: return launch_dp<16>(a, mode, lds, st);
  }
}

#if TGP_MT == 1
size_t rows4_lds_bytes(const Plan& p, bool train, int nw) {
  const Row4Lds L = row4_lds(p, train, p.nslots, nw);
  return L.nb == 0 ? (size_t)1 << 30 : L.total * sizeof(double);   // (not even one quadrature node per lane fits: not a candidate)
}
#endif

}  // namespace tgp
