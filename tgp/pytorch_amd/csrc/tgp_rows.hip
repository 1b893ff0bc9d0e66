// tgp_rows.hip -- dispatch of the fused row kernel over its compile-time tilings (MT = ceil(M/16), DP)
#include "tgp_rows.hpp"
#include "tgp_launch.hpp"

namespace tgp {

#define DECL(n) int launch_rows_mt##n(const RowArgs& a, int mode, size_t lds, hipStream_t st);
DECL(1) DECL(2) DECL(3) DECL(4) DECL(5) DECL(6) DECL(7) DECL(8)
#undef DECL

int launch_rows(const Plan& p, const tgp_model& md, const FlowProg& fp, const double* X, const double* Y,
                const double* rowp, double* g_rowp, double* mu, double* v, double* ws, bool train, hipStream_t st) {
  RowArgs a;
  a.p = p;
  a.X = X; a.Y = Y; a.rowp = rowp; a.g_rowp = g_rowp; a.mu = mu; a.v = v; a.ws = ws;
  a.prog = fp; a.xs = md.xs; a.wn = md.wn; a.scale = md.scale;
  const size_t lim = 160 * 1024 - 1024;
  int mode = 0;
  size_t lds = row_lds(p, 0, 0).total * sizeof(double);
  if (train) {
    mode = 1;
    lds = row_lds(p, 1, fp.nslots).total * sizeof(double);
    if (lds > lim) {
      mode = 2;
      lds = row_lds(p, 2, fp.nslots).total * sizeof(double);
    }
  }
  if (lds > lim) return TGP_E_LDS;
  switch (p.MT) {
    case 1: return launch_rows_mt1(a, mode, lds, st);
    case 2: return launch_rows_mt2(a, mode, lds, st);
    case 3: return launch_rows_mt3(a, mode, lds, st);
    case 4: return launch_rows_mt4(a, mode, lds, st);
    case 5: return launch_rows_mt5(a, mode, lds, st);
    case 6: return launch_rows_mt6(a, mode, lds, st);
    case 7: return launch_rows_mt7(a, mode, lds, st);
    case 8: return launch_rows_mt8(a, mode, lds, st);
  }
  return TGP_E_UNSUPPORTED;
}

}  // namespace tgp
