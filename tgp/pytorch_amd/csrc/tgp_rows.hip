// tgp_rows.hip -- dispatch of the fused row kernel over its compile-time tilings (MT = ceil(M/16), DP)
#include <cstdlib>
#include "tgp_rows2.hpp"
#include "tgp_launch.hpp"

namespace tgp {

#define DECL(n) int launch_rows_mt##n(const RowArgs& a, int mode, size_t lds, hipStream_t st);
DECL(1) DECL(2) DECL(3) DECL(4) DECL(5) DECL(6) DECL(7) DECL(8)
#undef DECL
#define DECL(n) int launch_rows2_mt##n(const RowArgs& a, int T, size_t lds, hipStream_t st);
DECL(1) DECL(2) DECL(3) DECL(4) DECL(5) DECL(6) DECL(7) DECL(8)
#undef DECL

// Which row kernel serves a training step of this shape: the team-split kernel (tgp_rows2.hpp) when the batch has at most
// 3 x 256 16-row groups and its LDS image (exchange regions | one-node flow stack | statistics tile, plus the flow
// gradient accumulators) fits one CU; the one-wave-per-group kernel otherwise.  Deterministic in (shape, flow program):
// every phase call of one step takes the same decision.
// Round-2 measurements: the team-split kernel occupies all 256 CUs but three co-resident waves per
// SIMD issue f64 MFMAs 80-100 cycles apart instead of 64 and the one-node-in-flight flow sweep loses the amortisation of
// the four-node one -- 54 us against 55 us (tanh 3x2), 44 against 47 (SAL x 2), 34 against 37 (SVGP) at Power size.
// Not a clear win, so it is OFF by default: tgp_set_rows_kernel(1) or TGP_ROWS2=1 selects it (tests run both).
static int g_rows_kernel = -1;  // -1: not decided yet (environment), 0: one wave per group, 1: team-split when eligible
void set_rows_kernel(int mode) { g_rows_kernel = mode != 0; }
void select_rows_kernel(Plan& p, const FlowProg& fp) {
  if (g_rows_kernel < 0) {
    const char* e = getenv("TGP_ROWS2");
    g_rows_kernel = (e != nullptr && atoi(e) != 0) ? 1 : 0;
  }
  const int enabled = g_rows_kernel;
  p.T2 = 0;
  const int T = row2_teams(p.N);
  if (!enabled || T == 0) return;
  if (row2_lds(p, T, fp.nslots).total * sizeof(double) > (size_t)160 * 1024 - 1024) return;
  p.T2 = T;
  p.nblocks = p.nb2;
}

int launch_rows(const Plan& p, const tgp_model& md, const FlowProg& fp, const double* X, const double* Y,
                const double* rowp, double* g_rowp, double* mu, double* v, double* ws, bool train, hipStream_t st) {
  RowArgs a;
  a.p = p;
  a.X = X; a.Y = Y; a.rowp = rowp; a.g_rowp = g_rowp; a.mu = mu; a.v = v; a.ws = ws;
  a.prog = fp; a.xs = md.xs; a.wn = md.wn; a.scale = md.scale;
  if (train && p.T2 > 0) {
    const size_t lds2 = row2_lds(p, p.T2, fp.nslots).total * sizeof(double);
    switch (p.MT) {
      case 1: return launch_rows2_mt1(a, p.T2, lds2, st);
      case 2: return launch_rows2_mt2(a, p.T2, lds2, st);
      case 3: return launch_rows2_mt3(a, p.T2, lds2, st);
      case 4: return launch_rows2_mt4(a, p.T2, lds2, st);
      case 5: return launch_rows2_mt5(a, p.T2, lds2, st);
      case 6: return launch_rows2_mt6(a, p.T2, lds2, st);
      case 7: return launch_rows2_mt7(a, p.T2, lds2, st);
      case 8: return launch_rows2_mt8(a, p.T2, lds2, st);
    }
    return TGP_E_UNSUPPORTED;
  }
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
