// tgp_rows.hip -- dispatch of the fused row kernel over its compile-time tilings (MT = ceil(M/16), DP)
#include "tgp_rows.hpp"
#include "tgp_launch.hpp"

namespace tgp {

#define DECL(n) int launch_rows_mt##n(const RowArgs& a, int mode, size_t lds, hipStream_t st);
DECL(1) DECL(2) DECL(3) DECL(4) DECL(5) DECL(6) DECL(7) DECL(8)
#undef DECL
size_t rows4_lds_bytes(const Plan& p, bool train, int nw);   // tgp_rows_inst.hip (needs tgp_rows4.hpp)

// Which row kernel: the 4-rows-per-wave kernel (tgp_rows4.hpp) where it measured faster -- a training launch (any
// likelihood) whose row blocks and passenger blocks number at most one per CU with one to spare (rows4_waves, tgp_dev.hpp: up to 32 (255 - MT)
// rows) -- and where its LDS plan fits a CU.  `sel` = tgp_model.plan & TGP_PLAN_ROWS_MASK: the caller's override for THIS call
// (A/B measurements, tests of a specific kernel); the selection is a property of the call, not of the process.
int choose_rows4(const Plan& p, bool train, int sel) {
  sel &= TGP_PLAN_ROWS_MASK;
  if (sel == TGP_PLAN_ROWS_K16 || sel == TGP_PLAN_ROWS_K || !train) return 0;
  int nw = rows4_waves(p.N, p.MT);
  if (sel == TGP_PLAN_ROWS4_NW4) nw = 4;
  if (sel == TGP_PLAN_ROWS4_NW8) nw = 8;
  if (nw == 0) return 0;
  if ((p.N + 4 * nw - 1) / (4 * nw) > plan_alloc_blocks(p.N)) return 0;   // (a forced size the workspace has no slabs for)
  const size_t need = rows4_lds_bytes(p, train, nw);
  if (need > 160 * 1024 - 2048) return 0;
  return nw;
}

// Rows per wave of k_rows: TGP_RW_SMALL (10) for a training launch with the flow likelihood and whose
// (row, node) pairs fill the lanes in one trip (10 S <= 64 x 5) at the sizes rows_rw() names, if the LDS plan fits; else 16.
// sel == TGP_PLAN_ROWS_K16 forces 16.
int rows_per_wave(const Plan& p, const FlowProg& fp, bool train, int sel) {
  if (!train || (sel & TGP_PLAN_ROWS_MASK) == TGP_PLAN_ROWS_K16 || p.lik != TGP_LIK_FLOW || p.nblk < 1) return 16;
  for (int b = 0; b < fp.nblk; ++b)      // per-row parameters: SAL blocks only (their per-pair partials use the block's own stack slots)
    if ((fp.blk[4 * b + 3] & TGP_FLAG_PER_ROW) && fp.blk[4 * b] != TGP_FLOW_SAL) return 16;
  if (rows_rw(p.N) != TGP_RW_SMALL || TGP_RW_SMALL * p.S > 64 * TGP_RW_NODES) return 16;
  if (row_lds(p, 1, p.nslots, TGP_RW_SMALL).total * sizeof(double) > 160 * 1024 - 1024) return 16;
  return TGP_RW_SMALL;
}

bool rows_train_lds_fits(const Plan& p, int nslots) {
  return row_lds(p, 2, nslots).total * sizeof(double) <= (size_t)160 * 1024 - 1024;
}

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
  if (p.nw4 > 0) mode = 100 + (train ? 50 : 0) + p.nw4;      // k_rows4<.., train, nw4> (its own LDS plan)
  else if (train && p.rw == TGP_RW_SMALL) {                 // k_rows<.., 1, 10>
    mode = 200;
    lds = row_lds(p, 1, fp.nslots, TGP_RW_SMALL).total * sizeof(double);
    if (lds > lim) return TGP_E_LDS;
  } else if (lds > lim) return TGP_E_LDS;
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
