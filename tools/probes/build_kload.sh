#!/bin/bash
# timing-only variant of the library for VERDICT r5 #1b (the covariance formed in k_gemm's A-operand loader): a scratch copy of
# csrc/ with tools/probes/kload.patch applied (-DTGP_GEMM_EXP_KLOAD in gemm_tile), tgp_big.o rebuilt from it, linked with the
# shipped objects into tools/probes/kload/, and gemm_bench linked against both libraries.   bash tools/probes/build_kload.sh
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
K="$ROOT/tools/probes/kload"
rm -rf "$K"; mkdir -p "$K/tgp/pytorch_amd" "$K/include"
cp -r "$ROOT/tgp/pytorch_amd/csrc" "$K/tgp/pytorch_amd/csrc"; cp "$ROOT/include/tgp_hip.h" "$K/include/"
rm -rf "$K/tgp/pytorch_amd/csrc/build"
(cd "$K" && patch -p1 -s < "$ROOT/tools/probes/kload.patch")
make -C "$ROOT/tgp/pytorch_amd/csrc" -j8 > /dev/null
(cd "$K/tgp/pytorch_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -DTGP_GEMM_EXP_KLOAD -c tgp_big.hip -o "$K/tgp_big.o")
OBJS=$(ls "$ROOT"/tgp/pytorch_amd/csrc/build/*.o | grep -v tgp_big.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$K/libtgp_hip.so" $OBJS "$K/tgp_big.o" -ldl
cd "$ROOT/tools/probes"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm_bench.hip -o kload/gemm_bench -Lkload -ltgp_hip -Wl,-rpath,'$ORIGIN'
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm_bench.hip -o gemm_bench -L../../tgp/pytorch_amd -ltgp_hip -Wl,-rpath,'$ORIGIN/../../tgp/pytorch_amd'
rm -rf "$K/tgp" "$K/include"
