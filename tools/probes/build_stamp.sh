# diagnostic build of the whole library with -DTGP_STAMPS into tools/probes/stamp/ (see README.md)
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
mkdir -p "$ROOT/tools/probes/stamp"
cd "$ROOT/tgp/pytorch_amd/csrc"
O=../../../tools/probes/stamp
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DTGP_STAMPS $TGP_EXTRA"   # (the Makefile's flags)
for f in tgp_api tgp_comm tgp_mm tgp_lik tgp_rows tgp_big tgp_gemm128 tgp_kmeans tgp_mlp; do /opt/rocm/bin/hipcc $F -c $f.hip -o $O/$f.o & done
wait
# (hipcc 7.2 crashes in its 'Rewrite AGPR-Copy-MFMA' pass on the STAMPED k_rows<8,8,1,10>; the shipped, unstamped build is fine: an
#  object that fails is built again with the MFMA form left to the compiler's heuristic -- diagnostics only)
F0="${F/-mllvm -amdgpu-mfma-vgpr-form/}"
for n in 1 2 3 4 5 6 7 8; do
  ( /opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=iterative-ilp -DTGP_MT=$n -c tgp_rows_inst.hip -o $O/mt$n.o 2> $O/mt$n.err ||
    /opt/rocm/bin/hipcc $F0 -mllvm -amdgpu-sched-strategy=iterative-ilp -DTGP_MT=$n -c tgp_rows_inst.hip -o $O/mt$n.o ) &
done
wait
cd ../../..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/stamp/libtgp_hip.so tools/probes/stamp/*.o -ldl
