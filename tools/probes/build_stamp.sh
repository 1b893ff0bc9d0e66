mkdir -p /root/repo/tools/probes/stamp
set -e
cd /root/repo/tgp/pytorch_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DTGP_STAMPS"
for f in tgp_api tgp_mm tgp_lik tgp_rows tgp_big; do /opt/rocm/bin/hipcc $F -c $f.hip -o ../../../tools/probes/stamp/$f.o & done
/opt/rocm/bin/hipcc $F -DTGP_MT=7 -c tgp_rows_inst.hip -o ../../../tools/probes/stamp/mt7.o
wait
cd ../../..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/stamp/libtgp_hip.so tools/probes/stamp/*.o
