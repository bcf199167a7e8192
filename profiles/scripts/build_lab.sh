# builds profiles/gemm_lab (against the in-tree library) and profiles/gemm_lab_stamps (against a copy of the library
# compiled with -DADN_GEMM_STAMPS under profiles/stamps/).  Run from the repo root; hipcc cross-compiles without a GPU.
set -eu
ROOT=$(pwd)
make -C ip_avsr_amd/csrc -j4 >/dev/null
hipcc -O2 -std=c++17 --offload-arch=gfx950 profiles/gemm_lab.cpp -Iip_avsr_amd/csrc -Iinclude \
      -Lip_avsr_amd/csrc -ladenet_hip -Wl,-rpath,'$ORIGIN/../ip_avsr_amd/csrc' -o profiles/gemm_lab
mkdir -p profiles/stamps/obj
for f in gemm_f32 gemm_bf16 gemm_x3f gemm_skinny compact elementwise lstm lstm_persistent lstm_cluster prep batch batchnorm convae rbm model; do
  src=ip_avsr_amd/csrc/$f.hip; obj=profiles/stamps/obj/$f.o
  if [ "$f" = gemm_bf16 ] || [ "$f" = gemm_x3f ] || [ ! -f $obj ] || [ $src -nt $obj ]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DADN_GEMM_STAMPS -Iinclude -c $src -o $obj &
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/stamps/libadenet_hip.so profiles/stamps/obj/*.o
hipcc -O2 -std=c++17 --offload-arch=gfx950 -DADN_GEMM_STAMPS profiles/gemm_lab.cpp -Iip_avsr_amd/csrc -Iinclude \
      -Lprofiles/stamps -ladenet_hip -Wl,-rpath,'$ORIGIN/stamps' -o profiles/gemm_lab_stamps
echo built
