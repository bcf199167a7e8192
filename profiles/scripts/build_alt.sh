# builds profiles/alt/libadenet_hip.so with extra compile flags for gemm_bf16.hip ($1, e.g. -DADN_PP_NS=5) and profiles/gemm_lab_alt
# against it: in-process A/B of a compile-time variant against the in-tree library (profiles/gemm_lab)
set -eu
FLAGS=${1:-}
mkdir -p profiles/alt/obj
for f in gemm_f32 gemm_bf16 gemm_x3f gemm_skinny compact elementwise lstm lstm_persistent lstm_cluster prep batch batchnorm convae rbm model; do
  src=ip_avsr_amd/csrc/$f.hip; obj=profiles/alt/obj/$f.o
  if [ "$f" = gemm_bf16 ] || [ "$f" = gemm_x3f ] || [ ! -f $obj ] || [ $src -nt $obj ]; then
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $FLAGS -Iinclude -c $src -o $obj &
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/alt/libadenet_hip.so profiles/alt/obj/*.o
hipcc -O2 -std=c++17 --offload-arch=gfx950 profiles/gemm_lab.cpp -Iip_avsr_amd/csrc -Iinclude \
      -Lprofiles/alt -ladenet_hip -Wl,-rpath,'$ORIGIN/alt' -o profiles/gemm_lab_alt
echo built alt "$FLAGS"
