#!/bin/bash
# Diagnostic builds of the sweep translation unit with extra -D flags (never shipped):
#   tools/build_variant.sh NAME "-DBODYFIT_SKIN_DEFER=2 ..."  ->  3dbodyanimation_amd/libbodyfit_NAME.so  (BODYFIT_LIB selects it)
# A NAME that starts with "st_" is built with the in-kernel stamps (BODYFIT_STAMPS; all translation units).
set -e
cd "$(dirname "$0")/../3dbodyanimation_amd/csrc"
name=$1; flags=$2
HIPCC=/opt/rocm/bin/hipcc
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=14"   # (the Makefile's HIPFLAGS)
mkdir -p _obj_var/$name
if [[ $name == st_* ]]; then
  for f in bodyfit_api k_sweep k_reduce k_lm_batched k_window_lm overlay; do
    $HIPCC $BASE -DBODYFIT_STAMPS $flags -c $f.hip -o _obj_var/$name/$f.o &
  done
  wait
  $HIPCC -O3 -mavx2 -mfma -std=c++17 -fPIC -c host_solver.cpp -o _obj_var/$name/host_solver.o
  $HIPCC -shared -fPIC --offload-arch=gfx950 -o ../libbodyfit_$name.so _obj_var/$name/*.o -lpthread
else
  make -s >/dev/null
  $HIPCC $BASE $flags -c k_sweep.hip -o _obj_var/$name/k_sweep.o
  objs=$(ls _obj/*.o | grep -v k_sweep.o)
  $HIPCC -shared -fPIC --offload-arch=gfx950 -o ../libbodyfit_$name.so _obj_var/$name/k_sweep.o $objs -lpthread
fi
echo built libbodyfit_$name.so
