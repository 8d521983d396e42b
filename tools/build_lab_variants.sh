#!/bin/bash
# Lab builds of libcosmo_hip.so: psd_polar.hip recompiled with one POLAR_LAB_* macro, linked with the production objects (make -C cosmo.jl_amd/csrc first).
# usage: tools/build_lab_variants.sh NO_MAINLOOP NO_EPILOGUE ...   ->  bench/_lab/libcosmo_hip_<NAME>.so   (timed by tools/batch_product_lab.py / gemm_lab.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/cosmo.jl_amd/csrc; mkdir -p $ROOT/bench/_lab
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DPOLAR_LAB_$v -c $C/psd_polar.hip -o $ROOT/bench/_lab/psd_polar_$v.o
  objs=$(ls $C/*.o | grep -v psd_polar.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o $ROOT/bench/_lab/libcosmo_hip_$v.so $objs $ROOT/bench/_lab/psd_polar_$v.o -ldl
  echo built bench/_lab/libcosmo_hip_$v.so
done
