#!/bin/bash
# tools/variant_flags.sh NAME FILE FLAGS...: a variant in which kernel file FILE (without .hip) is rebuilt with extra compiler
# FLAGS (e.g. -mllvm -amdgpu-sched-strategy=max-ilp) and NOT the per-file extras of the Makefile; every other object comes from
# the current build in ecrad_amd/csrc.
set -e
name=$1; f=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_variants/$name; mkdir -p "$out"
objs=""
for o in pool setup pipeline abi comm kernel_ica_sw kernel_ica_lw kernel_ica_lw_clear kernel_lw_scat kernel_tc kernel_prep kernel_optics kernel_rrtmg kernel_spartacus kernel_spartacus_lw kernel_ica_sw_exact kernel_tc_sw_exact; do
  if [ $o = $f ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $root/ecrad_amd/csrc/$o.hip -o "$out/$o.o"; objs="$objs $out/$o.o"
  else
    objs="$objs $root/ecrad_amd/csrc/$o.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$out/libecrad_hip.so" $objs -ldl
rm -f "$out/$f.o"
echo "built $out/libecrad_hip.so"
