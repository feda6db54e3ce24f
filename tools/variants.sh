#!/bin/bash
# tools/variants.sh NAME "EXTRA_HIPCC_FLAGS"
# Build a tuning/ablation variant of libecrad_hip.so into build_variants/NAME/ (git-ignored, but it
# travels to the GPU box; ECRAD_VARIANT_DIR puts it elsewhere, e.g. tests/_build/variants for the variants the tests load).  Time it with:  ECRAD_HIP_LIB=build_variants/NAME/libecrad_hip.so python bench.py ...
set -e
name=$1; shift
extra="$*"
root=$(cd "$(dirname "$0")/.." && pwd)
out=${ECRAD_VARIANT_DIR:-$root/build_variants}/$name
mkdir -p $out
src="pool setup pipeline abi comm kernel_ica_sw kernel_ica_lw kernel_ica_lw_clear kernel_lw_scat kernel_tc kernel_prep kernel_optics kernel_rrtmg kernel_spartacus kernel_spartacus_lw kernel_ica_sw_exact kernel_tc_sw_exact"
for f in $src; do
  x=""; [ $f = kernel_ica_lw_clear ] && x="-mllvm -amdgpu-sched-strategy=max-memory-clause"      # (as ecrad_amd/csrc/Makefile: EXTRA_kernel_ica_lw_clear)
  # (EXTRA_kernel_spartacus; a variant asking for ..FAST_DIV=0 -- nopack -- gets the correctly rounded float division as well)
  [[ $f = kernel_spartacus* ]] && [[ "$extra" != *FAST_DIV=0* ]] && x="-fno-hip-fp32-correctly-rounded-divide-sqrt -DECRAD_SP_FAST_DIV=1"
  [ $f = kernel_spartacus ] && x="$x -fno-slp-vectorize"      # (the longwave flux sweep, kernel_spartacus_lw, keeps the SLP vectoriser)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $x $extra -c $root/ecrad_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libecrad_hip.so $out/*.o -ldl
rm -f $out/*.o
echo "built $out/libecrad_hip.so"
