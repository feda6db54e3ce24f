#!/bin/bash
# tools/variant_files.sh NAME "FLAGS" file1 [file2 ...]: a tuning variant in which only the named kernel files are rebuilt
# with FLAGS; every other object comes from the current build in ecrad_amd/csrc (run make there first).
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_variants/$name; mkdir -p $out
objs=""
for o in pool setup pipeline abi comm kernel_ica_sw kernel_ica_lw kernel_ica_lw_clear kernel_lw_scat kernel_tc kernel_prep kernel_optics kernel_rrtmg kernel_spartacus kernel_spartacus_lw kernel_ica_sw_exact kernel_tc_sw_exact; do
  if [[ " $* " == *" $o "* ]]; then
    x=""; [ $o = kernel_ica_lw_clear ] && x="-mllvm -amdgpu-sched-strategy=max-memory-clause"
    [[ $o = kernel_spartacus* ]] && [[ "$flags" != *FAST_DIV=0* ]] && x="-fno-hip-fp32-correctly-rounded-divide-sqrt -DECRAD_SP_FAST_DIV=1"
    [ $o = kernel_spartacus ] && x="$x -fno-slp-vectorize"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $x $flags -c $root/ecrad_amd/csrc/$o.hip -o $out/$o.o &
    objs="$objs $out/$o.o"
  else
    objs="$objs $root/ecrad_amd/csrc/$o.o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libecrad_hip.so $objs -ldl
rm -f $out/*.o
echo "built $out/libecrad_hip.so"
