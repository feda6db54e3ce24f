#!/bin/bash
# oracle/build_ref_rrtm.sh -- TEST INFRASTRUCTURE.
# Compiles the reference's OWN RRTMG gas-optics routines (ifsrrtm/ + the few ifsaux modules they use),
# unmodified and from where they lie under $REF, plus oracle/ref_rrtm_wrappers.F90, into
# oracle/_ref/libecrad_refrrtm.so.  Nothing is copied into the repo; no stand-ins: these routines have no
# netCDF dependency (their tables come from the big-endian files RADRRTM / RADSRTM).
# The order of the ~180 files is found by retrying: a file that needs a module not yet built fails and
# is tried again in the next pass.
set -u
REF=${REF:-/root/reference}
FC=${FC:-/opt/rocm/bin/amdflang}
here=$(cd "$(dirname "$0")" && pwd)
out=$here/_ref/rrtm
mkdir -p "$out" && cd "$out" || exit 1
flags="-O1 -fPIC -cpp -I$REF/include -I$REF/ifsaux -I$REF/ifsrrtm"
files="$REF/ifsaux/parkind1.F90 $REF/drhook/yomhook_dummy.F90 $REF/ifsaux/yomlun_ecrad.F90 $REF/ifsaux/abor1.F90 \
       $REF/ifsaux/yomcst_ecrad.F90 $REF/ifsaux/yomdyncore.F90 $REF/ifsaux/yommp0_ifsaux.F90 $REF/ifsaux/yomtag.F90 \
       $REF/ifsaux/mpl_module.F90 $(ls $REF/ifsrrtm/*.F90)"
todo="$files"
for pass in 1 2 3 4 5 6 7 8 9 10; do
  next=""
  for f in $todo; do
    o=$(basename "$f" .F90).o
    if ! $FC $flags -c "$f" -o "$o" > "$o.log" 2>&1; then next="$next $f"; fi
  done
  n=$(echo $next | wc -w)
  echo "pass $pass: $n file(s) left"
  [ "$n" -eq 0 ] && break
  [ "$next" == "$todo" ] && { echo "no progress:"; for f in $next; do echo "  $f"; tail -3 "$(basename "$f" .F90).o.log"; done; exit 1; }
  todo="$next"
done
if [ -f "$here/ref_rrtm_wrappers.F90" ]; then
  $FC $flags -c "$here/ref_rrtm_wrappers.F90" -o ref_rrtm_wrappers.o || exit 1
fi
$FC -shared -o "$here/_ref/libecrad_refrrtm.so" *.o && echo "built $here/_ref/libecrad_refrrtm.so"
