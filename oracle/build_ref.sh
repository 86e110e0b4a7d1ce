#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY (oracle/).
#
# Builds the REAL reference (jmcdaniel43/Reactive_PB_NN_MD, /root/reference/src/*.f90)
# as an oracle, from the sources where they lie, into oracle/_ref/ (git-ignored):
#   oracle/_ref/ref_dump_std   ms_evb_simulation="no"  (reference default, src/glob_v.f90:45)
#   oracle/_ref/ref_dump_evb   ms_evb_simulation="yes" (the reference's own compile-time switch)
# Each binary = reference modules + oracle/ref_driver.f90 (dump driver, build-authored)
#             + oracle/dfti_mkl_adapter.c (Fortran->C calling-convention layer onto the
#               real Intel MKL DFTI that ships inside torch's libtorch_cpu.so).
#
# Nothing of the reference is copied into the repository: sources are compiled from
# /root/reference; the two files that need a one-token edit to be accepted by flang
# are edited in a mktemp scratch dir that is deleted afterwards:
#   md_integration.f90:317  "system_data.n_mole" (DEC dot-member) -> "system_data%n_mole"
#   glob_v.f90:45           ms_evb_simulation="no" -> "yes"   (EVB build only)
# The reference is compiled WITHOUT -fopenmp: its !$OMP lines are then comments, which
# (a) side-steps the stale duplicate directive at pair_int_real_space.f90:183 and
# (b) makes every summation order deterministic.
#
# Does nothing (exit 0) when /root/reference or flang is absent (e.g. on the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF=${REF_SRC:-/root/reference/src}
FLANG=${FLANG:-/opt/rocm/lib/llvm/bin/flang}
OUT="$HERE/_ref"
if [ ! -d "$REF" ] || [ ! -x "$FLANG" ]; then
  echo "build_ref.sh: reference sources or flang not present - skipping"; exit 0
fi
TORCHLIB=$(python3 - <<'EOF'
import importlib.util, os
s = importlib.util.find_spec("torch")
print(os.path.join(os.path.dirname(s.origin), "lib"))
EOF
)
[ -f "$TORCHLIB/libtorch_cpu.so" ] || { echo "libtorch_cpu.so (MKL DFTI provider) not found"; exit 1; }
mkdir -p "$OUT"
ORDER="mkl_dfti glob_v read_simulation_parameters general_routines intra_bonded_interactions pair_int_real_space pme total_energy_forces ms_evb initialize_routines md_integration"
FFLAGS="-O2 -ffp-contract=off"

build_one () {  # $1 = std|evb
  local S; S=$(mktemp -d)
  trap 'rm -rf "$S"' RETURN
  ( cd "$S"
    for f in $ORDER; do
      src="$REF/$f.f90"
      [ "$f" = mkl_dfti ] && src="$REF/mkl_modules/mkl_module_src/mkl_dfti.f90"
      if [ "$f" = md_integration ]; then
        sed 's/system_data\.n_mole/system_data%n_mole/' "$src" > md_integration.f90; src=md_integration.f90
      fi
      if [ "$f" = glob_v ] && [ "$1" = evb ]; then
        sed 's/ms_evb_simulation="no"/ms_evb_simulation="yes"/' "$src" > glob_v.f90; src=glob_v.f90
      fi
      "$FLANG" $FFLAGS -c "$src" -o "$f.o" 2> "$f.log" || { cat "$f.log"; exit 1; }
    done
    "$FLANG" $FFLAGS -c "$HERE/ref_driver.f90" -o ref_driver.o 2> drv.log || { cat drv.log; exit 1; }
    gcc -O2 -c "$HERE/dfti_mkl_adapter.c" -o dfti_mkl_adapter.o
    objs=""; for f in $ORDER; do objs="$objs $f.o"; done
    "$FLANG" ref_driver.o $objs dfti_mkl_adapter.o -L"$TORCHLIB" -ltorch_cpu -lc10 -l:libgomp.so \
        -Wl,-rpath,"$TORCHLIB" -o "$OUT/ref_dump_$1"
  )
  echo "built $OUT/ref_dump_$1"
}
build_one std
build_one evb
