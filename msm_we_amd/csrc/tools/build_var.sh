#!/bin/bash
# build_var.sh NAME "-DFLAGS" [train|assign] : experiment / profile build of one source into _var/lib_NAME.so (other
# objects from _build/).  Profile builds the tools expect:
#   build_var.sh seedprof   "-DSEED_PROFILE"     train     (tools/seedprof.py)
#   build_var.sh stepprof1  "-DSTEP_PROFILE=1"   train     (tools/stepprof.py; also stepprof2 with =2)
#   build_var.sh assignprof "-DASSIGN_PROFILE"   assign    (tools/assignprof.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p _var
SRC=${3:-train}
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $F $2 -c $SRC.hip -o _var/${SRC}_$1.o
OBJS=""
for o in ctx hostio assign project pca flux flux_sorted flux_rows train comm; do
  if [ "$o" = "$SRC" ]; then OBJS="$OBJS _var/${SRC}_$1.o"; else OBJS="$OBJS _build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o _var/lib_$1.so $OBJS -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
