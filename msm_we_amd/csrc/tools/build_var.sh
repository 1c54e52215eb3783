#!/bin/bash
# build_var.sh NAME "-DFLAGS" : experiment build of train.hip into _var/lib_NAME.so (other objects from _build/)
set -e
cd "$(dirname "$0")/.."
mkdir -p _var
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $F $2 -c train.hip -o _var/train_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o _var/lib_$1.so _build/ctx.o _build/assign.o _build/project.o _build/flux.o _build/flux_sorted.o _var/train_$1.o _build/comm.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
