#!/bin/bash
# usage: profiles/scripts/mkvariant.sh NAME "-DFOO=1 -DBAR=2" [SRC_ROOT]   -> scratch/v/libsph_NAME.so
# SRC_ROOT: another checkout of the repo to build from (e.g. `git worktree add /tmp/wt HEAD` for an A/B against HEAD)
set -e
cd /root/repo
N=$1; FLAGS=$2; SRC=${3:-/root/repo}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -fno-slp-vectorize -I$SRC/include $FLAGS"
T=/tmp/var_$N; mkdir -p $T scratch/v
for f in sph_capi sph_sort sph_pairs sph_halo sph_slab sph_compat; do
  /opt/rocm/bin/hipcc $FL -x hip -c $SRC/gpufluidsimulator_amd/csrc/$f.hip -o $T/$f.o &
done
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -I$SRC/include -x c++ -c $SRC/gpufluidsimulator_amd/csrc/particleSystem.cpp -o $T/ps.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o scratch/v/libsph_$N.so $T/*.o
echo built scratch/v/libsph_$N.so
