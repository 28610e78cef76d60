#!/bin/bash
# usage: profiles/scripts/mkvariant.sh NAME "-DFOO=1 -DBAR=2"   -> scratch/v/libsph_NAME.so
set -e
cd /root/repo
N=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -fno-slp-vectorize -Iinclude $*"
T=/tmp/var_$N; mkdir -p $T scratch/v
for f in sph_capi sph_sort sph_pairs sph_halo sph_slab sph_compat; do
  /opt/rocm/bin/hipcc $FL -x hip -c gpufluidsimulator_amd/csrc/$f.hip -o $T/$f.o &
done
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -Iinclude -x c++ -c gpufluidsimulator_amd/csrc/particleSystem.cpp -o $T/ps.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o scratch/v/libsph_$N.so $T/*.o
echo built scratch/v/libsph_$N.so
