#!/bin/bash
# Build the measurement tools of this directory for gfx950 (run from anywhere; binaries stay in-tree
# so that gpurun ships them).  step_v2 compares against the round-1 kernel, whose sources are taken
# from git (commit e7a8169) into _v1/ with the namespace renamed.
set -e
cd "$(dirname "$0")"
mkdir -p _v1
for f in g2048_device.h g2048_kernels.h g2048_kernels.hip g2048_pcg64.h; do
    git show e7a8169:gym-2048_amd/csrc/$f | sed 's/namespace g2048/namespace g2048v1/g; s/g2048::/g2048v1::/g; s/G2048_DEV/G2048V1_DEV/g; s/g2048_perm/g2048v1_perm/g; s/g2048_mulhi/g2048v1_mulhi/g; s/g2048_popc/g2048v1_popc/g; s/g2048_opaque/g2048v1_opaque/g; s/g2048_bfi/g2048v1_bfi/g; s/g2048_any/g2048v1_any/g' > _v1/$f
done
# r4_probe compares against the round-3 kernels (commit 1a3ea8e), every identifier renamed g2048 -> g2048r3
mkdir -p _r3
for f in g2048_device.h g2048_kernels.h g2048_kernels.hip g2048_pcg64.h; do
    git show 1a3ea8e:gym-2048_amd/csrc/$f | sed 's/g2048/g2048r3/g; s/G2048/G2048R3/g' > _r3/$(echo $f | sed 's/g2048/g2048r3/')
done
# r5_probe compares against the round-4 kernels (commit 8baa390), every identifier renamed g2048 -> g2048r4
mkdir -p _r4
for f in g2048_device.h g2048_kernels.h g2048_kernels.hip g2048_pcg64.h; do
    git show 8baa390:gym-2048_amd/csrc/$f | sed 's/g2048/g2048r4/g; s/G2048/G2048R4/g' > _r4/$(echo $f | sed 's/g2048/g2048r4/')
done
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for t in "$@"; do
    $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -o $t $t.hip   # same flags as the product build
done
