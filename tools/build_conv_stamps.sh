#!/bin/bash
# Diagnostic build: the convolution kernels with s_memtime phase stamps (-DCTDD_RES_STAMPS) as a library of their own,
# continuous-time-diffusion-models-for-discrete-data_amd/libres_stamps.so (git-ignored; read by tools/stamp_conv.py).
set -e
cd "$(dirname "$0")/../continuous-time-diffusion-models-for-discrete-data_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Wno-unused-function \
  -DCTDD_RES_STAMPS -DCTDD_PATCH_STAMPS -shared unet_kernels.hip misc.hip -o ../libres_stamps.so
