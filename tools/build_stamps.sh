#!/bin/bash
# Diagnostic build: the two S = 256 step kernels with s_memtime phase stamps (-DCTDD_S256_STAMPS) as a small library of their
# own, continuous-time-diffusion-models-for-discrete-data_amd/libctdd_stamps.so (git-ignored; read by tools/stamps_s256.py).
set -e
cd "$(dirname "$0")/../continuous-time-diffusion-models-for-discrete-data_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Wno-unused-function \
  -fno-slp-vectorize -DCTDD_S256_STAMPS -shared steps_s256.hip steps_s256_b16.hip misc.hip -o ../libctdd_stamps.so
