#!/bin/bash
# usage: exp/build.sh NAME [extra hipcc flags]  -> exp/libsearch_NAME.so (whole library, vq_search.hip rebuilt with the flags)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c vq_vae_amd/csrc/vq_search.hip -o exp/vq_search_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libsearch_$NAME.so exp/vq_search_$NAME.o \
  vq_vae_amd/csrc/vq_filter.o vq_vae_amd/csrc/vq_ema.o vq_vae_amd/csrc/vq_scatter_sorted.o vq_vae_amd/csrc/vq_stats.o vq_vae_amd/csrc/vq_boundary.o vq_vae_amd/csrc/vq_conv_wgrad.o
rm -f exp/vq_search_$NAME.o
echo built exp/libsearch_$NAME.so
