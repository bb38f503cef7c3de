#!/bin/bash
# usage: exp/build_filter.sh NAME [extra hipcc flags] -> exp/libfilter_NAME.so (whole library, vq_filter.hip rebuilt with the flags)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS "$@" -c vq_vae_amd/csrc/vq_filter.hip -o exp/vq_filter_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libfilter_$NAME.so exp/vq_filter_$NAME.o \
  vq_vae_amd/csrc/vq_search.o vq_vae_amd/csrc/vq_ema.o vq_vae_amd/csrc/vq_scatter_sorted.o vq_vae_amd/csrc/vq_stats.o vq_vae_amd/csrc/vq_boundary.o vq_vae_amd/csrc/vq_conv_wgrad.o
rm -f exp/vq_filter_$NAME.o
echo built exp/libfilter_$NAME.so
