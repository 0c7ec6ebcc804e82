#!/bin/bash
# Build an A/B variant of libptr_hip.so with extra -D flags: tools/build_variant.sh <name> <flags...>
# Output: variants/libptr_<name>.so (select at run time with PTR_HIP_LIBRARY=...).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
CS=$ROOT/metal-pathtracer-arm64_amd/csrc
mkdir -p $ROOT/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I$ROOT/include -I$CS/host -I$CS/kernels "$@" \
  -c $CS/kernels/wavefront.hip -o /tmp/wavefront_$NAME.o
# (the backend sees the same layout switches as the kernels: PTR_POOL_AOS, ...)
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -I$ROOT/include -I$CS/host -I$CS/kernels "$@" -c $CS/host/hip_backend.cpp -o /tmp/hip_backend_$NAME.o
HOST_OBJS=$(ls $CS/host/*.o | grep -v hip_backend.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/variants/libptr_$NAME.so $HOST_OBJS /tmp/hip_backend_$NAME.o /tmp/wavefront_$NAME.o -pthread
echo built $ROOT/variants/libptr_$NAME.so
