#!/bin/bash
# NNEST_STAMP diagnostic build of the quad / solo kernels and the training kernels into tools/ab/lib_STAMP.so (the other objects come from the normal build)
set -e
cd "$(dirname "$0")/../nnest_amd/csrc"
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNNEST_STAMP -c nnest_quad.hip -o /tmp/quad_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNNEST_STAMP -c nnest_solo.hip -o /tmp/solo_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNNEST_STAMP -c nnest_train.hip -o /tmp/train_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_STAMP.so nnest_abi.o nnest_kernels.o /tmp/quad_stamp.o /tmp/solo_stamp.o \
    /tmp/train_stamp.o nnest_spline.o nnest_spline_mh.o nnest_spline_train.o nnest_chol.o nnest_host.o -lpthread
