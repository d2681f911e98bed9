#!/bin/bash
# NNEST_STAMP diagnostic build of the quad / solo kernels and the training kernels into tools/ab/lib_STAMP.so (the other objects come from the normal build)
set -e
cd "$(dirname "$0")/../nnest_amd/csrc"
mkdir -p ../../tools/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNNEST_STAMP -c nnest_quad.hip -o /tmp/quad_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNNEST_STAMP -c nnest_solo.hip -o /tmp/solo_stamp.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNNEST_STAMP -c nnest_train.hip -o /tmp/train_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_STAMP.so nnest_abi.o nnest_kernels.o /tmp/quad_stamp.o /tmp/solo_stamp.o \
    /tmp/train_stamp.o nnest_spline.o nnest_spline_mh.o nnest_spline_train.o nnest_spline_rows.o nnest_chol.o nnest_host.o -lpthread
# the rows form of the spline training step with its stamps (tools/time_spline_train.py prints them through NNEST_HIP_LIB)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DNNEST_STAMP -c nnest_spline_rows.hip -o /tmp/rows_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_ROWS_STAMP.so nnest_abi.o nnest_kernels.o nnest_quad.o nnest_solo.o \
    nnest_train.o nnest_spline.o nnest_spline_mh.o nnest_spline_train.o /tmp/rows_stamp.o nnest_chol.o nnest_host.o -lpthread
