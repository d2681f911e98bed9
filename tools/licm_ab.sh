#!/bin/bash
# Developer diagnostic (GPU box): does `-mllvm -disable-machine-licm` pay for a translation unit?  The pair form of the spline proposal
# kernel lives on it (nnest_spline_mh.hip, Makefile); this script times the OTHER kernels with and without (profiles/r05/machine_licm_ab.txt:
# solo -3 %, quad +2 %, K5 and spline training unchanged -- they stay as they are).  Build the variants first, in the build container:
#   cd nnest_amd/csrc; for tu in nnest_solo nnest_train nnest_spline_train nnest_quad; do
#     hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -mllvm -disable-machine-licm -c $tu.hip -o /tmp/licm/$tu.o
#     hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/lib_LICM_$tu.so <the objects, /tmp/licm/$tu.o in place of $tu.o> -lpthread; done
echo "== default lib: spline K4"; python tools/time_spline.py 50 1000 2>&1 | grep K4-spline
timeout 900 python -m pytest tests/test_gpu_spline.py -x -q 2>&1 | tail -2
echo "== solo (headline) default vs LICM-off"; python tools/time_quad.py 2>/dev/null | grep -E "^solo fixed|^default|solo batch lag 8 " ; NNEST_HIP_LIB=$PWD/tools/ab/lib_LICM_nnest_solo.so python tools/time_quad.py 2>/dev/null | grep -E "^solo fixed|^default|solo batch lag 8 "
echo "== quad default vs LICM-off";  python tools/time_quad.py 2>/dev/null | grep -E "^quad fixed|quad batch lag 4" ; NNEST_HIP_LIB=$PWD/tools/ab/lib_LICM_nnest_quad.so python tools/time_quad.py 2>/dev/null | grep -E "^quad fixed|quad batch lag 4"
echo "== K5 default vs LICM-off"; python tools/time_train.py 2>/dev/null | head -2; NNEST_HIP_LIB=$PWD/tools/ab/lib_LICM_nnest_train.so python tools/time_train.py 2>/dev/null | head -2
echo "== spline train default vs LICM-off"; python tools/time_spline_train.py 2>/dev/null | tail -2; NNEST_HIP_LIB=$PWD/tools/ab/lib_LICM_nnest_spline_train.so python tools/time_spline_train.py 2>/dev/null | tail -2
