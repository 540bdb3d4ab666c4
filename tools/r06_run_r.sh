#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r06r
echo "== product build"; timeout -k 10 200 python tools/shiftdiag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06r/normal.txt
echo "== no products"; EBFI_DEV=1 EBFI_LIB_PATH=$PWD/ebfi-be_amd/lib_diag/libebfi_diag1.so timeout -k 10 200 python tools/shiftdiag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06r/noproducts.txt
echo "== no thick loads"; EBFI_DEV=1 EBFI_LIB_PATH=$PWD/ebfi-be_amd/lib_diag/libebfi_diag2.so timeout -k 10 200 python tools/shiftdiag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06r/noloads.txt
