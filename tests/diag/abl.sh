#!/bin/bash
# main-loop ablations of the 256x256 GEMM on one box (timing only: the ablated builds compute garbage)
for v in abl_base abl_NODMA abl_NOREAD abl_NODMA_NOREAD abl_NOMFMA abl_base; do
  GROMA_HIP_LIB=tests/diag/$v.so python tests/diag/gemm_shapes_ab.py $v 2>&1 | grep "^\[" | grep "qkv\|down\|vit fc1\|weighted"
done
