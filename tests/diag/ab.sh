#!/bin/bash
# one-box A/B of two library builds: bash tests/diag/ab.sh <libA.so> <libB.so> [script=tests/diag/gemm_shapes_ab.py] [rounds=2]
A=$1; B=$2; S=${3:-tests/diag/gemm_shapes_ab.py}; R=${4:-2}
for i in $(seq 1 $R); do
  GROMA_HIP_LIB=$A python $S A:$(basename $A .so) 2>&1 | grep "^\["
  GROMA_HIP_LIB=$B python $S B:$(basename $B .so) 2>&1 | grep "^\["
done
