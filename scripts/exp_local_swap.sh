#!/bin/bash
# run the local-predictor profile once per experimental library in exp_libs/ (ablations: results wrong on purpose)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/sif-xco2-cokriging_amd/libcokrige_hip.so
cp $L /tmp/product.so
for f in $ROOT/exp_libs/*.so; do
  cp $f $L
  echo "== $(basename $f)"
  $ROOT/scripts/prof_local.sh x 20000 400 | grep "assemble_t"
done
cp /tmp/product.so $L
