#!/bin/bash
# factor / solve times with each experimental library in exp_libs/ and with the product library, alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/sif-xco2-cokriging_amd/libcokrige_hip.so
cp $L /tmp/product.so
for rep in 1 2; do
  for f in $ROOT/exp_libs/*.so /tmp/product.so; do
    cp $f $L
    echo "== $(basename $f)"
    timeout -k 10 200 python $ROOT/scripts/ab_panel.py ${1:-20000} 2 2>/dev/null | head -2 | tail -1
  done
done
cp /tmp/product.so $L
