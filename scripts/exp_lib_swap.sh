#!/bin/bash
# run scripts/ab_groups.py once per experimental library in exp_libs/ (kernel experiments whose results
# are wrong on purpose -- timing only); restores the product library afterwards
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/sif-xco2-cokriging_amd/libcokrige_hip.so
cp $L /tmp/product.so
for f in $ROOT/exp_libs/*.so; do
  cp $f $L
  echo "== $(basename $f)"
  CK_AB_NOCHECK=1 timeout -k 10 200 python $ROOT/scripts/ab_groups.py 20000 3,7 2>&1 | tail -1
done
cp /tmp/product.so $L
echo "== product"
timeout -k 10 200 python $ROOT/scripts/ab_groups.py 20000 3,7 2>&1 | tail -1
