#!/bin/bash
# same-box timing of conv layers under variant libraries: scripts/exp_conv.sh "lib1 lib2 ..." "layer:kind ..."
for lk in $2; do
  li=${lk%%:*}; kind=${lk#*:}
  for lib in $1; do
    echo -n "$lib "; PYTHONPATH=$PWD VQA_HIP_LIB=$PWD/$lib timeout -k 10 100 python scripts/conv_micro.py $li $kind 20 2>&1 | tail -1 || exit 1
  done
done
