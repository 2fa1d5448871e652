#!/bin/bash
# Variant builds of the library for kernel experiments: tools/variants.sh tag "MACRO=V MACRO2=V2" [tag2 "..."] ...
# -> mcaller_amd/variants/<tag>.so (cross-compiled here; they travel to the GPU box with the snapshot; git-ignored)
mkdir -p mcaller_amd/variants
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  ( python3 - <<P
from mcaller_amd.build import build_lib
build_lib(out='mcaller_amd/variants/$tag.so', defines=tuple('$defs'.split()))
P
  ) &
done
wait
ls -la mcaller_amd/variants/
