#!/bin/bash
# srgpt_gemm per shape between library builds (and knob settings of the tuning build), same box:
#   scripts/experiments/ab_gemm_libs.sh "lib.so[:KNOB=v,KNOB=v] ..."   shapes = the prefill / ViT / extractor list below
SH="qkv:259:6144:4096 o:259:4096:4096 gateup:259:28672:4096 down:259:4096:14336 vqkv:1458:3456:1152 vout:1458:1152:1152 vfc1:1458:4304:1152 vfc2:1458:1152:4304 proj1:196:4096:4608 dc1:729:4608:1152 dc2:2916:4608:1152 vqkv16:11664:3456:1152 vfc1_16:11664:4304:1152 q4:1036:6144:4096 gu4:1036:28672:4096"
for rep in 1 2; do
for spec in "$@"; do
  lib=${spec%%:*}; kn=""; [ "$spec" != "$lib" ] && kn=$(echo ${spec#*:} | tr ',' ' ')
  echo "== [$lib $kn] rep $rep"
  env $kn SRGPT_LIB=$lib timeout 300 python scripts/experiments/ubench_gemm.py $SH 2>&1 | grep -v amdgpu.ids
done; done
