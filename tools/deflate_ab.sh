#!/bin/bash
# A/B of two builds of the DEFLATE kernel on one box: tools/deflate_ab.sh [depths=32,96]  (run from the repo root on the GPU box)
# "head" = snappy_amd/variants/libsnaphash_dfhead.so (built by hand from another revision or with another -D), "new" = the shipped library.
# Prints kernel ms per corpus and depth for both and compares the sha256 of every output: the rewrite must not change a byte.
set -e
D=${1:-32,96}
mkdir -p gpurun_out
SNAPHASH_LIB=$PWD/snappy_amd/variants/libsnaphash_dfhead.so timeout -k 10 400 python3 tools/deflate_corpora.py $D > gpurun_out/df_head.txt 2>&1
timeout -k 10 400 python3 tools/deflate_corpora.py $D > gpurun_out/df_new.txt 2>&1
echo "head:"; grep kernel gpurun_out/df_head.txt | cut -c1-75
echo "new:";  grep kernel gpurun_out/df_new.txt | cut -c1-75
if diff <(grep -o "sha256 of the output [0-9a-f]*" gpurun_out/df_head.txt) <(grep -o "sha256 of the output [0-9a-f]*" gpurun_out/df_new.txt) > /dev/null; then echo "outputs: byte-identical"; else echo "outputs: DIFFER"; exit 1; fi
