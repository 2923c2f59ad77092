#!/bin/bash
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD
python scripts/gpu_dbg1.py 2>&1 | tail -15
GGML_CDNA4_SPLIT_SELF=8 ./oracle/_ref/split_harness ggml_amd/lib/libggml-cdna4.so q4_K 4096 4096 512 2>&1 | tail -3
GGML_CDNA4_SPLIT_SELF=8 ./oracle/_ref/split_harness ggml_amd/lib/libggml-cdna4.so q4_K 4096 4096 512 2>&1 | tail -1
GGML_CDNA4_SPLIT_SELF=2 ./oracle/_ref/split_harness ggml_amd/lib/libggml-cdna4.so q4_K 4096 4096 512 2>&1 | tail -1
