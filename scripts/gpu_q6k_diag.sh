#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
H=oracle/_ref/split_harness; P=ggml_amd/lib/libggml-cdna4.so
run() { echo "--- $*"; timeout 120 "$@" 2>&1 | grep -v load_backend | cut -c1-700 | tail -2; }
run $H $P q6_K 1024 1024 200
run $H $P q6_K 1024 2048 200 shared
GGML_CDNA4_NO_GRAPHS=1 run $H $P q6_K 1024 2048 200 shared
GGML_CDNA4_NO_FUSE=1 run $H $P q6_K 1024 2048 200 shared
run $H $P q6_K 1024 2048 16 shared
run $H $P q6_K 1024 2048 1 shared
run $H $P q6_K 1024 2816 96 shared
run $H $P q6_K 512 2048 200 shared
run $H $P q5_K 1024 2048 200 shared
run $H $P q4_K 1024 2048 200 shared
