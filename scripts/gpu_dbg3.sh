#!/bin/bash
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD
python scripts/gpu_dbg4.py 2>&1 | grep -v amdgpu.ids | tail -12
