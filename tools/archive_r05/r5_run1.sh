#!/bin/bash
# first GPU run of round 5: the windowless fp16 form
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_planes_variant.py tests/test_gpu_f16_dynamic_range.py -x -q -s > gpurun_out/r5a/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r5a/pytest.txt
tail -40 gpurun_out/r5a/pytest.txt
timeout 600 python tools/conv_layers.py 0.2 > gpurun_out/r5a/layers_product.txt 2>&1
IGAN_LIB=$PWD/inclusivegan_amd/csrc/libigan_hip_noslp.so timeout 600 python tools/conv_layers.py 0.2 > gpurun_out/r5a/layers_noslp.txt 2>&1
tail -5 gpurun_out/r5a/layers_product.txt gpurun_out/r5a/layers_noslp.txt
