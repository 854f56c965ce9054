#!/bin/bash
# The documented A/B switches must all leave a correct step: the model / plan parity tests under each non-default setting.
mkdir -p gpurun_out/sweep
run() { tag=$1; shift; env "$@" timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_plan.py -q -x > gpurun_out/sweep/$tag.log 2>&1; echo "$tag: $(tail -n 1 gpurun_out/sweep/$tag.log)"; }
run chain MSDE_SCHNET_CHAIN=1
run nopair MSDE_CFCONV_PAIR=0
run nofusegin MSDE_FUSE_GIN=0
run noapply MSDE_FUSE_GIN_APPLY=0
run noframe MSDE_FUSE_FRAME=0 MSDE_FUSE_PAIR_LINEAR=0 MSDE_NOISE_IN_KERNEL=0
run clside MSDE_CL_ON_SIDE=1 MSDE_SIDE_WGRAD=1
run nosplit MSDE_DENSE_SPLIT=0 MSDE_GEOMETRY_ON_SIDE=0
