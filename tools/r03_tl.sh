cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
echo "== fused"; python tools/probes/step_timeline.py 2>/dev/null | tee gpurun_out/r03/timeline_fused.txt
echo "== MSDE_FUSE_GIN=0"; MSDE_FUSE_GIN=0 python tools/probes/step_timeline.py 2>/dev/null | tee gpurun_out/r03/timeline_unfused.txt
