cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
echo "== default"; python tools/probes/step_timeline.py --bucket 2>/dev/null | tee gpurun_out/r03/timeline_bucket.txt
