cd $GRAFT_REPO_ROOT
python tools/probes/step_timeline.py --bucket 2>/dev/null | tail -40 > gpurun_out/tl_mol.txt
sed -i 's/^MOL_KERNEL = True/MOL_KERNEL = False/' moleculesde_amd/geom3d/sde_2d_to_3d.py
python tools/probes/step_timeline.py --bucket 2>/dev/null | tail -40 > gpurun_out/tl_ops.txt
paste gpurun_out/tl_mol.txt gpurun_out/tl_ops.txt | cut -c1-150
