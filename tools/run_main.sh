cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gpu_plan.py -x -q -m gpu -k "stream_runner or oversized" 2>&1 | tail -5 > gpurun_out/main_tests.txt
python -m moleculesde_amd.pretrain --model_3d=SchNet --lr=1e-4 --batch_size=256 --gnn_3d_lr_scale=0.1 --dropout_ratio=0 --emb_dim=300 --epochs=3 --steps_per_epoch 200 --SDE_coeff_contrastive=1 --CL_similarity_metric=EBM_node_dot_prod --T=0.1 --normalize --SDE_coeff_generative_2Dto3D=1 --SDE_2Dto3D_model=SDEModel2Dto3D_02 --SDE_type_2Dto3D=VE --use_extend_graph --SDE_coeff_generative_3Dto2D=0 > gpurun_out/pretrain_main.log 2>&1
python bench.py --no_cpu_baseline --no_configs45 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
