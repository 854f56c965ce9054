cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --no_cpu_baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], 'bwd_w us', d['roofline']['avg_launch_us'], d['roofline'].get('workgroups'), 'frac', d['roofline']['frac'], 'fwd us', d['roofline_cfconv_fused_fwd']['avg_launch_us'])"; }
run A=1
run MSDE_CFBWD_PIPE=0
run MSDE_SIDE_CFBWD_WGS=192
run MSDE_SIDE_CFBWD_WGS=256
run MSDE_SIDE_CFBWD_WGS=96
