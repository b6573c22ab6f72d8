cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O
cat > /tmp/cfg.json <<'JSON'
{"seed": [3, 4], "total_timesteps": 3000000, "eval_freq": 10000, "save_dir": "/tmp/train_out"}
JSON
( time timeout 900 python -m distributional_rl_navigation_amd.train_iqn -C /tmp/cfg.json --n-envs 4096 -P 2 --total-grad-steps 9400 --grad-steps 4 --n-evals 2 ) > $O/numprocs.log 2>&1
tail -12 $O/numprocs.log | cut -c1-200
find /tmp/train_out -name "*.npz" -o -name "network_params.pth" | sort | head -12
