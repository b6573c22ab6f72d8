# A/B two builds of the library on the SAME GPU (act call at 65 536 envs, alternating): libsp_ablA.so vs libsp_ablB.so
D=distributional_rl_navigation_amd
cp $D/libmarinenav_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in A B; do cp $D/libsp_abl$v.so $D/libmarinenav_hip.so; echo -n "$v "; python scripts/act_micro.py 2 2>/dev/null | grep "n=  65536"; done; done
cp /tmp/lib_keep.so $D/libmarinenav_hip.so
