cd $GRAFT_REPO_ROOT
o=gpurun_out/mlp_wgrad_rows.txt; : > $o
for r in 8192 16384 32768 65536 126976 253952 507904 1015808; do timeout -k 10 200 python tools/mlp_wgrad_exp.py --rows $r 2>&1 | grep "dense\|fault\|rror" >> $o; done
grep -q "Memory access fault" $o && { cat $o; exit 1; }
cat $o
