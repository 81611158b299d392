cd $GRAFT_REPO_ROOT
o=gpurun_out/ring_conv.txt; : > $o
DN_TF_RING=1 timeout -k 10 240 python -m pytest tests/test_gpu_kernels.py -x -q -k "rows_transform" >> $o 2>&1 || { cat $o; exit 1; }
echo "== bench --graphs 4096 ring=1" >> $o
DN_TF_RING=1 timeout -k 10 200 python bench.py --graphs 4096 --no-cpu-baseline --steps 30 --warmup 5 2>&1 | tail -3 | cut -c1-300 >> $o
grep -q "Memory access fault" $o && { cat $o; exit 1; }
echo "== ring" >> $o; DN_TF_RING=1 timeout -k 10 300 python tools/sweep_exp.py --wgs 32 >> $o 2>&1 || { cat $o; exit 1; }
grep -q "Memory access fault" $o && { cat $o; exit 1; }
echo "== old kernel" >> $o; DN_TF_RING=0 timeout -k 10 300 python tools/sweep_exp.py --wgs 64 >> $o 2>&1 || { cat $o; exit 1; }
cat $o
