cd $GRAFT_REPO_ROOT
o=gpurun_out/fork.txt; : > $o
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -8 >> $o
grep -q "Memory access fault" $o && { cat $o; exit 1; }
for f in 1 0 1 0; do echo "== DN_TAIL_FORK=$f" >> $o; DN_TAIL_FORK=$f timeout -k 10 400 python tools/sweep_exp.py --wgs 32 --ab 2 2>&1 | grep "conv leg" >> $o; done
cat $o
