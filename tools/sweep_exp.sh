cd $GRAFT_REPO_ROOT
o=gpurun_out/ring_ns9.txt; : > $o
for abl in 0 3 0; do echo "== ring DN_TF_ABL=$abl" >> $o; DN_TF_ABL=$abl timeout -k 10 300 python tools/sweep_exp.py --wgs 32 --only baseline --reps 40 2>&1 | grep "^dir\|fault\|rror" >> $o; done
grep -q "Memory access fault" $o && { cat $o; exit 1; }
timeout -k 10 400 python tools/sweep_exp.py --wgs 32 --ab 4 >> $o 2>&1
grep -q "Memory access fault" $o && { cat $o; exit 1; }
cat $o
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "transform or sweep or ring" 2>&1 | tail -3
