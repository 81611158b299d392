cd $GRAFT_REPO_ROOT
o=gpurun_out/ring_ab.txt; : > $o
echo "== ring kernel" >> $o; DN_TF_RING=1 timeout -k 10 300 python tools/sweep_exp.py --wgs 32 --ab 6 >> $o 2>&1 || { cat $o; exit 1; }
echo "== register-staged kernel" >> $o; DN_TF_RING=0 timeout -k 10 300 python tools/sweep_exp.py --wgs 64 --ab 6 >> $o 2>&1 || { cat $o; exit 1; }
grep -q "Memory access fault" $o && { cat $o; exit 1; }
cat $o
