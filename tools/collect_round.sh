# GPU box: everything profiles/ holds for a round, in one call (tag r06 below): tools/collect_profiles.py (kernel statistics of the bench
# run, of the replayed step and of the roofline leg; per-launch HBM traffic; LDS conflicts; the config 3 / 4 / proteins runs; the bench
# line; the traffic record bench.py quotes) + the kernel list of one fresh-batch index build per workload.  Writes gpurun_out/profiles/;
# copy r06_* from there into profiles/ afterwards.     usage: gpurun -- 'bash tools/collect_round.sh'
python tools/collect_profiles.py --tag r06 --out gpurun_out/profiles > gpurun_out/r6_collect.log 2>&1; echo rc $?; tail -3 gpurun_out/r6_collect.log | cut -c1-300
(echo "# r06 -- kernels of ONE fresh-batch index build (tools/index_kernels.sh), last of four builds"; echo "## proteins (16,384 PROTEINS-shaped graphs, N = 645,196, E = 3,596,222, R = 16)"; bash tools/index_kernels.sh proteins; echo; echo "## config5"; bash tools/index_kernels.sh config5; echo; echo "## config3"; bash tools/index_kernels.sh config3) > gpurun_out/profiles/r06_index_build_kernels.txt 2>&1
ls gpurun_out/profiles | grep r06
