cd /root/repo
mkdir -p gpurun_out
R=$PWD
timeout 330 python -m pytest tests -x -q -m gpu < /dev/null > gpurun_out/gpu_tests_r1l.log 2>&1; tail -3 gpurun_out/gpu_tests_r1l.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" < /dev/null 2>&1 | tail -1
timeout 150 python bench.py < /dev/null > gpurun_out/bench_full_r1l.log 2>&1; tail -1 gpurun_out/bench_full_r1l.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1l -o r1l -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline < /dev/null > $R/gpurun_out/prof_r1l.log 2>&1
cd $R; ls gpurun_out/prof_r1l | head
