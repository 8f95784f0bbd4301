mkdir -p gpurun_out/s5h; O=gpurun_out/s5h
timeout 300 python bench.py --config i3d --steps 10 --warmup 3 > $O/b_i3d_own.json 2> $O/b_i3d_own.err; cut -c1-300 $O/b_i3d_own.json; tail -3 $O/b_i3d_own.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o x -- python $R/bench.py --config i3d --steps 10 --warmup 3 > $R/$O/b_i3d_prof.json 2> $R/$O/prof.err
cd $R; cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $O/i3d_kernel_stats.csv
head -12 $O/i3d_kernel_stats.csv | cut -c1-150
