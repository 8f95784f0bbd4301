mkdir -p gpurun_out/s5o; O=gpurun_out/s5o
python -m pytest tests/test_hip_parity.py -x -q -k "stem" > $O/pytest1.txt 2>&1; tail -2 $O/pytest1.txt
DMC_STEM_FWD=0 python bench.py --no-cpu-baseline --steps 30 2> $O/b.err | cut -c1-220
DMC_STEM_FWD=1 python bench.py --no-cpu-baseline --steps 30 2> $O/b.err | cut -c1-220
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o x -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/b_prof.json 2> $R/$O/prof.err
cd $R; grep -E "stem_fwd" $(find /tmp/prof -name '*kernel_stats.csv' | head -1) | cut -c1-200
