mkdir -p gpurun_out/s5a; O=gpurun_out/s5a
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python bench.py --no-cpu-baseline > $O/b_mio.json 2> $O/b_mio.err
python bench.py --no-cpu-baseline --own-conv 1 --conv-arith 1 > $O/b_x3.json 2> $O/b_x3.err
cut -c1-200 $O/b_mio.json $O/b_x3.json
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o x -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --own-conv 1 --conv-arith 1 > $R/$O/b_prof.json 2> $R/$O/prof.err
cd $R; f=$(find /tmp/prof -name '*kernel_trace.csv' | head -1); python tools/rocprof_region.py $f 20 > $O/region.csv; cp $(find /tmp/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
head -30 $O/region.csv | cut -c1-200
