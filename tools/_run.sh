mkdir -p gpurun_out/s5k; O=gpurun_out/s5k
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --config gan --no-cpu-baseline > $O/b_gan.json 2> $O/b_gan.err; cut -c1-250 $O/b_gan.json
python bench.py --no-cpu-baseline --steps 30 > $O/b.json 2> $O/b.err; cut -c1-250 $O/b.json
