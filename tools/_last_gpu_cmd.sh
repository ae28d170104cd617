mkdir -p gpurun_out/r3q; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3q/trace -- python3 bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline > gpurun_out/r3q/bench_trace.log 2>&1
cp gpurun_out/r3q/trace/*/*kernel_stats.csv gpurun_out/r3q/kernel_stats.csv
cut -d, -f1-4 gpurun_out/r3q/kernel_stats.csv
python3 -m pytest tests -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; tail -3 gpurun_out/r3q/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
