#!/bin/bash
# session 1 of round 4: side-stream overlap A/B, FNO GPU tests, quick bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s1
python tools/exp/ab_step.py "side_stream=0" "" --rounds 4 > gpurun_out/s1/ab_side.txt 2>&1
python tools/exp/ab_step.py "side_stream=0" "" --rounds 3 --graph > gpurun_out/s1/ab_side_graph.txt 2>&1
cat gpurun_out/s1/ab_side.txt gpurun_out/s1/ab_side_graph.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s1/pytest_gpu.log
python bench.py --no-extra --no-cpu-baseline --no-rollout > gpurun_out/s1/bench_quick.json 2> gpurun_out/s1/bench_quick.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s1/trace -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-rollout --no-extra > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/s1/trace.err; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
for f in $(find gpurun_out/s1/trace -name "*kernel_stats.csv" | head -1); do cp $f gpurun_out/s1/step_kernel_stats.csv; done
# keep the kernel trace of the last 2 steps only (start/end timestamps: does the side stream overlap?)
for f in $(find gpurun_out/s1/trace -name "*kernel_trace.csv" | head -1); do tail -100 $f | cut -d, -f8-12,14-16 > gpurun_out/s1/trace_tail.csv; head -1 $f > gpurun_out/s1/trace_head.csv; done
find gpurun_out/s1 -name "*.csv" -size +2M -delete
