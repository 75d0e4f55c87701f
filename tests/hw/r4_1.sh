cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spans" > gpurun_out/r4_1_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r4_1_tests.txt
timeout 600 python tests/hw/span_sweep.py > gpurun_out/r4_1_sweep.txt 2>&1
timeout 300 python tests/hw/prof_spans.py 60 > gpurun_out/r4_1_prof.txt 2>&1
SNAPMI_TESTING=1 SNAPMI_COMPRESS=waves timeout 600 python bench.py --steps 3 --warmup 1 --no-extras --no-pmc --no-cpu > gpurun_out/r4_1_bench_spans.txt 2>&1
tail -3 gpurun_out/r4_1_tests.txt; tail -25 gpurun_out/r4_1_sweep.txt
