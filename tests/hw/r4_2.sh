cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spans" > gpurun_out/r4_2_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r4_2_tests.txt
timeout 600 python tests/hw/span_sweep.py alice29.txt > gpurun_out/r4_2_sweep.txt 2>&1
timeout 300 python tests/hw/prof_spans.py 60 > gpurun_out/r4_2_prof.txt 2>&1
tail -3 gpurun_out/r4_2_tests.txt; grep -v "^{" gpurun_out/r4_2_sweep.txt | tail -25; cat gpurun_out/r4_2_prof.txt
