cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_reference_suite.py tests/test_gpu_tools.py -x -q -m gpu > gpurun_out/r4_3_tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r4_3_tests.txt
timeout 900 python bench_configs.py --plan sweep:4,adapters:4,pcie:4 > gpurun_out/r4_3_extras.txt 2>&1
tail -4 gpurun_out/r4_3_tests.txt; grep "^{" gpurun_out/r4_3_extras.txt
