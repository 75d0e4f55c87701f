#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_frame.py -m gpu -q -p no:cacheprovider -x -k "larger_than" 2>&1 | tail -15
