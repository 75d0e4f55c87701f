#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "fuzz or lane_order" 2>&1 | tail -6
