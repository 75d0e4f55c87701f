#!/bin/bash
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-extras 2>&1 | grep "kernel ms per step\|snapmi:" | cut -c1-300
