#!/bin/bash
bash tests/hw/final_profile.sh r2_v3 2>&1 | tail -12
