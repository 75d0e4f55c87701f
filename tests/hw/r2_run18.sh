#!/bin/bash
R=$PWD; O=$R/gpurun_out
bash tests/hw/modes.sh 10 2>&1 | tee $O/r2_modes_after.txt
