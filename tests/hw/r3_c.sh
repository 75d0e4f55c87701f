#!/bin/bash
# round 3, call C: instruction counters of the second / third generation decoders (2 GiB batch)
mkdir -p gpurun_out
make -C rust-snappy_amd/csrc profile > /dev/null 2>&1
timeout 250 python tests/hw/prof_decode2.py 60 > gpurun_out/r3_c_prof.txt 2>&1
for k in 2 3; do
  SNAPMI_TESTING=1 SNAPMI_DECODE_KERNEL=$k bash tests/hw/pmc_dec.sh k$k 2 \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
    "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
    "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS" \
    >> gpurun_out/r3_c_pmc.txt 2>&1
done
cat gpurun_out/r3_c_prof.txt gpurun_out/r3_c_pmc.txt
