cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc/p1 -o p1 -- python3 tools/conv_microbench.py l1 > gpurun_out/pmc/p1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc/p2 -o p2 -- python3 tools/conv_microbench.py l1 > gpurun_out/pmc/p2.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d gpurun_out/pmc/p3 -o p3 -- python3 tools/conv_microbench.py l1 > gpurun_out/pmc/p3.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d gpurun_out/pmc/p4 -o p4 -- python3 tools/conv_microbench.py l1 > gpurun_out/pmc/p4.txt 2>&1
ls -R gpurun_out/pmc | head -40; tail -3 gpurun_out/pmc/p1.txt
