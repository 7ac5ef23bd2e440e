# The GPU-box command list behind profiles/r06_* (run as ONE gpurun call: gpurun --timeout 3000 -- bash tools/r06_evidence.sh);
# afterwards, here: python tools/make_traffic.py gpurun_out/prof_r06 and copy the summaries named in profiles/README.md.
set -u
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q > $O/r06_pytest_gpu.log 2>&1; echo "rc=$?" >> $O/r06_pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_final.json 2> $O/r06_bench_final.err; echo "rc=$?" >> $O/r06_bench_final.err
cp $O/bench_detail_1gpu.json $O/r06_bench_final_detail.json
bash tools/prof_trace.sh r06 > $O/r06_prof_trace.txt 2>&1
bash tools/prof_pmc.sh r06 fetch FETCH_SIZE > $O/r06_pmc_fetch.txt 2>&1
bash tools/prof_pmc.sh r06 write WRITE_SIZE > $O/r06_pmc_write.txt 2>&1
bash tools/prof_pmc.sh r06 sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS > $O/r06_pmc_sq1.txt 2>&1
bash tools/prof_pmc.sh r06 sq2 GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > $O/r06_pmc_sq2.txt 2>&1
bash tools/prof_pmc.sh r06 sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL > $O/r06_pmc_sq3.txt 2>&1
bash tools/prof_pmc.sh r06 tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum > $O/r06_pmc_tcc.txt 2>&1
