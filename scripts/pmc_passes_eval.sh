# rocprofv3 counter passes of the config-4 evaluator (GPU box); one counter group per pass
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
OUT=${1:-gpurun_out/pmc_eval}
mkdir -p $OUT
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/evaluator_probe.py > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log; }
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/evaluator_probe.py > $OUT/trace.log 2>&1
find $OUT -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
