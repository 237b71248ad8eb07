# rocprofv3 counter passes of the training step (GPU box).  One counter group per pass, no trace domains beside --pmc.
# usage: bash scripts/pmc_passes.sh OUTDIR [bench args]
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
OUT=${1:-gpurun_out/pmc_r2}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5}
mkdir -p $OUT
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1; tail -c 300 $OUT/$name.log | head -c 200; echo; }
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5 > $OUT/trace.log 2>&1
find $OUT -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -path "*trace*" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
ls $OUT
