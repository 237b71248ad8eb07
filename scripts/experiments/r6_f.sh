# round 6, call f: what the conv epilogue costs per shape (upper bound of cross-tile epilogue hiding): the serialized per-launch
# table of the production library against a measurement build without any epilogue
mkdir -p gpurun_out/r6_f
LEAN="--no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5"
IEEE_PROFILE_DUMP=gpurun_out/r6_f/base.csv timeout 600 python bench.py --steps 6 --warmup 3 $LEAN > gpurun_out/r6_f/base.json 2> gpurun_out/r6_f/base.err
IEEE_BN_STRICT=0 IEEE_AMD_LIB=$PWD/scripts/experiments/libieee_noepi.so IEEE_PROFILE_DUMP=gpurun_out/r6_f/noepi.csv timeout 600 python bench.py --steps 6 --warmup 3 $LEAN > gpurun_out/r6_f/noepi.json 2> gpurun_out/r6_f/noepi.err
tail -3 gpurun_out/r6_f/noepi.err
python scripts/experiments/r6_noepi_compare.py gpurun_out/r6_f/base.csv gpurun_out/r6_f/noepi.csv > gpurun_out/r6_f/compare.txt 2>&1
cat gpurun_out/r6_f/compare.txt
