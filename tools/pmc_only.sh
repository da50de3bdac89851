set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/fin
mkdir -p $O
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.txt 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.txt 2>&1
python tools/pmc_aggregate.py $O/pmc_fetch $O/pmc_write $O/conv_hbm_traffic.json > $O/pmc_agg.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
