# Round-end measurement set on the GPU box (run through gpurun from the repo root); results land in gpurun_out/fin/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/fin
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 600 python bench.py > $O/bench.txt 2>&1
for wl in frame2voxel_pixel_distill frame2voxel_full frame2recon_full; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o step -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $wl > $O/prof_$wl.txt 2>&1
done
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.txt 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.txt 2>&1
python tools/pmc_aggregate.py $O/pmc_fetch $O/pmc_write $O/conv_hbm_traffic.json > $O/pmc_agg.txt 2>&1
find $O -name "*.csv" -size +20M -delete
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O
