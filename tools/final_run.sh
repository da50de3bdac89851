# Round-end measurement set on the GPU box (run through gpurun from the repo root); results land in gpurun_out/fin/.
# Everything under profiles/ is copied from here (tools/collect_profiles.py <round>).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/fin
rm -rf $O; mkdir -p $O
# the whole GPU suite twice on this box (no -x: every failure is listed); round 3 went red on a run-to-run difference
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu_run2.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 1200 python bench.py > $O/bench.txt 2>&1
# same box, same code, recurrent encoder in the plain order (one ConvLSTM launch per level and sub-window): the A/B of the skewed schedule
# (alternating short runs: the 100-step line above and a 30-step run are not the same thermal state)
for i in 1 2; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | cut -c1-160 >> $O/bench_no_skew.txt
  timeout 600 python bench.py --steps 30 --warmup 5 --no-skew --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | cut -c1-160 >> $O/bench_no_skew.txt
done
# RCCL call path with ONE rank (the box has one GPU): torch.distributed.run -> nccl process group -> bucketed all-reduce per step
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-extras > $O/bench_torchrun_world1.txt 2>&1
# same box: the product schedule (frozen front of step i+1 under the trainable back of step i) / front streams without the
# cross-step pipeline / everything on one stream -- alternating short runs
for i in 1 2; do for f in "" "--no-pipeline" "--no-overlap-teacher"; do
  echo "frame2voxel_pixel_distill $f" >> $O/bench_schedule_ab.txt
  timeout 600 python bench.py --steps 30 --warmup 5 $f --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | cut -c1-160 >> $O/bench_schedule_ab.txt
done; done
# kernel traces: ONE stream (a launch's duration is the kernel's own: what `roofline` quotes) and the product schedule
for wl in frame2voxel_pixel_distill frame2voxel_full frame2recon_full; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o step -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras --no-overlap-teacher --workload $wl > $O/prof_$wl.txt 2>&1
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipelined -o step -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > $O/prof_pipelined.txt 2>&1
for st in deeplab_fwd maskclip_fwd teacher_fwd; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stage_$st -o p -- python tools/bench_stage.py $st --iters 10 > $O/stage_$st.txt 2>&1
done
for r in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vox_raw$r -o p -- python tools/bench_voxelizer.py --raw $r --iters 20 > $O/vox_raw$r.txt 2>&1
done
# the dominant kernel alone: A/B against the round-5 kernels (interleaved rounds in one process) and its instruction-mix counters
timeout 600 python tools/bench_lstm_group.py --modes 1,0,4 --rounds 3 > $O/lstm_group_ab.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_w128 -o p -- python $GRAFT_REPO_ROOT/tools/bench_lstm_group.py --modes 4 --rounds 1 --iters 5 > /dev/null 2>&1 )
python tools/pmc_parse.py $O/pmc_w128 "conv3x3_[a-z0-9_]*kernel" > $O/pmc_w128.txt 2>&1 || true
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_w128_mfma -o p -- python $GRAFT_REPO_ROOT/tools/bench_lstm_group.py --modes 4 --rounds 1 --iters 5 > /dev/null 2>&1 )
python tools/mfma_util.py $O/pmc_w128_mfma > $O/pmc_w128_mfma.txt 2>&1 || true
timeout 600 python tools/bench_conv1x1.py > $O/conv1x1_ab.txt 2>&1
timeout 600 python tools/bench_conv3x3.py > $O/conv3x3_ab.txt 2>&1
for rep in 1 2; do for g in off on; do
  echo "teacher path cache-policy rules $g" >> $O/bench_cache_policy_ab.txt
  if [ $g = off ]; then OESS_W128_NT=0 OESS_APPLY_NT=0 OESS_BILINEAR_NT=0 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | cut -c1-160 >> $O/bench_cache_policy_ab.txt
  else timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | cut -c1-160 >> $O/bench_cache_policy_ab.txt; fi
done; done
for rep in 1 2; do for g in 1 0; do echo "OESS_W128_GEMM=$g" >> $O/bench_gemm_ab.txt; OESS_W128_GEMM=$g timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pmc --no-extras 2>&1 | tail -1 | cut -c1-160 >> $O/bench_gemm_ab.txt; done; done
{ timeout 400 python tools/bench_host_pools.py --json; timeout 400 python tools/bench_host_pools.py --json --no-copy; timeout 400 python tools/bench_host_pools.py --json --pools 1; } > $O/host_pools.txt 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python bench.py --child --steps 1 --warmup 1 --no-overlap-teacher > $O/pmc_mfma.txt 2>&1
python tools/mfma_util.py $O/pmc_mfma $O/mfma_util.json > $O/mfma_util.txt 2>&1 || true
timeout 600 bash tools/pmc_traffic.sh "python tools/bench_voxelizer.py --raw 1 --iters 5" "tri_sort|tri_splat" > $O/voxelizer_pmc.txt 2>&1
{ timeout 300 python tools/bench_train_loop.py --workers 4; timeout 300 python tools/bench_train_loop.py --workers 2; timeout 300 python tools/bench_train_loop.py --workers 4 --no-pipeline; timeout 300 python tools/bench_train_loop.py --workers 4 --no-prefetch; timeout 300 python tools/bench_train_loop.py --workers 4 --dataloader; timeout 300 python tools/bench_train_loop.py --workers 10 --dataloader; } > $O/train_loop.txt 2>&1
timeout 300 python tools/aten_probe.py frame2recon_full > $O/aten_probe_frame2recon_full.txt 2>&1 || true
timeout 300 python tools/bench_png.py > $O/png.txt 2>&1
timeout 300 python tools/bench_stage.py deeplab_fwd --breakdown > $O/deeplab_breakdown.txt 2>&1
timeout 300 python tools/bench_segmean.py > $O/segmean.txt 2>&1 || true
bash tools/step_sequence.sh > $O/step_sequence.txt 2>&1 || true
cp gpurun_out/seq_last_step.txt $O/seq_last_step.txt 2>/dev/null || true
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*.csv" -size +20M -delete
du -sh $O
