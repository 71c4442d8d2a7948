#!/bin/bash
# profiles of a round: bench lines, rocprofv3 kernel stats and PMC HBM traffic per workload, SQ counters of C3.
#   bash tools/gpu_profiles.sh r05     -> gpurun_out/r05/; tools/summarise_profiles.py r05 turns it into profiles/r05_*.
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
for w in C3 C2 C4 C5; do
  timeout 900 python bench.py --steps 20 --warmup 3 --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "bench $w rc=$?"
done
export TMPDIR=/tmp
cd /tmp
for w in C3 C2 C4 C5; do
  steps=3; [ $w = C5 ] && steps=1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $w > $OUT/stats_$w.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$w -- python3 $ROOT/bench.py --steps $steps --warmup 1 --no-cpu-baseline --workload $w > $OUT/fetch_$w.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$w -- python3 $ROOT/bench.py --steps $steps --warmup 1 --no-cpu-baseline --workload $w > $OUT/write_$w.log 2>&1
  echo "profiled $w"
done
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/sq1_C3 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq1_C3.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq2_C3 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq2_C3.log 2>&1
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/sq3_C3 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq3_C3.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq1_C4 -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload C4 > $OUT/sq1_C4.log 2>&1
cd $ROOT
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/ubench_wstream tools/ubench_wstream.hip && timeout 120 /tmp/ubench_wstream > $OUT/ubench_wstream.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/ubench_wform tools/ubench_wform.hip && timeout 120 /tmp/ubench_wform > $OUT/ubench_wform.txt 2>&1
# round 4: the similarity kernel by shape (tall alignments: a workgroup per column), the batch engine on small alignments
timeout 600 python tools/sim_shapes.py > $OUT/sim_shapes.jsonl 2>/dev/null
MSA_LG_SPLIT=1 CHECK=0 REPS=2 timeout 600 python tools/sim_shapes.py 20000 500 3 40000 300 4 5000 1000 9 > $OUT/sim_shapes_one_wave_per_column.jsonl 2>/dev/null
for shape in "1024 100 1000" "1024 60 600" "4096 40 300" "512 128 2000" "256 300 1200" "128 500 2000"; do
  timeout 300 python tools/small_batch.py $shape 2>/dev/null; MSA_BATCH_ENGINE=0 timeout 300 python tools/small_batch.py $shape 2>/dev/null
done > $OUT/small_batch.jsonl
MSA_BATCH_COLS_MAX=0 timeout 300 python tools/small_batch.py 1024 100 1000 2>/dev/null >> $OUT/small_batch.jsonl
MSA_BATCH_ENGINE_MAX=1e12 timeout 300 python tools/c5_engine.py 2>/dev/null > $OUT/c5_engine.jsonl
MSA_BATCH_ENGINE=0 timeout 300 python tools/c5_engine.py 2>/dev/null >> $OUT/c5_engine.jsonl
timeout 300 python tools/c5_counts.py > $OUT/c5_counts.jsonl 2>/dev/null
timeout 300 python tools/small_latency.py > $OUT/small_latency.jsonl 2>/dev/null
# round 4, late: the compact pipeline of small alignments against the ordinary launch sequence, the flat similarity kernel by size,
# and the kernels of two small trims as the profiler sees them
MSA_COMPACT=0 MSA_ZEROCOPY_KB=0 timeout 300 python tools/small_latency.py > $OUT/small_latency_ordinary_launch_sequence.jsonl 2>/dev/null
timeout 300 python tools/flat_sweep.py > $OUT/flat_sweep.jsonl 2>/dev/null
timeout 300 python tests/measure/fixtures_time.py > $OUT/fixtures_time.jsonl 2>/dev/null
( cd /tmp; : > $OUT/small_kernel_stats.txt
  for a in "46 1181 strict" "100 1000 automated1" "209 1227 strictplus" "500 2000 strict" "209 1227 overlap" "209 1227 representative" "1000 4000 automated1" "600 2500 automated1"; do
    rm -rf /tmp/small_prof
    timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/small_prof -- python3 $ROOT/tools/small_one.py $a 200 > /tmp/small_prof.log 2>&1
    grep "per upload" /tmp/small_prof.log >> $OUT/small_kernel_stats.txt
    f=$(find /tmp/small_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 $f | cut -d, -f1-4 | cut -c1-160 >> $OUT/small_kernel_stats.txt
  done )
timeout 120 python tools/cold_upload.py > $OUT/cold_upload.jsonl 2>/dev/null
for sw in MSA_PIPELINE=1,0 MSA_UPLOAD_DIRECT=1,0; do timeout 300 python tools/step_overheads.py C3 C2 C4 C5 --switch $sw 2>/dev/null | grep "ms/step"; done > $OUT/ab_switches.txt
timeout 120 python tools/bx_stamps.py 2>/dev/null | grep sim_ms > $OUT/bx_stamps.jsonl
timeout 120 python tools/bx_stamps.py 1000 4000 2000 2>/dev/null | grep sim_ms >> $OUT/bx_stamps.jsonl
timeout 300 python bench.py --workload REF --out $OUT/reference_shape.jsonl > $OUT/bench_REF.json 2> $OUT/bench_REF.err
timeout 300 python tools/c5_batch.py 1 2 4 6 8 > $OUT/c5_batch.jsonl 2>/dev/null
timeout 300 python tools/upload_time.py > $OUT/upload.txt 2>/dev/null
timeout 600 python tools/sim_by_data.py > $OUT/sim_by_data.jsonl 2>/dev/null
bash tools/gpu_c5_timeline.sh 4 > $OUT/c5_timeline.txt 2>/dev/null
bash tools/gpu_timeline.sh > /dev/null 2>&1
for w in C3 C2 C4; do cp $ROOT/gpurun_out/tl/timeline_$w.txt $OUT/ 2>/dev/null; done
# round 5: the similarity kernel's counters at 1000 x 4000 and 2000 x 10000 (VALU / scalar / texture busy, L2 hit rate), its stamps on the
# reference's ENOG fixture beside a synthetic alignment of that shape, kernels of different contexts side by side
{ echo "== 1000 x 4000"; bash tools/pmc_kernel.sh similarity_lg python3 tools/sim_once.py 1000 4000 2000; echo "== 2000 x 10000"; bash tools/pmc_kernel.sh similarity_lg python3 tools/sim_once.py 2000 10000 1003; } > $OUT/pmc_sim.txt 2>&1
timeout 120 python tools/sim_fixture_stamps.py > $OUT/sim_fixture_stamps.txt 2>/dev/null
timeout 300 python tools/sim_overlap.py > $OUT/sim_overlap.jsonl 2>/dev/null
RESIDENT=1 timeout 300 python tools/sim_overlap.py >> $OUT/sim_overlap.jsonl 2>/dev/null   # (W stays: a pass is the layout kernels + the similarity kernel)
# round 6: the collective path at one rank (RCCL), the front kernel / pair tiles by variant, the engine by trimmer kind, the strip loop
timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/c5_collective.py 15 > $OUT/c5_collective.jsonl 2> $OUT/c5_collective.err
SHAPES=1000x4000,600x2500 timeout 600 python tools/front_pairs_ab.py time > $OUT/front_pairs_ab.jsonl 2>/dev/null
for meth in gappyout overlap representative noduplicateseqs strict; do
  timeout 300 python tools/small_batch.py 1024 100 1000 $meth 2>/dev/null; MSA_BATCH_ENGINE=0 timeout 300 python tools/small_batch.py 1024 100 1000 $meth 2>/dev/null
done > $OUT/small_batch_engine_kinds.jsonl
( bash tools/ubench_lstrip.sh > /dev/null 2>&1 && timeout 300 tools/ubench_lstrip > $OUT/ubench_lstrip.txt 2>&1 )
ls $OUT | head -60
# keep only the small csv files (the merge back is limited to 64 MiB)
find $OUT -name "*kernel_trace.csv" -size +4M -delete
du -sh $OUT
