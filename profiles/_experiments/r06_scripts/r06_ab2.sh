#!/bin/bash
# round 6, second A/B: check of every front / pair variant, the narrow front kernel below 513 sequences, sixteen-row pair tiles at C3,
# the identity statistics with 1024 terms in flight, the collective path
export MSA_DIAGNOSTICS=1  # (the library reads its MSA_* diagnostic switches only under this one)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_ab2
mkdir -p $OUT; cd $ROOT
timeout 900 python tools/front_pairs_ab.py check > $OUT/check.txt 2>&1; echo "check rc=$?"; tail -3 $OUT/check.txt
SHAPES=1000x4000 timeout 600 python tools/front_pairs_ab.py time > $OUT/time.jsonl 2>$OUT/time.err; echo "time rc=$?"
AB_SMALL=1 SHAPES=150x1200,209x1227,300x1200,500x2000,512x6000 timeout 900 python tools/front_pairs_ab.py time > $OUT/time_small.jsonl 2>$OUT/time_small.err; echo "time small rc=$?"
for rep in 1 2; do
  for v in "" "MSA_PAIR_TI=16" "MSA_PAIR_TI=16 MSA_PAIR_K=4"; do
    echo "== C3 $v" >> $OUT/c3.txt
    env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('kernels_ms'))" >> $OUT/c3.txt
  done
done
P=29517
timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P tools/c5_collective.py 15 > $OUT/c5_collective.jsonl 2> $OUT/c5_collective.err; echo "collective rc=$?"
BATCH_R05=1 timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((P+1)) tools/c5_collective.py 15 > $OUT/c5_collective_r05.jsonl 2> $OUT/c5_collective_r05.err; echo "collective r05 rc=$?"
cut -c1-330 $OUT/time.jsonl; cut -c1-330 $OUT/time_small.jsonl; cat $OUT/c3.txt; cat $OUT/c5_collective.jsonl; echo; cat $OUT/c5_collective_r05.jsonl; tail -5 $OUT/c5_collective.err
