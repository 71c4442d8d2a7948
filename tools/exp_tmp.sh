OUT=gpurun_out/fin3; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest.txt
for shape in "1024 100 1000" "1024 60 600" "4096 40 300" "512 128 2000" "256 300 1200" "128 500 2000"; do
  timeout 300 python tools/small_batch.py $shape 2>/dev/null; MSA_BATCH_ENGINE=0 timeout 300 python tools/small_batch.py $shape 2>/dev/null
done > $OUT/small_batch.jsonl
MSA_BATCH_COLS_MAX=0 timeout 300 python tools/small_batch.py 1024 100 1000 2>/dev/null >> $OUT/small_batch.jsonl
MSA_BATCH_ENGINE_MAX=1e12 timeout 300 python tools/c5_engine.py 2>/dev/null > $OUT/c5_engine.jsonl
MSA_BATCH_ENGINE=0 timeout 300 python tools/c5_engine.py 2>/dev/null >> $OUT/c5_engine.jsonl
bash tools/gpu_fuzz.sh 200 150 > /dev/null 2>&1
cp gpurun_out/fuzz.txt $OUT/fuzz.txt
cat $OUT/pytest.txt $OUT/c5_engine.jsonl $OUT/fuzz.txt
