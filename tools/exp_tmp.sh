OUT=$(pwd)/gpurun_out/fin4; mkdir -p $OUT; ROOT=$(pwd)
export TMPDIR=/tmp
( cd /tmp; : > $OUT/small_kernel_stats.txt
  for a in "46 1181 strict" "100 1000 automated1" "209 1227 strictplus" "500 2000 strict" "209 1227 overlap" "209 1227 representative" "1000 4000 automated1"; do
    rm -rf /tmp/small_prof
    timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/small_prof -- python3 $ROOT/tools/small_one.py $a 200 > /tmp/small_prof.log 2>&1
    grep "per upload" /tmp/small_prof.log >> $OUT/small_kernel_stats.txt || tail -3 /tmp/small_prof.log >> $OUT/small_kernel_stats.txt
    f=$(find /tmp/small_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -d, -f1-4 | cut -c1-160 >> $OUT/small_kernel_stats.txt
  done )
cat $OUT/small_kernel_stats.txt
