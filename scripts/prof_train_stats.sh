export TMPDIR=/tmp GLNN_BENCH_EPOCHS=2
OUT=$PWD/gpurun_out/train_stats_${1:-c}; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python scripts/bench_train_sage.py ogbn-products > $OUT/log.txt 2>&1
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/stats
head -40 $OUT/kernel_stats.csv | cut -c1-200
