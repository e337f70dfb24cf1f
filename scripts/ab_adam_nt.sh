export TMPDIR=/tmp
for v in base adamnt base adamnt; do
  export GLNN_LIB_PATH=$PWD/variants/libglnn_$v.so
  python scripts/trace_student_any.py 100-2048-2048-47 4096 batch 0.2 kl 2>&1 | grep "ms per" | sed "s/^/$v /"
done
for v in base adamnt; do
  export GLNN_LIB_PATH=$PWD/variants/libglnn_$v.so
  rm -rf /tmp/st_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$v -- python scripts/trace_student_any.py 100-2048-2048-47 4096 batch 0.2 kl > /dev/null 2>&1
  f=$(ls /tmp/st_$v/*/*kernel_stats.csv | head -1); grep -i "adam" $f | cut -c1-120 | sed "s/^/$v /"
done
