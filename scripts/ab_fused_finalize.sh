for i in 1 2; do for f in 0 1; do echo "fused_finalize=$f"; GLNN_STUDENT_FUSED_FINALIZE=$f python scripts/bench_student.py products-MLP3w8 2>&1 | grep -v amdgpu.ids | tail -2; done; done
