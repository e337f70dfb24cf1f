# A/B of the pipelined GEMMs' blocked accumulation: variants/libglnn_<v>.so built with scripts/build_variant.sh (see NOTES.md, round 4)
VARIANTS=${VARIANTS:-"old blk8 blk16"}
for rep in 1 2; do for v in $VARIANTS; do GLNN_LIB_PATH=$PWD/variants/libglnn_$v.so python scripts/gemm_sustained.py $v 2>&1 | grep -v amdgpu.ids; done; done | tee gpurun_out/gemm_blk_ab.txt | cut -c1-190
