mkdir -p gpurun_out/abl; rm -f gpurun_out/abl/out.txt
for sh in ${SHAPES:-baseline llama3-8b}; do
for n in ${ABLS:-0}; do
  echo "== $sh: tiles per wave $n (3 = two tiles per workgroup sharing the prologue)" >> gpurun_out/abl/out.txt
  FUSED_AB_SHAPES=$sh FUSED_AB_AUTO_ONLY=1 FUSED_AB_I8_TILES=$n timeout 300 python scripts/fused_launch_ab.py 2>&1 | grep "launch" | sed "s/\"auto_is\": \"gemv-i8 rows-per-pass=1 group=128\", //" >> gpurun_out/abl/out.txt
done; done
cut -c1-140 gpurun_out/abl/out.txt
