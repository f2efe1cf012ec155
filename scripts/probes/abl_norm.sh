mkdir -p gpurun_out/abl; rm -f gpurun_out/abl/out.txt
for n in ${ABLS:-0}; do
  echo "== tiles per wave $n" >> gpurun_out/abl/out.txt
  FUSED_AB_AUTO_ONLY=1 FUSED_AB_I8_TILES=$n timeout 300 python scripts/fused_launch_ab.py 2>&1 | grep "launch" >> gpurun_out/abl/out.txt
done
cut -c1-140 gpurun_out/abl/out.txt
