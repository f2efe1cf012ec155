# same-session A/B of two builds of the library on the fused launches of a layer: old = tinychatengine_amd/lib/abl/libtce_old.so, new = the in-tree build
for v in ${ORDER:-old new old new}; do
  if [ $v = old ]; then export TCE_LIB_PATH=$PWD/tinychatengine_amd/lib/abl/libtce_old.so; else unset TCE_LIB_PATH; fi
  for sh in ${SHAPES:-baseline llama3-8b}; do
    echo "== $v $sh"
    FUSED_AB_SHAPES=$sh FUSED_AB_AUTO_ONLY=1 python scripts/fused_launch_ab.py 2>&1 | grep launch | sed "s/\"auto_is\": \"gemv-i8 rows-per-pass=1 group=128\", //"
  done
done
