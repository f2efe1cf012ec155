# same-session A/B of builds of the library on the fused launches of a layer: <name> = tinychatengine_amd/lib/abl/libtce_<name>.so, "new" = the in-tree build
for v in ${ORDER:-old new old new}; do
  if [ $v = new ]; then unset TCE_LIB_PATH; else export TCE_LIB_PATH=$PWD/tinychatengine_amd/lib/abl/libtce_$v.so; fi
  for sh in ${SHAPES:-baseline llama3-8b}; do
    echo "== $v $sh"
    FUSED_AB_SHAPES=$sh FUSED_AB_AUTO_ONLY=1 python scripts/fused_launch_ab.py 2>&1 | grep launch | sed "s/\"auto_is\": \"gemv-i8 rows-per-pass=1 group=128\", //"
  done
done
