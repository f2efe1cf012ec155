export TMPDIR=/tmp
echo "box: $(cat /sys/class/drm/card*/device/unique_id | head -1)"
for v in current "$@" current; do
  if [ $v = current ]; then unset TCE_LIB_PATH; else export TCE_LIB_PATH=$PWD/tinychatengine_amd/lib/abl/libtce_$v.so; fi
  echo "== $v"; STRESS_CASES=${STRESS_CASES:-0,1} STRESS_MODES=${STRESS_MODES:-61,62,64} timeout 120 python scripts/probes/pk_stress.py ${SECS:-4} 2>&1 | grep -v amdgpu | cut -c1-700
done
