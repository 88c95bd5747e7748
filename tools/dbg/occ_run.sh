# A/B of weight-gradient variants: alone (tools/time_wgrad.py) and in the step.  usage: occ_run.sh "<libs>"
LIBS=${1:-"- ring"}
for L in $LIBS; do
  if [ "$L" = "-" ]; then LIB=relationnetworks-clevr_amd/librn_hip.so; else LIB=tools/dbg/libs/librn_$L.so; fi
  echo "== $L"; RN_LIB=$LIB python tools/time_wgrad.py 2>/dev/null | head -3
done
bash tools/dbg/ab_libs.sh "$LIBS" 2
