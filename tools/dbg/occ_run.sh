# A/B of weight-gradient variants: alone (tools/time_wgrad.py) and in the step.  usage: occ_run.sh "<libs>" [reps] [bench args]
LIBS=${1:-"- ring"}; N=${2:-2}; shift 2
for L in $LIBS; do
  if [ "$L" = "-" ]; then LIB=relationnetworks-clevr_amd/librn_hip.so; else LIB=tools/dbg/libs/librn_$L.so; fi
  echo "== $L"; RN_LIB=$LIB python tools/time_wgrad.py 2>/dev/null | sed -n 3p
done
bash tools/dbg/ab_libs.sh "$LIBS" $N "$@"
