for L in - rabl1 rabl2 rabl4; do
  if [ "$L" = "-" ]; then LIB=relationnetworks-clevr_amd/librn_hip.so; else LIB=tools/dbg/libs/librn_$L.so; fi
  echo "== $L"; RN_LIB=$LIB python tools/time_wgrad.py 2>/dev/null | head -3
done
