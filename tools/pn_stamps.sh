for abl in 4096 4097 4104 4352; do
  echo "== DVQ_PN_ABL=$abl"
  DVQ_DIAG_LIB=1 DVQ_PN_STATS=1 DVQ_PN_ABL=$abl PN_REP=1 PN_B=4096 python3 tools/pn_quick.py 2>&1 | grep -E "^C=|phase ticks" | cut -c1-220 | tail -8
done
