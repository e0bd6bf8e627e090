for d in 0 1 8 2 4 5; do
  echo "RTX_DBG=$d"
  RTX_DBG=$d timeout -k 10 100 python tools/dbg/ablate.py || exit 1
done
