for sp in 131072 262144 1048576 4194304; do
  echo "RTX_SPLIT=$sp"
  RTX_SPLIT=$sp timeout -k 10 100 python tools/dbg/rank_time.py 2>&1 | grep -E "N=1 |N=8 band_rows=30" || exit 1
done
