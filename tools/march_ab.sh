for s in 1 2; do echo -n "split $s: "; EGO_MARCH_SPLIT=$s EGO_ALLOW_STALE_LIB=1 python tools/march_timing.py 2>&1 | tail -1 | cut -c1-110; done
