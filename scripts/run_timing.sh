echo "== NBUF=4 (HBM regime)"; NBUF=4 CFGS=32x12 DBG=256 python scripts/timing.py 2>&1 | tail -16
NBUF=4 CFGS=32x12 python scripts/timing.py 2>&1 | tail -14
