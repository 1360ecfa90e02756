python scripts/ab2.py --run base lazy lazylong lazytail
python -m pytest tests -x -q -m gpu -k "fused or regression or full_size" 2>&1 | tail -3
CFGS=32x12 DBG=256 python scripts/timing.py 2>&1 | tail -16
