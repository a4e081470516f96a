python tools/b16_quick.py 32 2>&1 | grep -o '"fwd_ms": [0-9.]*'
RECON_PROP_B16_ABL=1 python tools/b16_quick.py 32 2>&1 | grep -o '"fwd_ms": [0-9.]*'
