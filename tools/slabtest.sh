python tools/tune_tiles.py 2>&1 | grep -E "PICK M"
for cfg in 4 0 1 3; do for w in 512 1024; do for tn in 32 64 128; do echo "cfg=$cfg wgs=$w tn=$tn: $(S2VT_SLAB_CFG=$cfg S2VT_SLAB_WGS=$w S2VT_SLAB_TILE_N=$tn python tools/tune_tn.py | grep -E 'class 0 (64x32|64x64|64x128\(2x2\)|128x32)' | awk '{printf "%s %s/%s ", $3, $7, $8}')"; done; done; done
