for g in 3 30; do for t in 8 16 32; do RTX_PK_GROW=$g python3 tools/split_ab.py 2048:$t:4 2>&1 | tail -1 | sed "s/^/grow=$g /"; done; done
