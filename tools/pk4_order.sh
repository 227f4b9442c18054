#!/bin/bash
# slot order of the 4-wide shadow-ray records: 0 stack need (default), 1 area desc, 2 triangles desc, 3 area asc, 4 triangle density desc, 5 triangles asc
for m in "$@"; do echo "order=$m"; RTX_PK4_ORDER=$m python3 tools/pk_sweep.py 8:3; RTX_PK4_ORDER=$m python3 tools/wide_need.py 2>/dev/null | grep atrium; done
