#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
for th in 16 8; do echo "== $th copy threads"; SDFK_COPY_THREADS=$th timeout 300 python3 tools/managed_probe.py 2>&1 | grep "huge 0\|copy call\|reused"; done
