for w in 4 8 16; do for nt in 0 1; do echo "waves=$w nt=$nt"; SDFK_SAMPLE_WAVES=$w SDFK_SAMPLE_NT=$nt python tools/sample_probe.py 2>&1 | grep -E "plane|sphere"; done; done
