#!/bin/bash
python3 tools/det_time.py both 10240 2560 2>&1 | grep Nlambda
timeout 300 python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -k "deterministic or generic or repeatab" -p no:cacheprovider 2>&1 | tail -3
