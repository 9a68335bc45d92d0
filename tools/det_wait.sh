#!/bin/bash
LWHIP_LS_TIMING=1 LWHIP_LIB=$PWD/lightweaver_amd/variants/liblwhip_lstiming.so python3 tools/det_time.py det 10240 2>&1 | grep -E "Nlambda|lane sweep"
LWHIP_LS_TIMING=1 LWHIP_LIB=$PWD/lightweaver_amd/variants/liblwhip_lstiming.so python3 tools/det_time.py default 10240 2>&1 | grep -E "Nlambda|lane sweep"
