BZ_ENC_TRACE=1 timeout 300 python tools/e2e_multi.py 1024 0 2>&1 | grep "bz_enc" | tail -11
