for sk in 0 32 64 96 128 192 256; do echo "skew=$sk"; FSEA_SKEW=$sk python scripts/launch_size.py 8192 | grep -E "=4096 |=32768 "; done
