for b in 32768 65536 98304 100000 114688 131072 163840 196608 262144 1000000; do MT6_B=$b MT6_NOCHECK=1 MT6_ROUNDS=5 timeout -k 10 100 python tools/mt6_time.py | sed -e "s/^/B=$b  /"; done
