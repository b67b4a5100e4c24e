#!/bin/bash
for cfg in "0,0,0" "1,1024,1" "1,1024,2" "1,1024,3" "1,782,1" "1,782,2" "1,782,3" "2,1,1" "2,1,2" "2,8,2" "2,4,2" "2,32,2" "1,512,2"; do SSMQ_STAGGER=$cfg MT6_NOCHECK=1 MT6_ROUNDS=5 python tools/mt6_time.py | sed "s/^/stagger=$cfg /"; done
