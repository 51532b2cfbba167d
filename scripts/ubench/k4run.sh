cd /root/repo
B=scripts/ubench/build/train3_bench
K4=1 timeout -k 5 120 $B 64800 5 0
for s in 30000 60000 90000 130000 180000; do echo "stagger $s"; K4=2 K4_STAGGER=$s timeout -k 5 60 $B 64800 5 9 | grep k4; done
