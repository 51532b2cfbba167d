cd /root/repo
B=scripts/ubench/build/train3_bench
KC=1 timeout -k 5 120 $B ${1:-64800} 5 0
