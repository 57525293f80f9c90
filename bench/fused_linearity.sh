for n in 2.5e6 5e6 1e7 2e7 4e7; do python bench/fused_quick.py $n 100 2>&1 | grep "^p="; done
