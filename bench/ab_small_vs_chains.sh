for rows in 10000 20000 30000 45000 60000; do
  n=$((rows*20))
  echo "rows/partition=$rows K=20 p=50"
  python bench/irls_trace.py $n 50 20 2>&1 | grep "^fit" | tail -1
  DLSA_IRLS_SMALL=0 python bench/irls_trace.py $n 50 20 2>&1 | grep "^fit" | tail -1
done
for rows in 10000 30000; do
  n=$((rows*20)); echo "rows/partition=$rows K=20 p=100"
  python bench/irls_trace.py $n 100 20 2>&1 | grep "^fit" | tail -1
  DLSA_IRLS_SMALL=0 python bench/irls_trace.py $n 100 20 2>&1 | grep "^fit" | tail -1
done
