#!/bin/bash
# the driver's real-data branch on an airline-shaped synthetic CSV: bench/driver_csv.sh [rows]
ROWS=${1:-3000000}; D=gpurun_out/driver_csv; mkdir -p $D
python3 - <<PY
import numpy as np, pandas as pd, time
rng = np.random.default_rng(4); n = $ROWS
def zipf(L): p = 1.0 / np.arange(1, L + 1); return p / p.sum()
raw = pd.DataFrame({"Year": rng.integers(1987, 2009, n), "Month": rng.integers(1, 13, n), "DayofMonth": rng.integers(1, 29, n),
                    "DayOfWeek": rng.integers(1, 8, n), "DepTime": rng.uniform(0, 2400, n).round(0), "CRSDepTime": rng.uniform(0, 2400, n).round(0),
                    "CRSArrTime": rng.uniform(0, 2400, n).round(0),
                    "UniqueCarrier": np.array(["C%02d" % i for i in range(25)])[rng.choice(25, n, p=zipf(25))],
                    "ActualElapsedTime": rng.normal(120, 60, n).round(0), "Origin": np.array(["A%03d" % i for i in range(300)])[rng.choice(300, n, p=zipf(300))],
                    "Dest": np.array(["A%03d" % i for i in range(300)])[rng.choice(300, n, p=zipf(300))], "Distance": rng.normal(700, 400, n).round(0)})
eta = 0.3 * (raw["Distance"] - 700) / 400 + 0.2 * (raw["DayOfWeek"] == 5) - 0.2 * (raw["UniqueCarrier"] == "C01")
raw["ArrDelay"] = np.where(rng.random(n) < 1 / (1 + np.exp(-eta)), rng.uniform(1, 90, n), -rng.uniform(0, 30, n)).round(0)
raw.loc[rng.choice(n, n // 200, replace=False), "DepTime"] = np.nan
t = time.time(); raw.to_csv("$D/air.csv", index=False, na_rep="NA"); print("csv written: %d rows, %.1f s" % (n, time.time() - t))
PY
ls -la $D/air.csv | awk '{print "csv bytes", $5}'
rm -f $D/dummy_info.pkl $D/data_info.csv
python3 projects/logistic_dlsa.py --csv $D/air.csv --fit-intercept --dummy-info $D/dummy_info.pkl --data-info $D/data_info.csv --save $D/res.pkl 2>&1 | grep -v amdgpu | head -12
[ -n "$KEEP_CSV" ] || rm -f $D/air.csv
