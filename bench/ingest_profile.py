"""Where the host time of the real-data branch goes: python bench/ingest_profile.py file.csv (airline defaults)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from dlsa_amd import dummies, ingest
path = sys.argv[1]
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
T = [time.perf_counter()]
def tick(name):
    torch.cuda.synchronize(); T.append(time.perf_counter()); print("%-38s %.3f s" % (name, T[-1] - T[-2]), flush=True)
use, dc = list(ingest.AIRLINE_USECOLS_X), list(ingest.AIRLINE_DUMMY_COLUMNS)
pdf = ingest.read_csv_frame(path, use, ingest.AIRLINE_Y, dc); tick("read_csv_frame (arrow)")
counts = dummies.dummy_factors_counts(pdf, dc); tick("dummy_factors_counts")
info = dummies.select_dummy_factors(counts, [1, 1, 0.8, 0.9, 0.9], "000_OTHERS"); tick("select_dummy_factors")
baseline = [(c + "_000_OTHERS") if len(info["factor_dropped"][c]) > 0 else sorted(info["factor_selected_names"][c])[0] for c in dc]
data_info = ingest.data_info_from_frame(pdf, [c for c in use if c not in dc]); tick("data_info_from_frame")
sh = ingest.shard_from_frame(pdf, ingest.AIRLINE_Y, info, baseline, data_info, True); tick("shard_from_frame (codes + upload + group)")
print("rows", len(pdf), "p", len(sh["spec"].names))
