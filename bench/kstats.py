import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "syrk" in n or "logit_kernel<4, 4, true, false, true>" in n:
        print("   %-70s calls %s avg %.1f us min %.1f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
