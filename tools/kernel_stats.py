import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in r["Name"]:
        print("  ", r["Name"][:44].ljust(46), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
