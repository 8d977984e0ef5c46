import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=float(sys.argv[2]) if len(sys.argv)>2 else 12.0
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 22]:
    print("%-60s calls=%5s tot_ms=%7.3f avg_us=%8.2f pct=%5.1f"%(r['Name'].replace('chip::dev::(anonymous namespace)::','')[:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage'])))
print("total kernel ms per step", tot/1e6/steps)
