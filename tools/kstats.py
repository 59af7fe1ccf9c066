import csv,glob,sys
fs=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(fs[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:30]:
    print('%-72s %6s %9.1f us %5.1f%%' % (r['Name'][:72], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/tot*100))
