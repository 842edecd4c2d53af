import csv, sys, glob
f = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else sorted(glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f, "total ms", tot / 1e6)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    n = r['Name'].replace('f5::', '')
    n = n.replace('_ZN2f5', '').replace('gemm_tn_glds_kernel', 'G2').replace('gemm_tn_kernel', 'G1')[:100]
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f}ms {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%  {n}")
