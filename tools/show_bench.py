import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print("ms/step %.2f  distinct/s %.4g  occ/s %.4g" % (d['ms_per_step'], d['value'], d['occurrences_per_s']))
tot = 0
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']):
    tot += v['ms_per_step']
    print("  %-18s %7.2f ms  x%-3g  %s GB/s" % (k, v['ms_per_step'], v['launches_per_step'], ("%.0f" % v['algorithmic_GBps']) if v['algorithmic_GBps'] else "-"))
print("  kernels total %.1f ms" % tot)
