import re, collections, statistics
acc = collections.defaultdict(list)
for l in open("gpurun_out/r3/ab.txt"):
    m = re.match(r"(\S+) (cfg\d) .*kernel_us=([\d.]+)", l)
    if m: acc[(m.group(2), m.group(1))].append(float(m.group(3)))
for k in sorted(acc): print(k[0], k[1], "median %.2f" % statistics.median(acc[k]), acc[k])
