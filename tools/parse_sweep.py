import re,sys
for line in open(sys.argv[1]):
    if "GFLOP" not in line: continue
    name=line.split()[0]
    ts=[(float(m.group(2)),m.group(1)) for m in re.finditer(r"(\w+):c\d+/\d+t\s+([0-9.]+)us", line)]
    print(name, min(ts) if ts else None)
