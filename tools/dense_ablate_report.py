#!/usr/bin/env python3
"""Prints gpurun_out/dense_ablate.txt (tools/dense_ablate.sh) as a table: variant, ms/step, average ns per k_ds_* kernel."""
import re
import sys

for line in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/dense_ablate.txt"):
    if line.startswith("=="):
        print(line.strip())
        continue
    m = re.match(r'"(?:void )?(k_ds_\w+(?:<[^>]*>)?)', line)
    f = line.rsplit('",', 1)[1].strip().split(",")
    print("   %-32s calls %s avg_ns %s" % (m.group(1), f[0], f[2]))
