#!/usr/bin/env python
"""Run a command, then print its peak resident set size and wall time to stderr (stand-in for /usr/bin/time -v)."""
import resource, subprocess, sys, time
t0 = time.time()
rc = subprocess.call(sys.argv[1:])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print(f"peak_rss_gb {ru.ru_maxrss / 1048576.0:.2f} elapsed_s {time.time() - t0:.1f}", file=sys.stderr)
sys.exit(rc)
