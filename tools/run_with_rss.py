"""`/usr/bin/time -v` for a box that has none: runs the command, then prints its wall time and the largest resident set among
it and its descendants (ru_maxrss of RUSAGE_CHILDREN is the maximum over all waited-for descendants, in KiB on Linux)."""
import resource, subprocess, sys, time
t0 = time.time()
rc = subprocess.call(sys.argv[1:])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print(f'Elapsed (wall clock) seconds: {time.time() - t0:.1f}', file=sys.stderr)
print(f'Maximum resident set size of any process of the build: {ru.ru_maxrss / 1024:.0f} MiB', file=sys.stderr)
sys.exit(rc)
