# Sourced by every wrapper under tools/: a fatal error must name itself in the log of the run it happens in.
# glibc writes its heap / stack / fortify messages to /dev/tty unless LIBC_FATAL_STDERR_ is set (a run redirected into
# gpurun_out/ loses them); libdctfp.so prints the native frames of a SIGABRT / SIGSEGV when DCTFP_CRASH_BACKTRACE=1;
# PYTHONFAULTHANDLER prints the Python frames.
export LIBC_FATAL_STDERR_=1
export PYTHONFAULTHANDLER=1
export DCTFP_CRASH_BACKTRACE=1
