"""Tools that turn engineering knobs (dctfp_set_option names that only libdctfp_experiments.so knows) import this first."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('DCTFP_LIBRARY', os.path.join(ROOT, 'dctdomain_amd', 'libdctfp_experiments.so'))
