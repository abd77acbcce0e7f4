#!/bin/bash
# round 5, session l: the one-read top-k with plateau entries placed by position: parity, kernel times, the flush
cd "$(dirname "$0")/.." && . tools/env.sh
bash tools/r05_c.sh || exit 1
bash tools/r05_g.sh || exit 1
bash tools/r05_i.sh || exit 1
bash tools/r05_h.sh 2>&1 | grep -E "passed|failed|flush of"
