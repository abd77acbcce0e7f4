#!/bin/bash
cd "$(dirname "$0")/.." && . tools/env.sh
bash tools/r05_c.sh || exit 1
VARIANTS="stop1 stop2 full" bash tools/r05_d.sh
